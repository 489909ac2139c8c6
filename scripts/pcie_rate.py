"""PCIe-inclusive rate: train() fed host-resident bags every epoch (as the reference's DataLoader
does) instead of HBM-resident ones.  Reported in DESIGN.md only -- never bench.py's value."""
import os, sys, time, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_num_threads(8)
from moc_amd import main_moc as M, synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import helpers as H
dev = torch.device("cuda:0")
C, D, j, K = 2, 512, 400, 10
W, We = synth.make_bank(1234, D, C)
M.set_classifier_bank(W.to(dev), We.to(dev))
bags = [synth.make_bag_device(1234 + i, 15000, D, We, C, i % C, dev, torch.float32).cpu() for i in range(32)]
labels = [i % C for i in range(32)]
for pin in (False, True):
    hb = [b.pin_memory() for b in bags] if pin else bags
    for dt in ("fp32", "bf16"):
        args = H.make_args(C, j, K); args.bag_dtype = dt
        torch.manual_seed(0)
        model = M.senet(D, 4).to(dev); opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
        loader = H.ListLoader(hb, labels)
        for _ in range(2): M.train(model, loader, opt, dev, args)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 10
        for _ in range(n): M.train(model, loader, opt, dev, args)
        torch.cuda.synchronize(); dtm = time.perf_counter() - t0
        print(f"host bags (pinned={pin}) fp32 in host RAM -> {dt} in HBM: {32 * n / dtm:.0f} meta-steps/s ({dtm / n * 1e3:.1f} ms/epoch, {32 * 15000 * 512 * 4 / (dtm / n) / 1e9:.1f} GB/s host->device)")
