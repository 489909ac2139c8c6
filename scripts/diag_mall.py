"""Do the meta-steps run out of the Infinity Cache when nothing else touches memory?  Passes of 32 steps timed by events
around the steps only; between two passes, in line on the same stream: (a) nothing, (b) a 1-GB device copy that pushes
everything out of the 256-MB cache.  (32 slides x ~1,350 selected rows x 2 KB = 86 MB: the selected rows of a whole epoch
fit, and in scripts/diag_alone.py they are re-read every pass.)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moc_amd import engine, main_moc as M, synth
dev = torch.device("cuda:0")
Cc, D, j, K = 2, 512, 400, 10
W, We = synth.make_bank(1234, D, Cc)
M.set_classifier_bank(W.to(dev), We.to(dev))
bags = [synth.make_bag_device(1234 + i, 15000, D, We, Cc, i % Cc, dev, torch.float32) for i in range(32)]
res = M.ResidentBags(bags, [i % Cc for i in range(32)], dev)
torch.manual_seed(0)
model = M.senet(D, 4).to(dev); opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
bank = M._bank_for(res.X, dev)
plan = res.train_plan(Cc, Cc + 4, j, K, [])
batch, lab = plan["batch"], plan["labels"]
m, kept = engine.draw_row_masks(batch.total); batch.set_mask(m, kept); batch.phase_a(bank)
meta = engine.MetaState(model, opt)
big_a = torch.empty(1 << 30, dtype=torch.uint8, device=dev); big_b = torch.empty_like(big_a)
for _ in range(5):
    engine.train_steps(batch, meta, lab, 0, 32, 15)
torch.cuda.synchronize()
for label, between in (("nothing between the passes", None), ("a 1-GB copy between the passes", lambda: big_b.copy_(big_a)),
                       ("nothing between the passes", None), ("a 1-GB copy between the passes", lambda: big_b.copy_(big_a))):
    evs = []
    for _ in range(30):
        if between is not None:
            between()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); engine.train_steps(batch, meta, lab, 0, 32, 15); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 / 32 for a, b in evs)
    print(f"{label:34s} median {ts[len(ts) // 2]:6.2f} us/step  (min {ts[0]:.2f}, max {ts[-1]:.2f})")
