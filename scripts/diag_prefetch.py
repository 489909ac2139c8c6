"""Potential of warming the next slide's selected rows (MALL / L2) before its forward kernel: time
moc_meta_forward for one slide with and without a preceding read of the rows it gathers."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.set_num_threads(8)
from moc_amd import engine, main_moc as M, synth
dev = torch.device("cuda:0")
C, D, j, K = 2, 512, 400, 10
W, We = synth.make_bank(1234, D, C)
M.set_classifier_bank(W.to(dev), We.to(dev))
bags = [synth.make_bag_device(1234 + i, 15000, D, We, C, i % C, dev, torch.bfloat16) for i in range(32)]
res = M.ResidentBags(bags, [i % C for i in range(32)], dev)
torch.manual_seed(0)
model = M.senet(D, 4).to(dev)
bank = M._bank_for(res.X, dev)
plan = res.train_plan(C, C + 4, j, K, [])
batch = plan["batch"]
m, kept = engine.draw_row_masks(batch.total); batch.set_mask(m, kept); batch.phase_a(bank)
meta = engine.MetaState(model)
torch.cuda.synchronize()
ns = batch.n_sel.cpu().tolist()
junk = torch.empty(64 * 1024 * 1024, device=dev)          # 256 MB: flushes L2 and most of the MALL between trials
def trial(warm):
    ts = []
    for rep in range(3):
        for s in range(32):
            junk.add_(1.0)
            o = batch.row_off_host[s]
            rows = batch.sel_row[o:o + ns[s]]
            if warm:
                res.X[rows].float().sum()                  # reads the rows the forward will gather
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); engine.meta_forward(batch, meta, s, 1, 15); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]
for warm in (False, True, False, True):
    med, best = trial(warm)
    print(f"rows warmed beforehand: {warm!s:5s}  moc_meta_forward (image + forward launches) median {med:6.2f} us  min {best:6.2f} us")
