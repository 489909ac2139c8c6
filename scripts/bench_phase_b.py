"""Phase-B micro-benchmark: GPU time per meta-step and per kernel, launches back to back."""
import os, sys, time, types, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_num_threads(8)
from moc_amd import engine, main_moc as M, synth
from moc_amd._lib import lib, ptr, check
dev = torch.device("cuda:0")
Cc, D, j, K = int(os.environ.get("C", 2)), 512, 400, 10
dtype = torch.bfloat16 if os.environ.get("DT", "bf16") == "bf16" else torch.float32
W, We = synth.make_bank(1234, D, Cc)
M.set_classifier_bank(W.to(dev), We.to(dev))
n = 32
bags = [synth.make_bag_device(1234 + i, 15000, D, We, Cc, i % Cc, dev, dtype) for i in range(n)]
res = M.ResidentBags(bags, [i % Cc for i in range(n)], dev)
torch.manual_seed(0)
model = M.senet(D, 4).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
bank = M._bank_for(res.X, dev)
plan = res.train_plan(Cc, Cc + 4, j, K, [])
batch, lab = plan["batch"], plan["labels"]
m, kept = engine.draw_row_masks(batch.total)
batch.set_mask(m, kept)
batch.phase_a(bank)
meta = engine.MetaState(model, opt)
_, ws = batch.meta_ws()
s = engine._stream()
def timeit(fn, reps, per):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(reps): fn()
    e1.record(); th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * per), th * 1e6 / (reps * per)
print("S per slide:", batch.n_sel.cpu().tolist()[:8], "...")
g, h = timeit(lambda: engine.train_steps(batch, meta, lab, 0, n, 15), 20, n)
print(f"train_steps: {g:7.2f} us/step on the GPU timeline, host issue {h:6.2f} us/step")
g, h = timeit(lambda: check(lib().moc_meta_forward(C.byref(batch.c), C.byref(meta.c), C.byref(ws), 3, 1, 15, s), "f"), 200, 1)
print(f"meta_forward (1 slide): {g:7.2f} us  (host {h:5.2f})")
g, h = timeit(lambda: check(lib().moc_pool_loss(C.byref(batch.c), C.byref(ws), ptr(lab), 3, 1, s), "p"), 200, 1)
print(f"pool_loss    (1 slide): {g:7.2f} us  (host {h:5.2f})")
g, h = timeit(lambda: check(lib().moc_meta_forward(C.byref(batch.c), C.byref(meta.c), C.byref(ws), 0, n, 15, s), "f"), 50, n)
print(f"meta_forward (batched {n}): {g:7.2f} us/slide")
g, h = timeit(lambda: check(lib().moc_pool_loss(C.byref(batch.c), C.byref(ws), ptr(lab), 0, n, s), "p"), 50, n)
print(f"pool_loss    (batched {n}): {g:7.2f} us/slide")
g, h = timeit(lambda: batch.phase_a(bank), 20, n)
print(f"phase A      (batched {n}): {g:7.2f} us/slide")
t = batch.meta_ws()[0]
g, h = timeit(lambda: check(lib().moc_ce_loss(ptr(t["pooled"]), ptr(lab), 1, Cc, ptr(t["loss"]), ptr(t["pred"]), s), "c"), 500, 1)
print(f"ce_loss (1 WG, trivial): {g:7.2f} us  (host {h:5.2f})   <- per-kernel floor on this timeline")
x = torch.zeros(64, device=dev)
g, h = timeit(lambda: x.add_(1.0), 500, 1)
print(f"torch x.add_(1) on 64 floats: {g:7.2f} us  (host {h:5.2f})")
