"""Phase stamps of the wide one-launch step (pool_w1_step_wide_kernel, workgroup 0) and its forward at the
EBRAINS-30 shape, from the diagnostic build (make -C moc_amd/csrc stamps)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _stamps  # noqa: E402,F401  (builds build/libmoc_hip_stamps.so here if missing; sets MOC_HIP_LIB)
import torch
sys.path.insert(0, ROOT)
torch.set_num_threads(8)
from moc_amd import engine, main_moc as M, synth
from moc_amd._lib import lib
dev = torch.device("cuda:0")
DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
Cc, D, j, K, n = 30, 512, 400, 10, 30
W, We = synth.make_bank(1234, D, Cc)
M.set_classifier_bank(W.to(dev), We.to(dev))
bags = [synth.make_bag_device(1234 + i, 15000, D, We, Cc, i % Cc, dev, DT) for i in range(n)]
res = M.ResidentBags(bags, [i % Cc for i in range(n)], dev)
torch.manual_seed(0)
model = M.senet(D, 4).to(dev); opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
bank = M._bank_for(res.X, dev)
plan = res.train_plan(Cc, Cc + 4, j, K, [])
batch, lab = plan["batch"], plan["labels"]
m, kept = engine.draw_row_masks(batch.total); batch.set_mask(m, kept); batch.phase_a(bank)
meta = engine.MetaState(model, opt)
h = lib(); h.moc_debug_stamps.restype = C.c_int; h.moc_debug_stamps.argtypes = [C.c_void_p, C.c_int]
names = {0: "fwd begin", 1: "fwd mfma done", 2: "fwd end", 40: "step begin", 41: "top-K picked up, CE done", 42: "pair operands requested",
         43: "hidden rows / masks done", 50: "chunk 0: rows in LDS, next requested", 52: "chunk 0: barrier",
         53: "chunk 0: MFMAs done", 44: "W1 gradient chunks done", 45: "small gradients done", 46: "step end"}
acc = {}
for rep in range(20):
    engine.train_steps(batch, meta, lab, 0, n, 15)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 128)()
    assert h.moc_debug_stamps(buf, 128) == 0
    t = {k: buf[k] for k in names}
    order = sorted(names, key=lambda k: t[k])
    for a_, b_ in zip(order, order[1:]):
        acc.setdefault((a_, b_), []).append((t[b_] - t[a_]) / 100.0)
for (a_, b_), v in acc.items():
    v.sort()
    print(f"{names[a_]:28s} -> {names[b_]:28s} median {v[len(v)//2]:6.2f} us")
