"""Where a meta-step spends its microseconds: phase stamps from the diagnostic build
(make -C moc_amd/csrc stamps; built into build/ by scripts/_stamps.py).  Shares, not totals."""
import os, sys, ctypes as C
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _stamps  # noqa: E402,F401  (builds build/libmoc_hip_stamps.so here if missing; sets MOC_HIP_LIB)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_num_threads(8)
from moc_amd import engine, main_moc as M, synth
from moc_amd._lib import lib, check
dev = torch.device("cuda:0")
DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
Cc, D, j, K = 2, 512, 400, 10
W, We = synth.make_bank(1234, D, Cc)
M.set_classifier_bank(W.to(dev), We.to(dev))
bags = [synth.make_bag_device(1234 + i, 15000, D, We, Cc, i % Cc, dev, DT) for i in range(32)]
res = M.ResidentBags(bags, [i % Cc for i in range(32)], dev)
torch.manual_seed(0)
model = M.senet(D, 4).to(dev); opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
bank = M._bank_for(res.X, dev)
plan = res.train_plan(Cc, Cc + 4, j, K, [])
batch, lab = plan["batch"], plan["labels"]
m, kept = engine.draw_row_masks(batch.total); batch.set_mask(m, kept); batch.phase_a(bank)
meta = engine.MetaState(model, opt)
h = lib(); h.moc_debug_stamps.restype = C.c_int; h.moc_debug_stamps.argtypes = [C.c_void_p, C.c_int]
h.moc_debug_stamps_set.restype = C.c_int; h.moc_debug_stamps_set.argtypes = [C.c_void_p, C.c_int]
names = {0: "fwd begin", 3: "fwd row id here", 5: "fwd tile in LDS", 6: "fwd barrier", 1: "fwd mfma done", 7: "fwd partials met", 8: "fwd hidden done", 9: "fwd gates done", 2: "fwd end", 10: "pool begin",
         19: "pool records requested", 20: "pool records here", 11: "pool wave-max done", 12: "pool candidates done",
         13: "pool extraction done", 14: "pool CE done", 15: "pool W2 staged", 16: "pool pairs done", 17: "pool dh done", 18: "pool end",
         }   # (the W1 update is part of the pool kernel now: one-launch step)
acc = {}
BESIDE = len(sys.argv) > 2 and sys.argv[2] == "beside"   # score passes of the other work-array set on a side stream, all along
if BESIDE:
    other = plan["batches"][1]
    m2, kept2 = engine.draw_row_masks(other.total); other.set_mask(m2, kept2); other.phase_a(bank)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
for rep in range(20):
    engine.train_steps(batch, meta, lab, 0, 31, 15)
    torch.cuda.synchronize()
    init = (C.c_ulonglong * 128)()
    for i_ in (31, 33):
        init[i_] = (1 << 63)                            # minima start high, maxima at 0
    h.moc_debug_stamps_set(init, 128)
    if BESIDE:
        with torch.cuda.stream(side):
            for _ in range(3):
                check(lib().moc_scores(C.byref(other.c), engine.ptr(bank.image), engine._stream()), "moc_scores")
        import time; time.sleep(0.0002)                 # (the first score pass is under way)
    engine.train_steps(batch, meta, lab, 31, 1, 15)     # the stamps of ONE step (slide 31)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 128)()
    assert h.moc_debug_stamps(buf, 128) == 0
    for a_, b_, nm in ((31, 32, "fwd: first -> last workgroup START"), (31, 30, "fwd: first start -> last workgroup END"), (30, 33, "last fwd END -> first step workgroup START"),
                       (33, 34, "step: first -> last workgroup START"), (33, 35, "step: first start -> last workgroup END")):
        acc.setdefault(("span", nm), []).append((buf[b_] - buf[a_]) / 100.0)
    t = {k: buf[k] for k in names}
    cyc = {k: buf[64 + k] for k in names}
    for a_, b_ in ((0, 2), (10, 18)):
        acc.setdefault(("MHz", a_), []).append((cyc[b_] - cyc[a_]) / max(1, (t[b_] - t[a_])) * 100.0)
    order = sorted(names, key=lambda k: t[k])
    for a_, b_ in zip(order, order[1:]):
        acc.setdefault((a_, b_), []).append((t[b_] - t[a_]) / 100.0)   # 100 MHz -> us
for (a_, b_), v in acc.items():
    v.sort()
    if a_ == "span":
        print(f"{b_:52s} median {v[len(v)//2]:6.2f} us")
    elif a_ == "MHz":
        print(f"shader clock inside kernel starting at '{names[b_]}': median {v[len(v)//2]:7.0f} MHz")
    else:
        print(f"{names[a_]:24s} -> {names[b_]:24s} median {v[len(v)//2]:6.2f} us")
