"""Phase stamps of select_group_kernel (workgroup 0: slide 0, first group of eight columns) at the EBRAINS-30 evaluation
shape, from the diagnostic build (make -C moc_amd/csrc stamps): python scripts/diag_stamps_select.py [C] [slides] [rows]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _stamps  # noqa: E402,F401
import torch
sys.path.insert(0, ROOT)
from moc_amd import engine as E, synth
from moc_amd._lib import lib
Cc = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 202
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 15000
dev = torch.device("cuda:0")
W, We = synth.make_bank(1, 512, Cc)
X = torch.cat([synth.make_bag_device(10 + i, rows, 512, We, Cc, i % Cc, dev, torch.bfloat16) for i in range(ns)])
bank = E.Bank.get(W.to(dev), We.to(dev), torch.bfloat16, dev)
b = E.SlideBatch(X, [rows] * ns, Cc, Cc + 4, 400, 10)
b.phase_a(bank)
torch.cuda.synchronize()
h = lib(); h.moc_debug_stamps.restype = C.c_int; h.moc_debug_stamps.argtypes = [C.c_void_p, C.c_int]
names = {20: "start (arguments, row_off read)", 21: "sample keys in the pool", 22: "bounds selected (radix over 8 x 1024)",
         23: "sweep done (candidates in the pool)", 24: "top-j selected (radix over the pool)", 25: "flags marked"}
acc = {}
for rep in range(10):
    b.sel_flag.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); b.select(); e1.record()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 128)()
    assert h.moc_debug_stamps(buf, 128) == 0
    ks = sorted(names)
    for a_, b_ in zip(ks, ks[1:]):
        acc.setdefault((a_, b_), []).append((buf[b_] - buf[a_]) / 100.0)
    acc.setdefault("kernel", []).append(e0.elapsed_time(e1) * 1e3)
for k, v in acc.items():
    v = sorted(v)
    print(f"{names[k[1]] if k != 'kernel' else 'whole launch (events)':48s} {v[len(v) // 2]:8.2f} us")
