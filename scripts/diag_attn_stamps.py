"""Phase stamps of gated_attention_kernel (workgroup 0) from the diagnostic build
(make -C moc_amd/csrc stamps; built into build/ by scripts/_stamps.py)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _stamps  # noqa: E402,F401  (builds build/libmoc_hip_stamps.so here if missing; sets MOC_HIP_LIB)
import torch
sys.path.insert(0, ROOT)
from moc_amd import engine
from moc_amd._lib import lib
dev = torch.device("cuda:0")
N, L, D, K = [int(v) for v in (sys.argv[1:5] + ["15000", "512", "384", "1"][len(sys.argv) - 1:])]
g = torch.Generator().manual_seed(1)
h = torch.relu(torch.randn(N, L, generator=g)).to(dev)
Wa, Wb = (torch.randn(D, L, generator=g) * 0.05).to(dev), (torch.randn(D, L, generator=g) * 0.05).to(dev)
ba, bb = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
Wc, bc = torch.randn(K, D, generator=g).to(dev), torch.zeros(K, device=dev)
hh = lib(); hh.moc_debug_stamps.restype = C.c_int; hh.moc_debug_stamps.argtypes = [C.c_void_p, C.c_int]
names = {30: "begin", 31: "prologue done", 32: "main loop done", 33: "scores done", 34: "end"}
acc = {}
for rep in range(20):
    engine.gated_attention_pool(h, Wa, ba, Wb, bb, Wc, bc)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 128)()
    assert hh.moc_debug_stamps(buf, 128) == 0
    ks = sorted(names)
    for a_, b_ in zip(ks, ks[1:]):
        acc.setdefault((a_, b_), []).append((buf[b_] - buf[a_]) / 100.0)
        acc.setdefault(("MHz", a_, b_), []).append((buf[64 + b_] - buf[64 + a_]) / max(1, buf[b_] - buf[a_]) * 100.0)
for k, v in acc.items():
    v.sort()
    if k[0] == "MHz":
        print(f"   shader clock {names[k[1]]} -> {names[k[2]]}: {v[len(v) // 2]:.0f} MHz")
    else:
        print(f"{names[k[0]]:16s} -> {names[k[1]]:16s} median {v[len(v) // 2]:7.2f} us")
