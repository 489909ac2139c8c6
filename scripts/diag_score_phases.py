"""Where one wave of the streaming score kernel (three n-tiles) spends its cycles -- diagnostic build
(make -C moc_amd/csrc stamps; built into build/ by scripts/_stamps.py): locate + issue / wait for loads / LDS reads +
MFMAs / row epilogue, summed over the units of wave 0 of workgroup 0.

    python scripts/diag_score_phases.py [classes 30] [dim 512] [slides 120] [rows 15000] [masked 1] [compact statistics 1] [bf16|fp16|fp32]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _stamps  # noqa: E402,F401  (builds build/libmoc_hip_stamps.so here if missing; sets MOC_HIP_LIB)
import torch  # noqa: E402

sys.path.insert(0, ROOT)
from moc_amd import engine, synth  # noqa: E402
from moc_amd._lib import check, lib, ptr  # noqa: E402

Cc, D, n_slides, N, masked, compact = [int(v) for v in (sys.argv[1:7] + ["30", "512", "120", "15000", "1", "1"][len(sys.argv) - 1:])]
DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[sys.argv[7] if len(sys.argv) > 7 else "bf16"]
dev = torch.device("cuda:0")
W, We = synth.make_bank(1, D, Cc)
X = torch.randn(n_slides * N, D, device=dev).to(DT)
mask = (torch.rand(n_slides * N) > 0.5) if masked else None
b = engine.SlideBatch(X, [N] * n_slides, Cc, Cc + 4, 400, 10, mask=mask)
bank = engine.Bank.get(W, We, DT, dev)
b.c.flags = 1 if compact else 0                      # MOC_STATS_COMPACT
check(lib().moc_mask_compact(C.byref(b.c), engine._stream()), "mc")
hh = lib()
hh.moc_debug_stamps.restype = C.c_int
hh.moc_debug_stamps.argtypes = [C.c_void_p, C.c_int]
names = ["locate", "wait for the unit's loads", "LDS reads + MFMAs", "row epilogue (with its stores)", "issue of the next unit's loads"]
rows = []
for rep in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib().moc_scores(C.byref(b.c), ptr(bank.image), engine._stream()), "s")
    e1.record()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 128)()
    assert hh.moc_debug_stamps(buf, 128) == 0
    rows.append(([buf[40 + q] for q in range(5)], buf[46], e0.elapsed_time(e1) * 1e3))
rows.sort(key=lambda r: r[2])
ph, n, us = rows[len(rows) // 2]
tot = sum(ph)
print(f"C={Cc} D={D} {n_slides} x {N} masked={masked} compact={compact}: launch {us:.1f} us; wave 0 of workgroup 0: {n} units, {tot / max(n, 1):.0f} cycles per unit "
      f"({tot / us / 1e3:.2f} GHz if the wave was busy start to end)")
for q in (0, 4, 1, 2, 3):
    print(f"  {names[q]:32s} {ph[q] / max(n, 1):8.0f} cycles per unit  {100.0 * ph[q] / tot:5.1f} %")
