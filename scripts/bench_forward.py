#!/usr/bin/env python3
"""Time moc_meta_forward alone on an evaluation-sized batch: python scripts/bench_forward.py [C] [slides] [rows] [dtype].
With a library built with -DMOC_FWD_DIAG (make -C moc_amd/csrc FLAGS+=-DMOC_FWD_DIAG), MOC_FWD_DIAG=bits peels parts of
the 128-row kernel off: 1 no MFMA, 2 no W1 loads, 4 no row DMA, 8 no mix."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moc_amd import engine as E, main_moc as M, synth, _lib
C = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 202
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 15000
dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[sys.argv[4] if len(sys.argv) > 4 else "bf16"]
dev = torch.device("cuda:0")
W, We = synth.make_bank(1, 512, C)
X = torch.cat([synth.make_bag_device(10 + i, rows, 512, We, C, i % C, dev, dt) for i in range(ns)])
bank = E.Bank.get(W.to(dev), We.to(dev), dt, dev)
b = E.SlideBatch(X, [rows] * ns, C, C + 4, 400, 10)
if os.environ.get("ROWS64"):
    b.c.flags = _lib.MOC_FORWARD_ROWS64
b.phase_a(bank, for_eval=True)
torch.manual_seed(0)
model = M.senet(512, 4).to(dev)
meta = E.MetaState(model)
torch.cuda.synchronize()
S = int(b.n_sel.sum())
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        E.meta_forward(b, meta, 0, ns, 15, keep_hidden=False)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    print(f"C={C} slides={ns} rows={rows} {dt} selected={S} diag={os.environ.get('MOC_FWD_DIAG', '0')} rows64={bool(os.environ.get('ROWS64'))}: "
          f"{us:.1f} us per moc_meta_forward = {S * 512 * X.element_size() / us / 1e6:.2f} TB/s of selected rows")
