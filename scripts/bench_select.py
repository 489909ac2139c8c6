#!/usr/bin/env python3
"""Time moc_select alone on an evaluation-sized batch: python scripts/bench_select.py [C] [slides] [rows] [masked 0/1]."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moc_amd import engine as E, synth
C = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 202
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 15000
masked = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda:0")
W, We = synth.make_bank(1, 512, C)
X = torch.cat([synth.make_bag_device(10 + i, rows, 512, We, C, i % C, dev, torch.bfloat16) for i in range(ns)])
bank = E.Bank.get(W.to(dev), We.to(dev), torch.bfloat16, dev)
mask = (torch.rand(ns * rows) > 0.5).to(torch.uint8) if masked else None
b = E.SlideBatch(X, [rows] * ns, C, C + 4, 400, 10, mask=mask)
b.phase_a(bank)
torch.cuda.synchronize()
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b.sel_flag.zero_()
    e0.record()
    for _ in range(5):
        b.select()
    e1.record()
    torch.cuda.synchronize()
    print(f"C={C} slides={ns} rows={rows} masked={masked} variant={os.environ.get('MOC_SELECT_VARIANT', '0')} "
          f"per_column={os.environ.get('MOC_SELECT_KERNEL', '0')}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us per moc_select")
