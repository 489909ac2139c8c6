"""Randomised shapes for moc_gated_attention_pool against a float64 restatement (not part of the test suite)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moc_amd import engine  # noqa: E402
from oracle import baselines_oracle as BO  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
dev = torch.device("cuda:0")
bad = 0
for i in range(60):
    N = int(rng.choice([1, 2, 15, 63, 64, 65, 127, 1000, int(rng.integers(1, 20000))]))
    L = 16 * int(rng.integers(4, 65))
    D = int(rng.choice([128, 256, 384]))
    K = int(rng.integers(1, 9))
    g = torch.Generator().manual_seed(int(rng.integers(1, 1 << 30)))
    h = torch.relu(torch.randn(N, L, generator=g))
    sc = (2.0 / (L + D)) ** 0.5
    Wa, Wb = torch.randn(D, L, generator=g) * sc, torch.randn(D, L, generator=g) * sc
    ba, bb = torch.randn(D, generator=g) * 0.1, torch.randn(D, generator=g) * 0.1
    Wc, bc = torch.randn(K, D, generator=g) * float(rng.choice([0.1, 0.5, 2.0])), torch.randn(K, generator=g) * 0.1
    A_ref, M_ref = BO.gated_attention_pool(*[t.double() for t in (h, Wa, ba, Wb, bb, Wc, bc)])
    A, M = engine.gated_attention_pool(*[t.to(dev) for t in (h, Wa, ba, Wb, bb, Wc, bc)])
    ea = float((A.cpu().double() - A_ref).abs().max())
    em = float((M.cpu().double() - M_ref).abs().max())
    ok = ea < 2e-4 and em < 2e-4 and bool(torch.isfinite(M).all())
    bad += not ok
    print("ok  " if ok else "FAIL", f"N={N} L={L} D={D} K={K}  |dA|={ea:.2e} |dM|={em:.2e}", flush=True)
print(f"{60 - bad}/60 shapes within 2e-4")
sys.exit(1 if bad else 0)
