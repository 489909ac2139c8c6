"""Randomised shapes for moc_gated_attention_pool and moc_gated_attention_backward against a float64 restatement and
autograd on it (not part of the test suite; tests/test_gpu_baselines.py holds the fixed shapes)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moc_amd import engine  # noqa: E402
from moc_amd.model_clam import gated_attention_pool  # noqa: E402
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
    # backward: gradient arriving at A_raw, at M, or both; every input's gradient against float64 autograd, relative to
    # the gradient's own scale (floor 1e-2: d_bc is a cancelling sum when only M is used)
    use = str(rng.choice(["A", "M", "AM"]))
    uA, uM = torch.randn(K, N, generator=g), torch.randn(K, L, generator=g)
    ops = (h, Wa, ba, Wb, bb, Wc, bc)
    ref_in = [t.double().requires_grad_(True) for t in ops]
    Ar, Mr = BO.gated_attention_pool(*ref_in)
    ref = torch.autograd.grad((Ar * uA.double()).sum() * ("A" in use) + (Mr * uM.double()).sum() * ("M" in use), ref_in)
    got_in = [t.to(dev).requires_grad_(True) for t in ops]
    Ag, Mg = gated_attention_pool(*got_in)
    got = torch.autograd.grad((Ag * uA.to(dev)).sum() * ("A" in use) + (Mg * uM.to(dev)).sum() * ("M" in use), got_in)
    eg = max(float((a.double().cpu() - b).abs().max()) / max(float(b.abs().max()), 1e-2) for a, b in zip(got, ref))
    ok = ok and eg < 1e-4
    bad += not ok
    print("ok  " if ok else "FAIL", f"N={N} L={L} D={D} K={K}  |dA|={ea:.2e} |dM|={em:.2e}  backward ({use}) {eg:.2e} of scale", flush=True)
print(f"{60 - bad}/60 shapes: forward within 2e-4, every gradient within 1e-4 of its scale")
sys.exit(1 if bad else 0)
