"""Replay ONE case of scripts/fuzz_parity.py (same seed, same draw order) and print what it compares."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import fuzz_parity as F
want, seed = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
dev = torch.device("cuda:0")
orig = np.testing.assert_allclose
def loud(a, b, **k):
    if want == cur[0]:
        print("   losses gpu", np.asarray(a), "\n   losses ref", np.asarray(b), "\n   |d|", np.abs(np.asarray(a) - np.asarray(b)).max())
    return orig(a, b, **k)
np.testing.assert_allclose = loud
cur = [0]
for i in range(want + 1):
    cur[0] = i
    try:
        d = F.one_case(rng, dev, i)
        if i == want: print("ok", d)
    except F.NearTie as e:
        if i == want: print("TIE ", e)
    except AssertionError as e:
        if i == want: print("FAIL", str(e)[:600])
