"""After replaying one fuzz case: which parameters differ from the oracle's, and how large the oracle's second
moments are there (tiny v = the gradient was ~0 there: Adam's step is sign-like and flips on noise)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_parity as F
import helpers as H
from moc_amd import main_moc as M, synth
from oracle import moc_oracle as O
want, seed0 = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed0)
dev = torch.device("cuda:0")
grab = {}
orig = H.assert_adam_params_close
def spy(p, q, v, **k):
    grab.update(p=p, q=q, v=v, k=k)
    return orig(p, q, v, **k)
H.assert_adam_params_close = spy
F.H.assert_adam_params_close = spy
for i in range(want + 1):
    try:
        F.one_case(rng, dev, i)
    except (AssertionError, F.NearTie) as e:
        if i == want: print(str(e)[:300])
p, q, v = [np.asarray(t, dtype=np.float64) for t in (grab["p"], grab["q"], grab["v"])]
d = np.abs(p - q)
bad = np.flatnonzero(d > 1e-4)
print(f"{bad.size} parameters differ by more than 1e-4 of {p.size}; step count {grab['k'].get('step')}")
print("index ranges: W1 [0, H*D), b1, W2, b2 follow;  first/last bad:", bad[:5], bad[-5:])
print("oracle sqrt(v) at the bad ones: median %.3e  max %.3e;  over all: median %.3e" % (np.median(np.sqrt(v[bad])), np.sqrt(v[bad]).max(), np.median(np.sqrt(v))))
print("diff/lr at the bad ones: min %.2f median %.2f max %.2f" % ((d[bad] / 1e-3).min(), np.median(d[bad] / 1e-3), (d[bad] / 1e-3).max()))
