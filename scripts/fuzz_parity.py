"""Randomised parity sweep: fresh random cases (tests/fuzz_core.py) -- two epochs of train() and one evaluation()
against the oracle.  The committed list tests/golden/fuzz_cases.json is replayed by tests/test_gpu_fuzz.py under
`pytest -m gpu`; this script looks for new ones.

    python scripts/fuzz_parity.py [--cases 60] [--seed 1] [--dump cases.json]

--dump writes every case that was set aside or failed (explicit dicts, ready to append to the committed list).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_core as F  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--dump", default=None)
    ap.add_argument("--wide", action="store_true", help="wide banks (45-76 classes) instead of the standard mix")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(a.seed)
    bad = ties = 0
    keep = []
    t0 = time.time()
    for i in range(a.cases):
        c = (F.draw_wide_case if a.wide else F.draw_case)(rng, i)
        c["origin"] = f"{'wide' if a.wide else 'fuzz'} seed {a.seed}"
        try:
            r = F.run_case(c, dev)
            if r == "ok":
                print("ok  ", F.describe(c), flush=True)
            else:
                ties += 1
                keep.append(dict(c, expect="set aside", reason=r[1]))
                print("TIE ", F.describe(c), "--", r[1], flush=True)
        except AssertionError as e:
            bad += 1
            keep.append(dict(c, expect="FAIL", reason=str(e)[:300]))
            print("FAIL", str(e)[:1500], flush=True)
    print(f"{a.cases - bad - ties}/{a.cases - ties} cases agree with the oracle ({time.time() - t0:.0f} s); "
          f"{ties} set aside (a top-j or top-K boundary tied, or a hidden unit sat on its ReLU boundary)")
    if a.dump:
        json.dump(keep, open(a.dump, "w"), indent=1)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
