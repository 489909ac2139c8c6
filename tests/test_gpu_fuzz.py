"""The randomised parity harness under the driver's eyes: a FIXED, committed list of cases (tests/golden/fuzz_cases.json,
made by tests/golden/make_fuzz_cases.py: every case the sweeps ever set aside or failed on, plus a spread over class
counts, dims, storage types, sizes, topj, topk and discards) replayed against the oracle -- two epochs of train() and
one evaluation() each.  A case must agree, or the committed classifier (tests/fuzz_core.py) must set it aside for a
reason the ORACLE's own numbers show, printed with its margin; set-asides are counted and bounded."""
import json
import os

import pytest

import fuzz_core as F

pytestmark = pytest.mark.gpu

CASES = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_cases.json")))
CHUNK = 8


def test_list_is_what_the_verdict_asked_for():
    assert len(CASES) >= 60
    descs = [F.describe(c) for c in CASES]
    assert any("C=30" in d and "bfloat16" in d and "j=40" in d and "[10847, 2, 6303]" in d for d in descs), "seed 23 #87 is missing"
    assert {c["dtype"] for c in CASES} == {"float32", "bfloat16", "float16"}
    assert {2, 3, 12, 30, 40} <= {c["C"] for c in CASES}
    assert any(max(c["sizes"]) > 8192 and c["j"] * (2 * c["C"] + 2) > 8192 for c in CASES), "no case reaches S > 8192"


@pytest.mark.parametrize("lo", range(0, len(CASES), CHUNK))
def test_committed_cases_agree_or_are_set_aside_by_the_committed_classifier(gpu_device, lo):
    aside, compared = [], 0
    for c in CASES[lo:lo + CHUNK]:
        c = dict(c)
        r = F.run_case(c, gpu_device)              # raises on a real disagreement (gradients at 2e-6, then two epochs)
        compared += c["_grad_check"] == "compared"
        if c["_grad_check"] != "compared":
            print("GRADIENT CHECK SKIPPED", F.describe(c), "--", c["_grad_check"])
        if r != "ok":
            aside.append((F.describe(c), r[1], c.get("expect", "ok")))
            print("SET ASIDE", F.describe(c), "--", r[1])
        else:
            assert c.get("expect", "ok") in ("ok", "set aside")
    # only the cases the committed list marks as undefined (`expect: set aside`, listed first) may be set aside: a case
    # the list expects to agree and the classifier excuses anyway is a failure, not a quiet green
    known = sum(1 for c in CASES[lo:lo + CHUNK] if c.get("expect") == "set aside")
    unknown = [a for a in aside if a[2] != "set aside"]
    assert not unknown and len(aside) <= known, aside
    assert compared >= len(CASES[lo:lo + CHUNK]) - 2 - known, f"only {compared} of the chunk's cases had their gradients compared element-wise"
