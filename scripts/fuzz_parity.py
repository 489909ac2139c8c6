"""Randomised parity sweep (not part of the test suite): random class counts, dims, storage types, ragged
slide sizes, topj / topk, discarded selectors -- two epochs of train() and one evaluation() against the
oracle, the same comparison tests/test_gpu_parity.py::test_train_and_eval_match_oracle_on_odd_shapes makes.

    python scripts/fuzz_parity.py [--cases 60] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from moc_amd import main_moc as M, synth  # noqa: E402
from oracle import moc_oracle as O  # noqa: E402


class NearTie(Exception):
    pass


def boundary_margin(x, W, We, C, j):
    """Smallest gap between the j-th and (j+1)-th key over the 2C+2 selector columns of one masked slide."""
    lg, le = x @ W, x @ We
    cols = [lg[:, c] for c in range(C)] + [c_ for c_ in torch.softmax(lg, 1).T]
    t2 = lg.topk(min(2, C), 1).values
    cols += [(t2[:, 0] - t2[:, -1]).abs(), -le[:, C:].sum(1)]
    gap = float("inf")
    for v in cols:
        s = v.sort(descending=True).values
        if j < s.numel():
            gap = min(gap, float(s[j - 1] - s[j]))
    return gap


def topk_margin(seed, D, ref_bags, labels, W, We, C, j, K, discard):
    """Smallest gap between the k-th and (k+1)-th mixed score of any class over the oracle's two epochs."""
    torch.manual_seed(seed)
    m = O.Senet(D, 4)
    o = O.make_optimizer(m)
    gap = float("inf")
    for epoch in range(2):
        torch.manual_seed(seed + 1 + epoch)
        for x, y in zip(ref_bags, labels):
            mask = O.draw_mask(x.size(0))
            sr = O.slide_process(x, W, We, C, j, mask=mask, discard=discard)
            with torch.no_grad():
                mixed = O.mix_train(m(sr["selected_feat"]), sr, discard)
            k = min(K, mixed.size(0))
            if mixed.size(0) > k:
                srt = mixed.sort(0, descending=True).values
                gap = min(gap, float((srt[k - 1] - srt[k]).min()))
            O.train_step(m, o, x, torch.as_tensor(y), W, We, C, j, K, mask, discard)
    return gap


def one_case(rng, dev, idx):
    C = int(rng.choice([2, 2, 3, 4, 5, 8, 12, 16, 20, 30, 40]))
    D = int(rng.choice([256, 512, 512, 768, 1024]))
    dtype = [torch.float32, torch.bfloat16, torch.float16][int(rng.integers(0, 3))]
    K = int(rng.choice([1, 3, 5, 10, 10, 16, 20]))
    j = int(rng.choice([5, 40, 100, 400, 3000]))
    ns = int(rng.integers(max(2, min(C, 6)), 9)) if C <= 8 else int(rng.integers(2, 5))
    big = rng.random() < 0.25                                  # some cases with enough rows for S > 4096 / > 8192
    sizes = [int(rng.integers(1, 40)) if rng.random() < 0.15 else
             int(rng.integers(6000, 14000)) if big else int(rng.integers(200, 2500)) for _ in range(ns)]
    all_sel = ["delta_softmax", "delta_diff", "bottomk"]
    discard = [s for s in all_sel if rng.random() < 0.2]
    desc = f"#{idx} C={C} D={D} {str(dtype).split('.')[-1]} K={K} j={j} sizes={sizes} discard={discard}"
    W, We = synth.make_bank(int(rng.integers(1, 1 << 30)), D, C)
    bags, labels = synth.make_slide_set(int(rng.integers(1, 1 << 30)), sizes, D, We, C)
    labels = [int(rng.integers(0, C)) for _ in sizes]
    bags = [b.to(dtype) for b in bags]
    ref_bags = [b.to(torch.float32) for b in bags]
    seed = int(rng.integers(1, 1 << 30))
    torch.manual_seed(seed)
    ref_model = O.Senet(D, 4)
    ref_opt = O.make_optimizer(ref_model)
    torch.manual_seed(seed)
    model = M.senet(D, 4).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    M.set_classifier_bank(W.to(dev), We.to(dev))
    args = H.make_args(C, j, K, discard)
    res = M.ResidentBags(bags, labels, dev)
    for epoch in range(2):
        torch.manual_seed(seed + 1 + epoch)
        ref_losses = O.train_epoch(ref_model, ref_opt, ref_bags, labels, W, We, C, j, K, discard=discard)
        torch.manual_seed(seed + 1 + epoch)
        M.train(model, res, opt, dev, args)
        got = M.train.last[0].meta_ws()[0]["loss"].cpu().numpy()
        try:
            np.testing.assert_allclose(got, np.asarray(ref_losses), atol=1e-4, err_msg=desc)
        except AssertionError:
            # A selector whose j-th and (j+1)-th keys tie (or nearly: softmax columns at 12+ classes do, exactly) has
            # no defined winner -- torch.topk's pick among equal keys is unspecified and the keys themselves differ in
            # the last bit between the CPU's and the GPU's exp.  Such a case says nothing about parity.
            torch.manual_seed(seed + 1 + epoch)
            gap = min(boundary_margin(x[O.draw_mask(x.size(0))], W, We, C, j) for x in ref_bags)
            if gap < 1e-6:
                raise NearTie(f"{desc}: selection boundary margin {gap:.1e}")
            raise
    try:
        H.assert_adam_params_close(H.flat_params(model), H.flat_params(ref_model), H.flat_state(ref_opt, "exp_avg_sq"),
                                   step=2 * len(sizes), grad_noise=1e-6, what=desc)
    except AssertionError:
        # One hidden unit whose pre-activation on a pooled row sits within rounding of zero has its ReLU open on one
        # side and shut on the other: that unit's whole W1 row (and b1 / W2 entries) then takes sign-like Adam steps
        # in one run and none in the other.  Not a parity statement either: set aside when every parameter off by
        # more than 1e-4 belongs to at most two hidden units.
        d = np.abs(np.asarray(H.flat_params(model), dtype=np.float64) - np.asarray(H.flat_params(ref_model), dtype=np.float64))
        bad = np.flatnonzero(d > 1e-4)
        HID = 64
        units = set()
        for i in bad.tolist():
            if i < HID * D: units.add(i // D)                       # W1[h, :]
            elif i < HID * D + HID: units.add(i - HID * D)          # b1[h]
            elif i < HID * D + HID + 4 * HID: units.add((i - HID * D - HID) % HID)   # W2[:, h]
            else: units.add(-1)                                     # b2: not explained by one unit
        if bad.size and -1 not in units and len(units) <= 2:
            raise NearTie(f"{desc}: hidden unit(s) {sorted(units)} at the ReLU boundary ({bad.size} parameters)")
        # A class whose K-th and (K+1)-th mixed scores nearly tie pools a different row on either side: the loss moves
        # by the gap (nothing), the gradient by a whole row.  Replay the oracle and look at the margins it had.
        gap = topk_margin(seed, D, ref_bags, labels, W, We, C, j, K, discard)
        if gap < 5e-6:
            raise NearTie(f"{desc}: top-K boundary margin {gap:.1e} in the oracle's own run")
        raise
    if len(set(labels)) == C:                                  # AUC needs every class present
        ev_ref = O.evaluation(ref_model, ref_bags, labels, W, We, C, j, K, discard=discard)
        ev = M.evaluation(model, res, dev, args)
        assert abs(ev["loss"] - ev_ref["loss"]) < 1e-4 and ev["acc"] == ev_ref["acc"] and abs(ev["auc"] - ev_ref["auc"]) < 2e-3, \
            f"{desc}: evaluation {ev} vs {ev_ref}"
    return desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(a.seed)
    bad = ties = 0
    t0 = time.time()
    for i in range(a.cases):
        try:
            desc = one_case(rng, dev, i)
            print("ok  ", desc, flush=True)
        except NearTie as e:
            ties += 1
            print("TIE ", str(e), flush=True)
        except AssertionError as e:
            bad += 1
            print("FAIL", str(e)[:1500], flush=True)
    print(f"{a.cases - bad - ties}/{a.cases - ties} cases agree with the oracle ({time.time() - t0:.0f} s); "
          f"{ties} set aside (a top-j or top-K boundary tied, or a hidden unit sat on its ReLU boundary)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
