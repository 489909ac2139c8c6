"""Pin oracle/moc_oracle.py to the fixtures the reference's own Python produced
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import helpers as H
from moc_amd import synth
from oracle import moc_oracle as O


def test_selectors_match_reference():
    g = H.golden("selectors")
    for cid, N, C, j, seed in g["cases"]:
        W, We, x = H.bank_and_bag(seed, N, C, cid % C)
        lge = x @ We
        # same generator + same aten ops on the same host class: bit-equal here; on a
        # different CPU the matmul may differ in the last bit, hence the stored copy
        assert np.allclose(lge.numpy(), g[f"c{cid}_logits_ext"], atol=1e-6)
        lge = torch.from_numpy(g[f"c{cid}_logits_ext"])
        lg = lge[:, :C]
        got = {"top": O.sel_top(lg, [j]), "softmax": O.sel_softmax(lg, [j]),
               "gap": O.sel_gap(lg, [j]), "lowbg": O.sel_low_background(lge, [j], C)}
        keys = H.selector_keys(lge, C)
        for name, idx in got.items():
            exp = g[f"c{cid}_{name}"]
            assert tuple(idx.shape) == exp.shape == (min(j, N), C)
            for c in range(C):
                kc = keys[name][:, c if keys[name].size(1) > 1 else 0]
                H.assert_topj_set(idx[:, c].tolist(), exp[:, c].tolist(), kc, what=f"c{cid} {name}[{c}]")
            if name in ("top", "softmax"):      # value-ordered output, exact on this host
                assert np.array_equal(idx.numpy().astype(np.int32), exp)


def test_slide_process_matches_reference():
    g = H.golden("slide_process")
    for cid, N, C, j, rm, dmask, seed in g["cases"]:
        W, We, x = H.bank_and_bag(seed, N, C, cid % C)
        mask = H.unpack_mask(g[f"c{cid}_mask"], N)
        r = O.slide_process(x, W, We, C, topj=j, mask=mask if rm else None,
                            discard=H.discard_from_mask(dmask))
        assert r["selected_index"] == g[f"c{cid}_selected_index"].tolist()
        for key, name in (("logits_top_classifier", "top"), ("logits_delta_softmax_classifier", "softmax"),
                          ("logits_delta_diff_classifier", "gap"), ("logits_bottomk_irrel_classifier", "lowbg")):
            np.testing.assert_allclose(r[key].numpy(), g[f"c{cid}_{name}"], atol=1e-6, rtol=0)


def test_draw_mask_replays_reference_stream():
    g = H.golden("slide_process")
    for cid, N, C, j, rm, dmask, seed in g["cases"]:
        if rm:
            torch.manual_seed(int(seed))
            assert torch.equal(O.draw_mask(int(N)), H.unpack_mask(g[f"c{cid}_mask"], N))


def test_pooling_matches_reference():
    g = H.golden("pooling")
    for cid, N, C, seed in g["cases"]:
        W, We, x = H.bank_and_bag(seed, N, C, 0)
        lg, lge = x @ W, x @ We
        for K in (1, 10):
            np.testing.assert_allclose(O.pool_top(lg, [K])[1][K].numpy(), g[f"c{cid}_K{K}_topj"], atol=1e-6)
            np.testing.assert_allclose(O.pool_softmax(lg, [K])[1][K].numpy(), g[f"c{cid}_K{K}_softmax"], atol=1e-6)
            np.testing.assert_allclose(O.pool_gap(lg, [K])[1][K].numpy(), g[f"c{cid}_K{K}_gap"], atol=1e-6)
            np.testing.assert_allclose(O.pool_low_background(lge, [K], C)[1][K].numpy(),
                                       g[f"c{cid}_K{K}_lowbg"], atol=1e-6)
        preds, pooled, idx = O.pool_top(lg, [10], return_indices=True)
        assert np.array_equal(idx.numpy().astype(np.int32), g[f"c{cid}_topj_idx"])
        assert np.array_equal(preds[10].numpy().astype(np.int32), g[f"c{cid}_topj_pred"])


def test_detection_mode_matches_reference():
    """detection=True of the low-background helpers (index.py:65-68, :83-84; classifier.py:146-149, :161-162)."""
    g = H.golden("detection")
    for cid, N, Ct, j, bottomk, seed in g["cases"]:
        lge = torch.from_numpy(g[f"c{cid}_logits_ext"])
        kw = {} if bottomk < 0 else {"bottomk": int(bottomk)}
        idx = O.sel_low_background(lge, [int(j)], 1, detection=True, **kw)
        assert np.array_equal(idx.numpy().astype(np.int32), g[f"c{cid}_idx"])
        preds, pooled, pidx = O.pool_low_background(lge, [1, int(j)], 1, return_indices=True, detection=True, **kw)
        assert np.array_equal(pidx.numpy().astype(np.int32), g[f"c{cid}_pool_idx"])
        np.testing.assert_allclose(pooled[1].numpy(), g[f"c{cid}_pooled_1"], atol=1e-6)
        np.testing.assert_allclose(pooled[int(j)].numpy(), g[f"c{cid}_pooled_j"], atol=1e-6)
        assert np.array_equal(preds[int(j)].numpy().astype(np.int32), g[f"c{cid}_pred_j"])


def test_train_steps_match_reference():
    g = H.golden("train")
    for cid, ns, N, C, j, K, dmask, seed in g["cases"]:
        W, We = synth.make_bank(seed, 512, C)
        bags, labels = synth.make_slide_set(seed + 100, [N] * ns, 512, We, C)
        masks = H.unpack_masks(g[f"c{cid}_masks"], [N] * ns)
        model = H.seeded_senet(O, seed)
        np.testing.assert_array_equal(H.flat_params(model), g[f"c{cid}_init"])
        opt = O.make_optimizer(model)
        discard = H.discard_from_mask(dmask)
        for s in range(ns):
            loss, pooled = O.train_step(model, opt, bags[s], torch.tensor(labels[s]), W, We, C, j, K,
                                        masks[s], discard)
            assert abs(float(loss) - g[f"c{cid}_loss"][s]) < 1e-6
            np.testing.assert_allclose(pooled.numpy()[0], g[f"c{cid}_pooled"][s], atol=1e-6)
            if s == 0:
                np.testing.assert_allclose(H.flat_grads(model), g[f"c{cid}_grad1"], atol=1e-7)
            if s in (0, ns - 1):
                H.assert_adam_params_close(H.flat_params(model), g[f"c{cid}_params_s{s}"], g[f"c{cid}_v_s{s}"],
                                           step=s + 1, grad_noise=1e-8, what=f"c{cid} step {s}")
                np.testing.assert_allclose(H.flat_state(opt, "exp_avg"), g[f"c{cid}_m_s{s}"], atol=1e-7)
                np.testing.assert_allclose(H.flat_state(opt, "exp_avg_sq"), g[f"c{cid}_v_s{s}"], atol=1e-9)


def test_evaluations_match_reference():
    g = H.golden("evaluation")
    for cid, ns, N, C, j, K, dmask, repeat_num, seed in g["cases"]:
        W, We = synth.make_bank(seed, 512, C)
        bags, labels = synth.make_slide_set(seed + 100, [N] * ns, 512, We, C)
        model = H.seeded_senet(O, seed)
        np.testing.assert_array_equal(H.flat_params(model), g[f"c{cid}_init"])
        div = int(repeat_num) or ns
        ev = O.evaluation(model, bags, labels, W, We, C, j, K, H.discard_from_mask(dmask), len_dataset=div)
        np.testing.assert_allclose([ev["loss"], ev["acc"], ev["auc"]], g[f"c{cid}_eval"], atol=1e-6)
        for name in ("topj", "delta_softmax", "delta_diff", "bottomk"):
            zs = O.zs_evaluation(bags, labels, W, We, C, K, pooling=name, len_dataset=div)
            np.testing.assert_allclose([zs["loss"], zs["acc"], zs["auc"]], g[f"c{cid}_zs_{name}"], atol=1e-6)
        for mode in ("avg", "sum", "max"):
            ab = O.ablation_evaluation(bags, labels, W, We, C, j, K, mode, len_dataset=div)
            np.testing.assert_allclose([ab["loss"], ab["acc"], ab["auc"]], g[f"c{cid}_abl_{mode}"], atol=1e-6)


def test_selector_asserts_like_reference():
    lg = torch.randn(10, 3)
    with pytest.raises(AssertionError, match="more bg classes"):
        O.sel_low_background(lg, [5], 3)
