"""Parity of the HIP path (through the C ABI) with the oracle and with the golden
fixtures the reference produced.  Needs an MI355X: run with -m gpu.

Tolerances: north_star states 1e-4 fp32 for outputs; index work is compared as
sets, exact except rows whose key lies within helpers.KEY_BAND of a top-j
boundary (fp32 dot products summed in a different order than MKL's differ by
~1e-7, which may swap the rows ranked j and j+1)."""
import numpy as np
import pytest
import torch

import helpers as H
from moc_amd import synth
from oracle import moc_oracle as O

pytestmark = pytest.mark.gpu

ATOL = H.ATOL
TIGHT = 2e-6      # fp32 re-association noise on O(1) cosine logits


@pytest.fixture(scope="module")
def dev(gpu_device):
    return gpu_device


def _mm():
    from moc_amd import main_moc
    return main_moc


def _engine():
    from moc_amd import engine
    return engine


# ------------------------------------------------------------------ score pass
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,C,D", [(1, 2, 512), (17, 2, 512), (1000, 3, 512), (4099, 30, 512), (700, 64, 1024), (333, 2, 256),
                                   (2500, 20, 512), (1500, 50, 512), (900, 13, 256), (1100, 30, 1024),
                                   (3001, 64, 1024), (1100, 100, 512), (530, 124, 256), (260, 80, 768)])
def test_scores_and_row_stats_match_oracle(dev, dtype, N, C, D):
    E = _engine()
    W, We = synth.make_bank(100 + N, D, C)
    x = synth.make_bag(200 + N, N, D, We, C, label=0)
    xs = x.to(dtype)                     # storage rounding is part of the INPUT
    xr = xs.to(torch.float32)
    batch = E.SlideBatch(xs.to(dev).contiguous(), [N], C, C + 4, 10, 10)
    batch.scores(E.Bank.get(W, We, dtype, dev))
    st = batch.stats.cpu()
    lg, lge = xr @ W, xr @ We
    keys = H.selector_keys(lge, C)
    np.testing.assert_allclose(st[:C].t().numpy(), lg.numpy(), atol=TIGHT, rtol=0)
    np.testing.assert_allclose(st[C:2 * C].t().numpy(), keys["softmax"].numpy(), atol=TIGHT, rtol=0)
    np.testing.assert_allclose(st[2 * C].numpy(), keys["gap"][:, 0].numpy(), atol=2 * TIGHT, rtol=0)
    np.testing.assert_allclose(st[2 * C + 1].numpy(), lge[:, C:].sum(1).numpy(), atol=4 * TIGHT, rtol=0)
    np.testing.assert_allclose(st[2 * C + 2].numpy(), lge[:, C:].max(1)[0].numpy(), atol=TIGHT, rtol=0)
    assert int(batch.sel_flag.sum()) == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_scores_many_slides_wide_bank_is_chunked_consistently(dev, dtype):
    """C = 30 keeps the whole 3-tile bank image in LDS, leaving room for ~160 slides' metadata: a
    400-slide batch goes out in chunks and must give, row for row, what one long slide gives."""
    E = _engine()
    C, D = 30, 512
    W, We = synth.make_bank(31, D, C)
    sizes = [17 + (i * 37) % 90 for i in range(400)]
    x = synth.make_bag(32, sum(sizes), D, We, C, label=3).to(dtype).to(dev).contiguous()
    bank = E.Bank.get(W, We, dtype, dev)
    many = E.SlideBatch(x, sizes, C, C + 4, 10, 10)
    many.scores(bank)
    one = E.SlideBatch(x, [sum(sizes)], C, C + 4, 10, 10)
    one.scores(bank)
    assert torch.equal(many.stats, one.stats)
    xr = x.cpu().to(torch.float32)
    np.testing.assert_allclose(many.stats[:C].t().cpu().numpy(), (xr @ W).numpy(), atol=TIGHT, rtol=0)


@pytest.mark.parametrize("C", [2, 30])
def test_scores_short_first_tiles_and_masked_slots(dev, C):
    """The streaming kernel cuts tiles at absolute multiples of 16 slots: a slide whose base is not one starts with a
    short tile (1 ... 15 rows), slides of 1 ... 17 rows may lie inside one tile of their neighbours' numbering.  Row for
    row the statistics must equal those of one long slide (every row is independent), and with a row mask slot
    base + j must hold the statistics of the j-th kept row."""
    E = _engine()
    D = 512
    W, We = synth.make_bank(41, D, C)
    sizes = [1, 7, 8, 9, 15, 16, 17, 31, 33, 3, 1, 64, 100, 2, 14, 1, 1, 47, 16, 5]
    T = sum(sizes)
    x = synth.make_bag(42, T, D, We, C, label=1).to(torch.bfloat16).to(dev).contiguous()
    bank = E.Bank.get(W, We, torch.bfloat16, dev)
    one = E.SlideBatch(x, [T], C, C + 4, 10, 10)
    one.scores(bank)
    many = E.SlideBatch(x, sizes, C, C + 4, 10, 10)
    many.scores(bank)
    assert torch.equal(many.stats, one.stats)
    g = torch.Generator().manual_seed(43)
    mask = torch.rand(T, generator=g) > 0.5
    mask[:1] = True                                              # (slides that keep no row are allowed; keep the first one non-empty)
    masked = E.SlideBatch(x, sizes, C, C + 4, 10, 10, mask=mask)
    masked.scores(bank)
    st, ref = masked.stats.cpu(), one.stats.cpu()
    nk = masked.n_kept.cpu().tolist()
    base = 0
    for i, n in enumerate(sizes):
        kept = torch.nonzero(mask[base:base + n]).flatten()
        assert nk[i] == len(kept)
        assert torch.equal(st[:, base:base + len(kept)], ref[:, base + kept]), f"slide at {base} ({n} rows, {len(kept)} kept)"
        base += n


def test_scores_timed_is_the_same_launch_with_the_kernels_own_stamps(dev):
    """moc_scores_timed (bench.py's roofline): the same statistics as moc_scores, and two events whose distance is a
    kernel's duration -- microseconds, not the milliseconds of a host round trip."""
    E = _engine()
    C, D, N = 2, 512, 20000
    W, We = synth.make_bank(51, D, C)
    x = synth.make_bag(52, N, D, We, C, label=0).to(torch.bfloat16).to(dev).contiguous()
    bank = E.Bank.get(W, We, torch.bfloat16, dev)
    a, b = E.SlideBatch(x, [N], C, C + 4, 10, 10), E.SlideBatch(x, [N], C, C + 4, 10, 10)
    a.scores(bank)
    e0, e1 = E.timed_scores(b, bank)
    torch.cuda.synchronize()
    assert torch.equal(a.stats, b.stats)
    assert 1e-3 < e0.elapsed_time(e1) < 5.0


def test_scores_uses_W_for_foreground_and_Wext_for_background(dev):
    """main_moc.py:336-337: logits come from W, only the background from W_ext."""
    E = _engine()
    W, We = synth.make_bank(5, 512, 2)
    We = We.clone()
    We[:, :2] = torch.flip(We[:, :2], dims=[1]) * 0.5          # make W_ext[:, :C] != W
    x = synth.make_bag(6, 300, 512, We, 2, label=1)
    batch = E.SlideBatch(x.to(dev), [300], 2, 6, 10, 10)
    batch.scores(E.Bank.get(W, We, torch.float32, dev))
    np.testing.assert_allclose(batch.stats[:2].t().cpu().numpy(), (x @ W).numpy(), atol=TIGHT)
    batch.scores(E.Bank.get(W, We, torch.float32, dev, fg_from_ext=True))
    np.testing.assert_allclose(batch.stats[:2].t().cpu().numpy(), (x @ We[:, :2]).numpy(), atol=TIGHT)


# ------------------------------------------------------------------ selection: exact on identical keys
def _stats_from_logits_ext(lge, C):
    k = H.selector_keys(lge, C)
    return torch.cat([lge[:, :C].t(), k["softmax"].t(), k["gap"].t(), lge[:, C:].sum(1, keepdim=True).t(),
                      lge[:, C:].max(1, keepdim=True)[0].t()], 0).contiguous()


def test_select_union_is_exact_on_reference_keys(dev):
    """Feed the selectors the reference's own logits: the union must equal the union of the
    reference's four index tensors bit for bit (integer/index work)."""
    E = _engine()
    g = H.golden("selectors")
    for cid, N, C, j, seed in g["cases"]:
        lge = torch.from_numpy(g[f"c{cid}_logits_ext"])
        x = torch.zeros(int(N), 256)
        for dbits in (0, 1, 2, 4, 8, 5, 14):
            discard = H.discard_from_mask(dbits)
            batch = E.SlideBatch(x.to(dev), [int(N)], int(C), int(C) + 4, int(j), 10, discard)
            batch.stats.copy_(_stats_from_logits_ext(lge, int(C)))
            batch.sel_flag.zero_()
            batch.select()
            batch.gather_candidates()
            S = int(batch.n_sel.item())
            got = batch.sel_idx[:S].cpu().tolist()
            exp = set()
            for name, sel in (("top", "topk"), ("softmax", "delta_softmax"), ("gap", "delta_diff"), ("lowbg", "bottomk")):
                if sel not in discard:
                    exp.update(g[f"c{cid}_{name}"].flatten().tolist())
            if got != sorted(exp):
                # the softmax keys are recomputed by this host's libm: a last-ulp difference from
                # the host that wrote the fixture may swap two rows AT the boundary, nothing else
                keys = H.selector_keys(lge, int(C))
                amb = H.ambiguous_rows(keys, int(j), band=3e-7)
                assert (set(got) ^ exp) <= amb, f"case {cid} discard {discard}: {sorted((set(got) ^ exp) - amb)[:8]}"
            assert got == sorted(got) and len(set(got)) == len(got)
            assert np.array_equal(batch.sel_row[:S].cpu().numpy(), np.asarray(got))


def test_select_ties_take_lowest_rows(dev):
    E = _engine()
    N, C = 3000, 2
    lge = torch.zeros(N, C + 4)
    lge[:, 0] = torch.arange(N).float() % 7          # heavy ties
    lge[:, 1] = -lge[:, 0]
    lge[:, 2:] = 1.0
    batch = E.SlideBatch(torch.zeros(N, 256, device=dev), [N], C, C + 4, 500, 10, ["delta_softmax", "delta_diff", "bottomk"])
    batch.stats.copy_(_stats_from_logits_ext(lge, C))
    batch.sel_flag.zero_()
    batch.select()
    batch.gather_candidates()
    got = set(batch.sel_idx[: int(batch.n_sel.item())].cpu().tolist())
    exp = set()
    for c in range(C):
        col = lge[:, c]
        order = sorted(range(N), key=lambda i: (-float(col[i]), i))[:500]
        exp.update(order)
    assert got == exp


# ------------------------------------------------------------------ slide_process vs the reference fixtures
def _check_slide_process(dev, r, x_kept, W, We, C, j, discard, exp_idx, exp_cands, band=H.KEY_BAND):
    lge = x_kept @ We
    lge = torch.cat([x_kept @ W, lge[:, C:]], 1)
    keys = H.selector_keys(lge, C)
    amb = H.ambiguous_rows({k: v for k, v in keys.items()
                            if {"top": "topk", "softmax": "delta_softmax", "gap": "delta_diff", "lowbg": "bottomk"}[k] not in discard},
                           j, band)
    got, exp = set(r["selected_index"]), set(int(i) for i in exp_idx)
    assert r["selected_index"] == sorted(got)
    assert (got ^ exp) <= amb, f"selected_index differs outside the tie band: {sorted((got ^ exp) - amb)[:10]}"
    pos_g = {v: i for i, v in enumerate(r["selected_index"])}
    pos_e = {int(v): i for i, v in enumerate(exp_idx)}
    common = sorted(got & exp)
    ig, ie = [pos_g[v] for v in common], [pos_e[v] for v in common]
    assert torch.equal(r["selected_feat"].cpu(), x_kept[r["selected_index"]])
    for key, name in (("logits_top_classifier", "top"), ("logits_delta_softmax_classifier", "softmax"),
                      ("logits_delta_diff_classifier", "gap"), ("logits_bottomk_irrel_classifier", "lowbg")):
        assert tuple(r[key].shape) == (len(got), C)
        np.testing.assert_allclose(r[key].cpu().numpy()[ig], exp_cands[name][ie], atol=1e-5, rtol=0)


def test_slide_process_matches_reference_fixtures(dev):
    M = _mm()
    g = H.golden("slide_process")
    for cid, N, C, j, rm, dmask, seed in g["cases"]:
        W, We, x = H.bank_and_bag(seed, N, C, cid % C)
        mask = H.unpack_mask(g[f"c{cid}_mask"], N)
        discard = H.discard_from_mask(dmask)
        torch.manual_seed(int(seed))                       # same CPU stream as the reference run
        r = M.slide_process(x.to(dev), W.to(dev), We.to(dev), int(C), topj=int(j), random_mask=bool(rm),
                            discard_classifiers=discard)
        cands = {n: g[f"c{cid}_{n}"] for n in ("top", "softmax", "gap", "lowbg")}
        _check_slide_process(dev, r, x[mask], W, We, int(C), int(j), discard, g[f"c{cid}_selected_index"], cands)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_slide_process_full_size_vs_oracle(dev, dtype):
    """BASELINE config 2 shape: 15k x 512, C=2, topj=400, row mask on."""
    M = _mm()
    N, C, j = 15000, 2, 400
    W, We = synth.make_bank(31, 512, C)
    x = synth.make_bag(32, N, 512, We, C, label=1).to(dtype)
    xr = x.to(torch.float32)
    torch.manual_seed(77)
    mask = O.draw_mask(N)
    torch.manual_seed(77)
    r = M.slide_process(x.to(dev), W.to(dev), We.to(dev), C, topj=j, random_mask=True)
    ref = O.slide_process(xr, W, We, C, topj=j, mask=mask)
    cands = {"top": ref["logits_top_classifier"].numpy(), "softmax": ref["logits_delta_softmax_classifier"].numpy(),
             "gap": ref["logits_delta_diff_classifier"].numpy(), "lowbg": ref["logits_bottomk_irrel_classifier"].numpy()}
    assert r["selected_feat"].dtype == dtype
    r = dict(r, selected_feat=r["selected_feat"].to(torch.float32))
    _check_slide_process(dev, r, xr[mask], W, We, C, j, [], ref["selected_index"], cands)


def test_slide_process_edge_cases(dev):
    M = _mm()
    W, We = synth.make_bank(3, 512, 2)
    x = synth.make_bag(4, 5, 512, We, 2, label=0)
    r = M.slide_process(x.to(dev), W.to(dev), We.to(dev), 2, topj=400)
    assert r["selected_index"] == [0, 1, 2, 3, 4]                    # N < topj: every row
    r = M.slide_process(x.to(dev), W.to(dev), We.to(dev), 2, topj=400,
                        discard_classifiers=["topk", "delta_softmax", "delta_diff", "bottomk"])
    assert r["selected_index"] == [] and tuple(r["logits_top_classifier"].shape) == (0, 2)
    with pytest.raises(AssertionError, match="more bg classes"):
        M.slide_process(x.to(dev), W.to(dev), W.to(dev), 2, topj=4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        M.slide_process(x, W, We, 2)


# ------------------------------------------------------------------ pooling / ranking helpers
def test_pooling_functions_match_reference_fixtures(dev):
    from moc_amd import patch_selection_classifier as P
    g = H.golden("pooling")
    for cid, N, C, seed in g["cases"]:
        W, We, x = H.bank_and_bag(seed, N, C, 0)
        lg, lge = (x @ W).to(dev), (x @ We).to(dev)
        for K in (1, 10):
            got = {"topj": P.topj_pooling(lg, [K]), "softmax": P.delta_softmax_classifier_pooling(lg, [K]),
                   "gap": P.delta_diff_classifier_pooling(lg, [K]),
                   "lowbg": P.bottomk_irrel_classifier_pooling(lge, [K], coords_list=int(C))}
            for name, (preds, pooled) in got.items():
                assert tuple(pooled[K].shape) == (1, C) and preds[K].dtype == torch.int64
                np.testing.assert_allclose(pooled[K].cpu().numpy(), g[f"c{cid}_K{K}_{name}"], atol=1e-5, rtol=0)
        preds, pooled, idx = P.topj_pooling(lg, [10], return_indices=True)
        assert np.array_equal(idx.cpu().numpy().astype(np.int32), g[f"c{cid}_topj_idx"])
        assert np.array_equal(preds[10].cpu().numpy().astype(np.int32), g[f"c{cid}_topj_pred"])


def test_index_functions_match_reference_fixtures(dev):
    from moc_amd import patch_selection_classifier_index as I
    g = H.golden("selectors")
    for cid, N, C, j, seed in g["cases"]:
        if cid % 3:      # a third of the cases is plenty here; the union test covers all
            continue
        lge = torch.from_numpy(g[f"c{cid}_logits_ext"])
        lg = lge[:, :C].contiguous()
        keys = H.selector_keys(lge, int(C))
        got = {"top": I.index_topj_classifier(lg.to(dev), [int(j)]),
               "softmax": I.index_delta_softmax_classifier(lg.to(dev), [int(j)]),
               "gap": I.index_delta_diff_classifier(lg.to(dev), [int(j)]),
               "lowbg": I.index_bottomk_irrel_classifier(lge.to(dev), [int(j)], int(C))}
        for name, idx in got.items():
            exp = g[f"c{cid}_{name}"]
            assert idx.dtype == torch.int64 and tuple(idx.shape) == exp.shape
            idx = idx.cpu()
            for c in range(int(C)):
                kc = keys[name][:, c if keys[name].size(1) > 1 else 0]
                H.assert_topj_set(idx[:, c].tolist(), exp[:, c].tolist(), kc, what=f"c{cid} {name}[{c}]")
            if name == "top":     # value order: keys non-increasing down each column
                v = torch.gather(lg, 0, idx)
                assert bool((v[1:] <= v[:-1]).all())


def test_detection_mode_matches_reference_fixtures(dev):
    """detection=True (index.py:65-68, :83-84; classifier.py:146-149, :161-162): one foreground column, the rows with the
    least background mass ranked by the foreground logit and by their largest background logit."""
    from moc_amd import patch_selection_classifier as P, patch_selection_classifier_index as I
    g = H.golden("detection")
    for cid, N, Ct, j, bottomk, seed in g["cases"]:
        lge = torch.from_numpy(g[f"c{cid}_logits_ext"]).to(dev)
        kw = {} if bottomk < 0 else {"bottomk": int(bottomk)}
        idx = I.index_bottomk_irrel_classifier(lge, [int(j)], 1, detection=True, **kw)
        assert idx.dtype == torch.int64 and tuple(idx.shape) == g[f"c{cid}_idx"].shape == (min(int(j), int(N)), 2)
        assert np.array_equal(idx.cpu().numpy().astype(np.int32), g[f"c{cid}_idx"]), cid
        preds, pooled, pidx = P.bottomk_irrel_classifier_pooling(lge, [1, int(j)], return_indices=True, coords_list=1,
                                                                 detection=True, **kw)
        assert np.array_equal(pidx.cpu().numpy().astype(np.int32), g[f"c{cid}_pool_idx"]), cid
        np.testing.assert_allclose(pooled[1].cpu().numpy(), g[f"c{cid}_pooled_1"], atol=1e-5, rtol=0)
        np.testing.assert_allclose(pooled[int(j)].cpu().numpy(), g[f"c{cid}_pooled_j"], atol=1e-5, rtol=0)
        assert np.array_equal(preds[int(j)].cpu().numpy().astype(np.int32), g[f"c{cid}_pred_j"])


def test_topk_mean_edges(dev):
    E = _engine()
    v = torch.tensor([[3.0, 1.0, 2.0, 2.0, 5.0]], device=dev)
    pooled, idx, cnt = E.topk_mean(v, v, 3, want_idx=True)
    assert abs(float(pooled[0, 0]) - (5 + 3 + 2) / 3) < 1e-6 and idx[0, 0].tolist() == [4, 0, 2] and int(cnt[0, 0]) == 3
    pooled, idx, cnt = E.topk_mean(v, v, 10, want_idx=True)          # K > N: mean over all (S < K)
    assert abs(float(pooled[0, 0]) - 13 / 5) < 1e-6 and int(cnt[0, 0]) == 5 and idx[0, 0].tolist()[:5] == [4, 0, 2, 3, 1]
    pooled = E.topk_mean(v, v, 2, smallest=True)
    assert abs(float(pooled[0, 0]) - 1.5) < 1e-6
    big = torch.randn(3, 50000, device=dev)
    pooled, idx, _ = E.topk_mean(big, big, 400, want_idx=True)
    ref = big.cpu().topk(400, dim=1)
    np.testing.assert_allclose(pooled[0].cpu().numpy(), ref[0].mean(1).numpy(), atol=1e-5)
    assert torch.equal(idx[0].cpu().long(), ref[1])


@pytest.mark.parametrize("n,K", [(70, 16), (1000, 10), (5000, 10), (16384, 16), (40000, 3), (5000, 1)])
def test_topk_mean_small_k_paths(dev, n, K):
    """K <= 16 takes the per-wave-maxima bound and an LDS candidate list: few waves with a key (n < 64 K: every
    key is a candidate), ordinary columns, heavy ties (rows in ascending order among equals) and columns of ONE
    value (the list overflows: radix path) must all give the (value desc, row asc) order."""
    E = _engine()
    g = torch.Generator().manual_seed(n * 31 + K)
    cols = torch.stack([torch.randn(n, generator=g),                       # distinct values
                        (torch.arange(n) % 5).float(),                     # heavy ties
                        torch.full((n,), 0.25),                            # one value: everything ties
                        -torch.arange(n).float()])                         # descending: the first K rows
    pooled, idx, cnt = E.topk_mean(cols.to(dev), cols.to(dev), K, want_idx=True)
    for c in range(cols.size(0)):
        order = sorted(range(n), key=lambda i: (-float(cols[c, i]), i))[:K]
        assert idx[0, c].tolist() == order, (c, idx[0, c].tolist()[:6], order[:6])
        assert abs(float(pooled[0, c]) - float(cols[c, order].double().mean())) < 1e-5 and int(cnt[0, c]) == K
    lo = E.topk_mean(cols.to(dev), cols.to(dev), K, smallest=True)
    for c in range(cols.size(0)):
        assert abs(float(lo[0, c]) - float(cols[c].sort().values[:K].double().mean())) < 1e-5


@pytest.mark.parametrize("N,j", [(2049, 400), (6000, 400), (6000, 1024), (30000, 1000), (6000, 1500)])
def test_select_candidate_path_edges(dev, N, j):
    """topj <= 1024 on more than 2048 rows runs the exact select on the keys above the topj-th largest per-thread
    maximum: columns of one value (list overflow), ties across the boundary (column order needed) and topj just
    beyond the path's limit must all give the reference's sets (ties: lowest rows first)."""
    E = _engine()
    C = 2
    g = torch.Generator().manual_seed(N + j)
    lge = torch.zeros(N, C + 4)
    lge[:, 0] = torch.randn(N, generator=g)                                # distinct values
    lge[:, 1] = (torch.arange(N) % 3).float()                              # ties across the boundary
    lge[:, 2:] = 0.5                                                       # psi_beta: one value everywhere
    batch = E.SlideBatch(torch.zeros(N, 256, device=dev), [N], C, C + 4, j, 10, ["delta_softmax", "delta_diff"])
    batch.stats.copy_(_stats_from_logits_ext(lge, C))
    batch.sel_flag.zero_()
    batch.select()
    batch.gather_candidates()
    got = set(batch.sel_idx[: int(batch.n_sel.item())].cpu().tolist())
    exp = set(range(j))                                                    # psi_beta: all equal -> the first j rows
    for c in range(C):
        exp.update(sorted(range(N), key=lambda i: (-float(lge[i, c]), i))[:j])
    assert got == exp


# ------------------------------------------------------------------ train steps
def _run_train_case(dev, bags, labels, masks, W, We, C, j, K, discard, model_seed, dtype=torch.float32):
    M, E = _mm(), _engine()
    M.set_classifier_bank(W.to(dev), We.to(dev))
    torch.manual_seed(model_seed)
    model = M.senet(512, 4).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    X, sizes = M._pack([b.to(dtype) for b in bags], dev, dtype)
    batch = E.SlideBatch(X, sizes, C, C + 4, j, K, discard, mask=torch.cat(masks))
    batch.phase_a(E.Bank.get(M.zeroshot_weights, M.zeroshot_weights_ext, dtype, dev))
    lab = torch.tensor(labels, dtype=torch.int64, device=dev)
    return M, E, model, opt, batch, lab


def test_train_steps_match_reference_fixtures(dev):
    g = H.golden("train")
    for cid, ns, N, C, j, K, dmask, seed in g["cases"]:
        ns, N, C, j, K = int(ns), int(N), int(C), int(j), int(K)
        W, We = synth.make_bank(seed, 512, C)
        bags, labels = synth.make_slide_set(seed + 100, [N] * ns, 512, We, C)
        masks = H.unpack_masks(g[f"c{cid}_masks"], [N] * ns)
        discard = H.discard_from_mask(dmask)
        M, E, model, opt, batch, lab = _run_train_case(dev, bags, labels, masks, W, We, C, j, K, discard, int(seed))
        np.testing.assert_array_equal(H.flat_params(model), g[f"c{cid}_init"])
        use = E.train_use_bits(discard)
        # gradients of the first step, without updating
        meta_g = E.MetaState(model, None, need_grads=True)
        E.train_grad(batch, meta_g, lab, 0, use)
        grads = torch.cat([t.reshape(-1) for t in meta_g.grads]).cpu().numpy()
        np.testing.assert_allclose(grads, g[f"c{cid}_grad1"], atol=2e-6, rtol=1e-4)
        # then the real thing, one step at a time so state can be compared after step 1 and ns
        meta = E.MetaState(model, opt)
        t, _ = batch.meta_ws()
        for s in range(ns):
            E.train_steps(batch, meta, lab, s, 1, use)
            if s in (0, ns - 1):
                H.assert_adam_params_close(H.flat_params(model), g[f"c{cid}_params_s{s}"], g[f"c{cid}_v_s{s}"],
                                           step=s + 1, grad_noise=1e-6, what=f"c{cid} step {s}")
                np.testing.assert_allclose(H.flat_state(opt, "exp_avg"), g[f"c{cid}_m_s{s}"], atol=1e-6 * (s + 1))
                np.testing.assert_allclose(H.flat_state(opt, "exp_avg_sq"), g[f"c{cid}_v_s{s}"], atol=1e-8)
        np.testing.assert_allclose(t["loss"].cpu().numpy(), g[f"c{cid}_loss"], atol=ATOL)
        np.testing.assert_allclose(t["pooled"].cpu().numpy(), g[f"c{cid}_pooled"], atol=ATOL)
        assert all(int(float(opt.state[p]["step"])) == ns for p in model.parameters())


def test_reference_loop_body_kept_by_the_caller_runs_on_hip(dev):
    """main_moc.py:390-410 written out by the caller around slide_process -- `model(selected_feat)`, the gated mix in
    torch, topj_pooling, F.cross_entropy, loss.backward(), optimizer.step() -- goes through senet.forward's autograd
    function (moc_meta_forward / moc_senet_backward) and the differentiable topj_pooling: gradients of the first step
    against the reference's autograd (train.npz grad1), loss and the parameters after torch's own Adam step."""
    import torch.nn.functional as F
    from moc_amd import patch_selection_classifier as P
    g = H.golden("train")
    for cid, ns, N, C, j, K, dmask, seed in g["cases"]:
        ns, N, C, j, K = int(ns), int(N), int(C), int(j), int(K)
        W, We = synth.make_bank(seed, 512, C)
        bags, labels = synth.make_slide_set(seed + 100, [N] * ns, 512, We, C)
        masks = H.unpack_masks(g[f"c{cid}_masks"], [N] * ns)
        discard = H.discard_from_mask(dmask)
        M = _mm()
        torch.manual_seed(int(seed))
        model = M.senet(512, 4).to(dev)
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
        feats = bags[0][masks[0].bool()].to(dev)                   # (the fixture's mask instead of a fresh draw)
        lbl = torch.tensor([labels[0]], device=dev)
        r = M.slide_process(feats, W.to(dev), We.to(dev), n_classes=C, topj=j, random_mask=False, discard_classifiers=discard)
        weights = model(r["selected_feat"])
        assert weights.requires_grad and tuple(weights.shape) == (r["selected_feat"].size(0), 4)
        final_logits = torch.zeros_like(r["logits_top_classifier"])
        for k, (name, key) in enumerate((("topk", "logits_top_classifier"), ("delta_softmax", "logits_delta_softmax_classifier"),
                                         ("delta_diff", "logits_delta_diff_classifier"), ("bottomk", "logits_bottomk_irrel_classifier"))):
            if name not in discard:
                final_logits = final_logits + weights[:, k].unsqueeze(1) * r[key]
        logits = P.topj_pooling(final_logits, [K])[1][K]
        loss = F.cross_entropy(logits, lbl)
        opt.zero_grad()
        loss.backward()
        grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu().numpy()
        np.testing.assert_allclose(grads, g[f"c{cid}_grad1"], atol=2e-6, rtol=1e-4)
        np.testing.assert_allclose(float(loss), g[f"c{cid}_loss"][0], atol=ATOL)
        opt.step()
        H.assert_adam_params_close(H.flat_params(model), g[f"c{cid}_params_s0"], g[f"c{cid}_v_s0"], step=1, grad_noise=1e-6,
                                   what=f"c{cid} caller loop")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        M.senet(512, 4)(torch.zeros(4, 512))


@pytest.mark.parametrize("C,K,D,dtype", [(30, 10, 512, torch.bfloat16), (64, 10, 1024, torch.float16), (2, 10, 512, torch.float32)])
def test_gradient_only_step_matches_autograd(dev, C, K, D, dtype):
    """moc_train_grad (the data-parallel step's first half: gradients out, no update) on the narrow and the
    wide one-launch kernels against autograd on the oracle."""
    M, E = _mm(), _engine()
    j, sizes = 40, [700, 900, 800]
    W, We = synth.make_bank(600 + C, D, C)
    bags, labels = synth.make_slide_set(6600 + C, sizes, D, We, C)
    bags = [b.to(dtype) for b in bags]
    torch.manual_seed(9)
    ref = O.Senet(D, 4)
    torch.manual_seed(9)
    model = M.senet(D, 4).to(dev)
    M.set_classifier_bank(W.to(dev), We.to(dev))
    X, sz = M._pack(bags, dev, dtype)
    batch = E.SlideBatch(X, sz, C, C + 4, j, K, [])
    batch.phase_a(E.Bank.get(M.zeroshot_weights, M.zeroshot_weights_ext, dtype, dev))
    lab = torch.tensor(labels, dtype=torch.int64, device=dev)
    meta = E.MetaState(model, None, need_grads=True)
    for s_i in range(len(sizes)):
        E.train_grad(batch, meta, lab, s_i, 15)
        got = torch.cat([t.reshape(-1) for t in meta.grads]).cpu().numpy()
        sr = O.slide_process(bags[s_i].to(torch.float32), W, We, C, j, mask=None)
        pooled = O.pool_top(O.mix_train(ref(sr["selected_feat"]), sr), [K])[1][K]
        loss = torch.nn.functional.cross_entropy(pooled, torch.tensor([labels[s_i]]))
        grads = torch.autograd.grad(loss, list(ref.parameters()))
        exp = torch.cat([t.reshape(-1) for t in grads]).numpy()
        np.testing.assert_allclose(got, exp, atol=2e-6, rtol=1e-4)


def test_train_function_matches_oracle_epoch(dev):
    """moc_amd.main_moc.train over a loader == the oracle's sequential loop, same init, same masks
    (drawn from the same CPU generator state)."""
    M = _mm()
    C, j, K, ns, N = 2, 400, 10, 8, 6000
    W, We = synth.make_bank(41, 512, C)
    bags, labels = synth.make_slide_set(4100, [N + 37 * i for i in range(ns)], 512, We, C)
    torch.manual_seed(9)
    ref_model = O.Senet(512, 4)
    ref_opt = O.make_optimizer(ref_model)
    torch.manual_seed(9)
    model = M.senet(512, 4).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    M.set_classifier_bank(W.to(dev), We.to(dev))
    args = H.make_args(C, j, K)
    for epoch in range(2):
        torch.manual_seed(1000 + epoch)
        ref_losses = O.train_epoch(ref_model, ref_opt, bags, labels, W, We, C, j, K)
        torch.manual_seed(1000 + epoch)
        M.train(model, H.ListLoader(bags, labels), opt, dev, args)
        batch, _ = M.train.last
        np.testing.assert_allclose(batch.meta_ws()[0]["loss"].cpu().numpy(), np.asarray(ref_losses), atol=ATOL)
    H.assert_adam_params_close(H.flat_params(model), H.flat_params(ref_model), H.flat_state(ref_opt, "exp_avg_sq"),
                               step=2 * ns, grad_noise=1e-6, what="after 2 epochs")
    ev_ref = O.evaluation(ref_model, bags, labels, W, We, C, j, K)
    ev = M.evaluation(model, H.ListLoader(bags, labels), dev, args)
    assert abs(ev["loss"] - ev_ref["loss"]) < ATOL and ev["acc"] == ev_ref["acc"] and abs(ev["auc"] - ev_ref["auc"]) < 2e-3


def test_train_on_resident_bags_with_repeats_and_bf16(dev):
    """repeat_num > real_len revisits slides (dataset_generic.py:380-393), each visit with a fresh
    mask; bf16 storage is compared with the oracle fed the same bf16-rounded values."""
    M = _mm()
    C, j, K = 3, 100, 10
    W, We = synth.make_bank(51, 512, C)
    bags, labels = synth.make_slide_set(5100, [900, 1100, 1000], 512, We, C)
    bags = [b.to(torch.bfloat16) for b in bags]
    ref_bags = [b.to(torch.float32) for b in bags]
    order = [0, 1, 2, 0, 1]
    torch.manual_seed(3)
    ref_model = O.Senet(512, 4)
    ref_opt = O.make_optimizer(ref_model)
    torch.manual_seed(3)
    model = M.senet(512, 4).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    M.set_classifier_bank(W.to(dev), We.to(dev))
    res = M.ResidentBags(bags, labels, dev, repeat_num=5)
    assert len(res) == 5 and res.real_len() == 3 and res.X.dtype == torch.bfloat16
    torch.manual_seed(123)
    ref_losses = O.train_epoch(ref_model, ref_opt, [ref_bags[k] for k in order], [labels[k] for k in order],
                               W, We, C, j, K)
    torch.manual_seed(123)
    M.train(model, res, opt, dev, H.make_args(C, j, K))
    np.testing.assert_allclose(M.train.last[0].meta_ws()[0]["loss"].cpu().numpy(), np.asarray(ref_losses), atol=ATOL)
    H.assert_adam_params_close(H.flat_params(model), H.flat_params(ref_model), H.flat_state(ref_opt, "exp_avg_sq"),
                               step=5, grad_noise=1e-6, what="resident bf16")
    ev = M.evaluation(model, res, dev, H.make_args(C, j, K))
    assert res.repeat_num == 5                      # restored like main_moc.py:499
    ev_ref = O.evaluation(ref_model, ref_bags, labels, W, We, C, j, K, len_dataset=5)
    assert abs(ev["loss"] - ev_ref["loss"]) < ATOL and ev["acc"] == ev_ref["acc"]


# ------------------------------------------------------------------ evaluation loops
def test_evaluations_match_reference_fixtures(dev):
    M = _mm()
    from moc_amd import patch_selection_classifier as P
    g = H.golden("evaluation")
    for cid, ns, N, C, j, K, dmask, repeat_num, seed in g["cases"]:
        ns, N, C, j, K = int(ns), int(N), int(C), int(j), int(K)
        W, We = synth.make_bank(seed, 512, C)
        bags, labels = synth.make_slide_set(seed + 100, [N] * ns, 512, We, C)
        torch.manual_seed(int(seed))
        model = M.senet(512, 4).to(dev)
        np.testing.assert_array_equal(H.flat_params(model), g[f"c{cid}_init"])
        M.set_classifier_bank(W.to(dev), We.to(dev))
        args = H.make_args(C, j, K, H.discard_from_mask(dmask))
        rn = int(repeat_num) or None

        def close(got, exp, what):
            assert abs(got["loss"] - exp[0]) < ATOL, (what, got, exp)
            assert abs(got["acc"] - exp[1]) < 1e-12, (what, got, exp)
            assert abs(got["auc"] - exp[2]) < 2e-3, (what, got, exp)     # north_star: AUC within +-0.002

        loader = H.ListLoader(bags, labels, rn)
        close(M.evaluation(model, loader, dev, args), g[f"c{cid}_eval"], "evaluation")
        assert loader.dataset.repeat_num == (rn or ns)     # evaluation() leaves len(dataset) there (main_moc.py:470,:499)
        for name, fn in (("topj", P.topj_pooling), ("delta_softmax", P.delta_softmax_classifier_pooling),
                         ("delta_diff", P.delta_diff_classifier_pooling), ("bottomk", P.bottomk_irrel_classifier_pooling)):
            loader = H.ListLoader(bags, labels, rn)
            close(M.zs_evaluation(loader, dev, args, pooling_func=fn), g[f"c{cid}_zs_{name}"], "zs " + name)
            assert loader.dataset.repeat_num == rn
        # any OTHER callable is called as the reference calls it (main_moc.py:431-432): on feats @ zeroshot_weights_ext
        # with coords_list=n_classes -- here a wrapper around the bottom-k pooling, which must land on the same fixture
        def custom(logits_ext, topj, **kw):
            assert logits_ext.is_cuda and logits_ext.shape == (N, C + 4) and kw == {"coords_list": C}
            return P.bottomk_irrel_classifier_pooling(logits_ext, topj, **kw)
        loader = H.ListLoader(bags, labels, rn)
        close(M.zs_evaluation(loader, dev, args, pooling_func=custom), g[f"c{cid}_zs_bottomk"], "zs custom callable")
        assert loader.dataset.repeat_num == rn
        for mode in ("avg", "sum", "max"):
            args.ablation_study = mode
            close(M.ablation_evaluation(H.ListLoader(bags, labels, rn), dev, args), g[f"c{cid}_abl_{mode}"], "ablation " + mode)


def test_eval_quirk_delta_bottomk_string(dev):
    """main_moc.py:491 tests for "delta_bottomk": discarding "bottomk" removes psi_beta from the
    SELECTION but its term still enters the eval mix."""
    E = _engine()
    assert E.eval_use_bits(["bottomk"]) == 15 and E.eval_use_bits(["delta_bottomk"]) == 7
    assert E.eval_use_bits(["topk", "delta_diff"]) == 1 | 2 | 8 and E.train_use_bits(["topk", "delta_diff"]) == 2 | 8


# ------------------------------------------------------------------ size-independent properties at full size
def test_full_size_properties(dev):
    """32 slides x ~15k rows (BASELINE config 2): row permutation leaves every slide's pooled
    logits unchanged; batching does not change per-slide results; n_sel respects its bound."""
    M, E = _mm(), _engine()
    C, j, K = 2, 400, 10
    W, We = synth.make_bank(61, 512, C)
    Wd, Wed = W.to(dev), We.to(dev)
    sizes = synth.bag_sizes(5, 32, 15000, fixed=False)
    bags = [synth.make_bag_device(6100 + i, n, 512, We, C, i % C, dev) for i, n in enumerate(sizes)]
    labels = [i % C for i in range(32)]
    torch.manual_seed(0)
    model = M.senet(512, 4).to(dev)
    M.set_classifier_bank(Wd, Wed)
    args = H.make_args(C, j, K)
    X, _ = M._pack(bags, dev, torch.float32)
    lab = torch.tensor(labels, device=dev)
    meta = E.MetaState(model)

    def run(Xp, szs):
        b = E.SlideBatch(Xp, szs, C, C + 4, j, K)
        b.phase_a(E.Bank.get(Wd, Wed, torch.float32, dev))
        E.meta_forward(b, meta, 0, len(szs), 15)
        E.pool_loss(b, lab[: len(szs)], 0, len(szs))
        return b, b.meta_ws()[0]["pooled"].clone()

    b_all, pooled_all = run(X, sizes)
    ns = b_all.n_sel.cpu()
    assert int(ns.max()) <= j * (2 * C + 2) and int(ns.min()) >= j
    # (1) one slide alone == the same slide inside the batch
    _, p3 = run(bags[3].contiguous(), [sizes[3]])
    assert torch.equal(p3[0], pooled_all[3])
    # (2) permuting a slide's rows permutes selected_index, nothing else
    perm = torch.randperm(sizes[3], device=dev)
    bp, pp = run(bags[3][perm].contiguous(), [sizes[3]])
    np.testing.assert_allclose(pp[0].cpu().numpy(), pooled_all[3].cpu().numpy(), atol=1e-6)
    S = int(bp.n_sel[0])
    o = b_all.row_off_host[3]
    orig = set(b_all.sel_idx[o:o + int(ns[3])].cpu().tolist())
    assert set(perm[bp.sel_idx[:S].long()].cpu().tolist()) == orig
    # (3) evaluation over the 32 slides agrees with the oracle on the same bytes
    cpu_bags = [b.cpu() for b in bags]
    torch.manual_seed(0)
    ref_model = O.Senet(512, 4)
    ev_ref, ref_pooled = O.evaluation(ref_model, cpu_bags, labels, W, We, C, j, K, return_logits=True)
    np.testing.assert_allclose(pooled_all.cpu().numpy(), ref_pooled.numpy(), atol=ATOL)
    ev = M.evaluation(model, H.ListLoader(cpu_bags, labels), dev, args)
    assert abs(ev["loss"] - ev_ref["loss"]) < ATOL and ev["acc"] == ev_ref["acc"] and abs(ev["auc"] - ev_ref["auc"]) < 2e-3


@pytest.mark.parametrize("rows64", [False, True])
@pytest.mark.parametrize("dtype,D,C", [(torch.bfloat16, 512, 3), (torch.float16, 512, 3), (torch.bfloat16, 768, 3), (torch.float16, 256, 3),
                                       (torch.bfloat16, 512, 30), (torch.bfloat16, 1024, 20), (torch.float32, 512, 3), (torch.float32, 256, 30)])
def test_batched_forward_is_bit_identical_to_one_slide_at_a_time(dev, dtype, D, C, rows64):
    """Many slides at once take the 128-row forward kernel (rows by LDS-DMA, W1 fragments shared by eight row tiles) or,
    with MOC_FORWARD_ROWS64 / fewer selectable rows, the 64-row one (fragments shared by four); the meta-step's one-slide
    kernel must give the same bits: hidden layer, gates and mixed scores."""
    M, E = _mm(), _engine()
    from moc_amd import _lib
    j, K = 150, 10
    W, We = synth.make_bank(71, D, C)
    Wd, Wed = W.to(dev), We.to(dev)
    sizes = [900, 1500, 64, 2100, 333, 1207, 2600]
    bags = [synth.make_bag_device(7100 + i, n, D, We, C, i % C, dev, dtype) for i, n in enumerate(sizes)]
    torch.manual_seed(3)
    model = M.senet(D, 4).to(dev)
    X, _ = M._pack(bags, dev, dtype)
    b = E.SlideBatch(X, sizes, C, C + 4, j, K)
    b.c.flags = _lib.MOC_FORWARD_ROWS64 if rows64 else 0
    b.phase_a(E.Bank.get(Wd, Wed, dtype, dev))
    assert bool(b.c.flags & _lib.MOC_FORWARD_ROWS64) == rows64
    meta = E.MetaState(model)
    t = b.meta_ws()[0]
    E.meta_forward(b, meta, 0, len(sizes), 15)
    torch.cuda.synchronize()
    together = {k: t[k].clone() for k in ("mixed", "H1", "gates")}
    for k in together:
        t[k].zero_()
    for i in range(len(sizes)):
        E.meta_forward(b, meta, i, 1, 15)
    torch.cuda.synchronize()
    ns = b.n_sel.cpu().tolist()
    for i, n in enumerate(ns):
        o = b.row_off_host[i]
        assert torch.equal(together["mixed"][:, o:o + n], t["mixed"][:, o:o + n])
        assert torch.equal(together["H1"][o:o + n], t["H1"][o:o + n])
        assert torch.equal(together["gates"][o:o + n], t["gates"][o:o + n])


@pytest.mark.parametrize("dtype,D,C", [(torch.float16, 1024, 30), (torch.bfloat16, 512, 40)])
def test_one_big_slide_takes_the_64_row_forward_with_the_16_row_bits(dev, dtype, D, C):
    """ONE slide with 16,384 or more selectable rows (the training forward of the 64-way x 50 k shape) goes through the
    64-row forward kernel; with MOC_FORWARD_ROWS16 it stays on the sixteen-row one: the same hidden layer, gates and
    mixed scores, bit for bit (main_moc.py:390-403)."""
    M, E = _mm(), _engine()
    from moc_amd import _lib
    j, K = 400, 10
    W, We = synth.make_bank(73, D, C)
    Wd, Wed = W.to(dev), We.to(dev)
    sizes = [21000, 500]
    bags = [synth.make_bag_device(7300 + i, n, D, We, C, i % C, dev, dtype) for i, n in enumerate(sizes)]
    torch.manual_seed(5)
    model = M.senet(D, 4).to(dev)
    X, _ = M._pack(bags, dev, dtype)
    out = {}
    for rows16 in (False, True):
        b = E.SlideBatch(X, sizes, C, C + 4, j, K)
        b.c.flags = _lib.MOC_FORWARD_ROWS16 if rows16 else 0
        b.phase_a(E.Bank.get(Wd, Wed, dtype, dev))
        assert bool(b.c.flags & _lib.MOC_FORWARD_ROWS16) == rows16
        assert min(j * (2 * C + 2), b.c.max_rows) >= 16384           # (the bound the launcher looks at)
        meta = E.MetaState(model)
        t = b.meta_ws()[0]
        for k in ("mixed", "H1", "gates"):
            t[k].zero_()
        E.meta_forward(b, meta, 0, 1, 15)
        E.meta_forward(b, meta, 1, 1, 15)
        torch.cuda.synchronize()
        n0 = int(b.n_sel.cpu()[0])
        assert n0 > 4096
        out[rows16] = ({k: t[k].clone() for k in ("mixed", "H1", "gates")}, b.n_sel.cpu().tolist(), list(b.row_off_host[:2]))
    (a, ns, off), (c, ns2, off2) = out[False], out[True]
    assert ns == ns2 and off == off2
    for i, n in enumerate(ns):
        o = off[i]
        assert torch.equal(a["mixed"][:, o:o + n], c["mixed"][:, o:o + n])
        assert torch.equal(a["H1"][o:o + n], c["H1"][o:o + n])
        assert torch.equal(a["gates"][o:o + n], c["gates"][o:o + n])


@pytest.mark.parametrize("D,C,j,for_eval", [(512, 2, 400, False), (256, 3, 150, False), (768, 5, 90, False), (1024, 30, 40, False),
                                            (512, 64, 20, False), (512, 30, 60, True), (256, 6, 100, True)])
def test_fp32_forward_split_over_four_wave_groups_gives_the_four_wave_bits(dev, D, C, j, for_eval):
    """fp32 bags: the one-slide forward of the training step splits the columns over four wave groups (sixteen waves, every
    row requested whole at once); with MOC_FORWARD_FOUR_WAVES it is the four-wave kernel, one chain per wave over all the
    columns, folded into the same ((p0 + p1) + p2) + p3: hidden layer, gates and mixed scores bit for bit, one slide at a
    time and several per launch, short last tiles included; for_eval: the variant that reads the candidate scores from the
    score pass's statistics (MOC_CAND_FROM_STATS)."""
    M, E = _mm(), _engine()
    from moc_amd import _lib
    K = 10
    W, We = synth.make_bank(72, D, C)
    sizes = [900, 1500, 64, 17, 2100, 333]
    bags = [synth.make_bag_device(7200 + i, n, D, We, C, i % C, dev, torch.float32) for i, n in enumerate(sizes)]
    torch.manual_seed(4)
    model = M.senet(D, 4).to(dev)
    X, _ = M._pack(bags, dev, torch.float32)
    b = E.SlideBatch(X, sizes, C, C + 4, j, K)
    b.phase_a(E.Bank.get(W.to(dev), We.to(dev), torch.float32, dev), for_eval=for_eval)
    assert bool(b.c.flags & _lib.MOC_CAND_FROM_STATS) == for_eval       # (evaluation of a wide bank: candidate scores read from the statistics)
    meta = E.MetaState(model)
    t = b.meta_ws()[0]
    out = {}
    for four in (False, True):
        b.c.flags = (b.c.flags & ~_lib.MOC_FORWARD_FOUR_WAVES) | (_lib.MOC_FORWARD_FOUR_WAVES if four else 0)
        for k in ("mixed", "H1", "gates"):
            t[k].zero_()
        for i in range(len(sizes)):
            E.meta_forward(b, meta, i, 1, 15)
        torch.cuda.synchronize()
        out[four] = {k: t[k].clone() for k in ("mixed", "H1", "gates")}
        for k in ("mixed", "H1", "gates"):
            t[k].zero_()
        E.meta_forward(b, meta, 1, 3, 15)          # three slides per launch: still the 16-row kernels
        torch.cuda.synchronize()
        out[four, 3] = {k: t[k].clone() for k in ("mixed", "H1", "gates")}
    assert float(out[False]["H1"].abs().sum()) > 0
    for k in ("mixed", "H1", "gates"):
        assert torch.equal(out[False][k], out[True][k]), k
        assert torch.equal(out[False, 3][k], out[True, 3][k]), k


# ------------------------------------------------------------------ paths the fixtures do not reach
@pytest.mark.parametrize("C,K,j,D,dtype,sizes", [
    (2, 20, 100, 512, torch.float32, [900, 700, 1100]),        # K > 16: general pooling + step kernels
    (2, 1, 50, 512, torch.bfloat16, [600, 800]),               # K = 1
    (3, 10, 5000, 512, torch.float32, [300, 420, 380]),        # topj > N': every kept row selected
    (2, 10, 60, 768, torch.bfloat16, [500, 650]),              # 1536-byte rows: the 512-B-unit kernels
    (2, 10, 400, 512, torch.float32, [12, 30, 9]),             # S < K on some slides (mean over S rows)
    (5, 13, 40, 256, torch.float32, [400, 300, 350, 500, 450]),
    (2, 10, 400, 512, torch.float16, [3000, 2500, 2800, 3300]),  # fp16 storage, the one-launch step
    (3, 10, 100, 1024, torch.float16, [700, 900, 800]),          # fp16, 2-KiB rows
    (12, 10, 60, 512, torch.float16, [900, 1000, 800, 950] * 3), # fp16, C*K = 120 pairs: general step kernels, 2-tile bank
    (30, 10, 50, 512, torch.bfloat16, [700] * 30),               # EBRAINS-30 shape: 3-tile bank, 300 pairs
    (64, 10, 30, 1024, torch.float16, [900] * 64),               # the 64-way x 1024-d shape: wide step kernel, 640 pairs
    (40, 16, 40, 512, torch.float32, [800] * 40),                # fp32 storage, 640 pairs of 16 per class
    (20, 7, 400, 512, torch.bfloat16, [9000, 8000] * 10),        # S ~ 6-8 k selected rows (8 scores per thread)
    (2, 10, 400, 512, torch.bfloat16, [60000, 2000, 33]),        # the largest bag of the synthetic recipe (clipped at 60 k) beside small ones
    (3, 10, 10, 512, torch.float32, [60000, 17]),                # ... fp32 storage, the reference's default topj
])
def test_train_and_eval_match_oracle_on_odd_shapes(dev, C, K, j, D, dtype, sizes):
    M = _mm()
    W, We = synth.make_bank(900 + C * K, D, C)
    bags, labels = synth.make_slide_set(9100 + K, sizes, D, We, C)
    bags = [b.to(dtype) for b in bags]
    ref_bags = [b.to(torch.float32) for b in bags]
    torch.manual_seed(4)
    ref_model = O.Senet(D, 4)
    ref_opt = O.make_optimizer(ref_model)
    torch.manual_seed(4)
    model = M.senet(D, 4).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    M.set_classifier_bank(W.to(dev), We.to(dev))
    args = H.make_args(C, j, K)
    res = M.ResidentBags(bags, labels, dev)
    for epoch in range(2):
        torch.manual_seed(300 + epoch)
        ref_losses = O.train_epoch(ref_model, ref_opt, ref_bags, labels, W, We, C, j, K)
        torch.manual_seed(300 + epoch)
        M.train(model, res, opt, dev, args)
        got = M.train.last[0].meta_ws()[0]["loss"].cpu().numpy()
        np.testing.assert_allclose(got, np.asarray(ref_losses), atol=ATOL)
    H.assert_adam_params_close(H.flat_params(model), H.flat_params(ref_model), H.flat_state(ref_opt, "exp_avg_sq"),
                               step=2 * len(sizes), grad_noise=1e-6, what=f"C={C} K={K} D={D}")
    if len(set(labels)) == C:
        ev_ref = O.evaluation(ref_model, ref_bags, labels, W, We, C, j, K)
        ev = M.evaluation(model, res, dev, args)
        assert abs(ev["loss"] - ev_ref["loss"]) < ATOL and ev["acc"] == ev_ref["acc"] and abs(ev["auc"] - ev_ref["auc"]) < 2e-3


# ------------------------------------------------------------------ BASELINE.json configs 3 / 4 / 5 at their stated sizes
@pytest.mark.parametrize("name,C,K,j,D,dtype,sizes", [
    ("cfg 3, RCC 3-way", 3, 10, 400, 512, torch.bfloat16, [15000, 14000, 16000]),
    # topj (2C + 2) = 24,800 > every slide: the union is every kept row -- ~7.5 k to 11 k in train (mask), up to
    # 22,000 in evaluation: beyond 8,192 the pooling is topk_mean_kernel's and the wide step kernel picks it up
    ("cfg 4, EBRAINS-30 30-way", 30, 10, 400, 512, torch.bfloat16, [15000, 22000, 15000, 19000]),
    ("cfg 5, 64-way x 1024-d fp16", 64, 10, 400, 1024, torch.float16, [50000, 50000]),
])
def test_full_size_wide_configs_match_oracle(dev, name, C, K, j, D, dtype, sizes):
    """One epoch of train() and one evaluation pass at the sizes BASELINE.json quotes for its wide configurations,
    against the oracle: per-step losses 1e-4, parameters within the Adam noise bound, pooled evaluation logits 1e-4
    (the AUC needs every class among the slides, so the comparison is on what the AUC is computed from)."""
    M = _mm()
    W, We = synth.make_bank(4400 + C, D, C)
    bags, labels = synth.make_slide_set(44000 + C, sizes, D, We, C)
    labels = [(7 * i + 1) % C for i in range(len(sizes))]
    bags = [b.to(dtype) for b in bags]
    ref_bags = [b.to(torch.float32) for b in bags]
    torch.manual_seed(9)
    ref_model = O.Senet(D, 4)
    ref_opt = O.make_optimizer(ref_model)
    torch.manual_seed(9)
    model = M.senet(D, 4).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    M.set_classifier_bank(W.to(dev), We.to(dev))
    args = H.make_args(C, j, K)
    res = M.ResidentBags(bags, labels, dev)
    torch.manual_seed(77)
    ref_losses = O.train_epoch(ref_model, ref_opt, ref_bags, labels, W, We, C, j, K)
    torch.manual_seed(77)
    M.train(model, res, opt, dev, args)
    got = M.train.last[0].meta_ws()[0]["loss"].cpu().numpy()
    n_sel = M.train.last[0].n_sel.cpu().numpy()
    if C >= 30:
        assert n_sel.max() > 8192, "this case is here to reach the unions beyond 8,192 rows"
    np.testing.assert_allclose(got, np.asarray(ref_losses), atol=ATOL, err_msg=name)
    H.assert_adam_params_close(H.flat_params(model), H.flat_params(ref_model), H.flat_state(ref_opt, "exp_avg_sq"),
                               step=len(sizes), grad_noise=1e-6, what=name)
    # evaluation: the per-slide pooled logits and losses (main_moc.py:472-498), full bags, no mask
    ref_model.eval()
    ref_pooled, ref_loss = [], []
    with torch.no_grad():
        for x, y in zip(ref_bags, labels):
            sr = O.slide_process(x, W, We, C, j)
            pooled = O.pool_top(O.mix_eval(ref_model(sr["selected_feat"]), sr), [K])[1][K]
            ref_pooled.append(pooled)
            ref_loss.append(float(torch.nn.functional.cross_entropy(pooled, torch.as_tensor(labels[len(ref_loss)]).view(1))))
    model.eval()
    with torch.no_grad():
        pooled, _, losses = M._eval_pass(res, dev, args, "eval", model=model)
    np.testing.assert_allclose(pooled.numpy(), torch.cat(ref_pooled, 0).numpy(), atol=ATOL, err_msg=name)
    np.testing.assert_allclose(losses, ref_loss, atol=ATOL, err_msg=name)


# ------------------------------------------------------------------ round 3: compact statistics, grouped selector
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C,D,sizes,j", [(13, 512, [3000, 2500, 70], 100), (30, 512, [5000, 4100, 2049, 300], 400),
                                        (30, 512, [2600, 2300], 1024), (64, 1024, [3000, 2200], 400), (20, 256, [2500], 50)])
def test_compact_statistics_and_grouped_selector_give_the_same_bits(dev, dtype, C, D, sizes, j):
    """Wide banks write C + 5 statistics per row and let the selector / candidate gather re-form the softmax columns
    (include/moc_hip.h MOC_STATS_COMPACT), and take one workgroup per slide and group of eight columns (select_group_kernel)
    instead of one per column.  Neither may change a bit: union flags, selected_index, n_sel and all 2C + 2 candidate
    columns are compared with the full layout + per-column selector, masked and unmasked, incl. discarded selectors."""
    E = _engine()
    from moc_amd import _lib
    if dtype == torch.float32 and C == 64:
        pytest.skip("fp32 storage beyond four n-tiles takes the generic kernel (covered by its own shapes)")
    W, We = synth.make_bank(333 + C, D, C)
    bags, _ = synth.make_slide_set(4400 + C, sizes, D, We, C)
    X = torch.cat(bags).to(dev).to(dtype).contiguous()
    bank = E.Bank.get(W.to(dev), We.to(dev), dtype, dev)
    g = torch.Generator().manual_seed(5)
    mask = (torch.rand(sum(sizes), generator=g) > 0.5).to(torch.uint8)
    for discard, m in (((), None), ((), mask), (("delta_softmax",), mask), (("topk", "bottomk"), None)):
        outs = []
        for compact, per_column in ((False, True), (True, False), (True, True), (False, False)):
            b = E.SlideBatch(X, sizes, C, C + 4, j, 10, discard, mask=m)
            b.c.flags = _lib.MOC_SELECT_PER_COLUMN if per_column else 0
            keep = E.COMPACT_STATS
            E.COMPACT_STATS = compact
            try:
                b.phase_a(bank)
            finally:
                E.COMPACT_STATS = keep
            assert bool(b.c.flags & _lib.MOC_STATS_COMPACT) == compact
            torch.cuda.synchronize()
            outs.append((b.sel_flag.clone(), b.n_sel.clone(), b.sel_idx.clone(), b.cand.clone(), b.n_kept.clone() if m is not None else None))
        ref = outs[0]
        off = [0]
        for n in sizes:
            off.append(off[-1] + n)
        for o in outs[1:]:
            assert torch.equal(o[1], ref[1]), "n_sel differs"
            for s_i in range(len(sizes)):
                nk = int(ref[4][s_i]) if m is not None else sizes[s_i]
                S = int(ref[1][s_i])
                lo = off[s_i]
                assert torch.equal(o[0][lo:lo + nk], ref[0][lo:lo + nk]), f"union flags of slide {s_i} differ"
                assert torch.equal(o[2][lo:lo + S], ref[2][lo:lo + S]), f"selected_index of slide {s_i} differs"
                assert torch.equal(o[3][:, lo:lo + S], ref[3][:, lo:lo + S]), f"candidate scores of slide {s_i} differ"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,sizes,j,discard", [(30, [3000, 2600, 2200, 500, 2049], 400, ()), (13, [2500, 2300, 2400, 2100], 100, ("delta_diff",)),
                                              (5, [1500, 1400, 1300, 1200], 50, ())])
def test_evaluation_reads_candidates_from_the_statistics_with_the_same_bits(dev, dtype, C, sizes, j, discard):
    """Evaluation passes of wide banks never materialise the [2C+2, S] candidate columns (MOC_CAND_FROM_STATS): the forward
    reads a selected row's scores from the score pass's statistics through sel_idx.  The pooled logits, losses and
    predictions must be the bits of the pass that gathers the candidates first -- in both forward kernels (the 64-row
    one for launches over >= 4 slides of 16-bit bags, the 16-row one otherwise)."""
    M, E = _mm(), _engine()
    W, We = synth.make_bank(70 + C, 512, C)
    bags, labels = synth.make_slide_set(7000 + C, sizes, 512, We, C)
    M.set_classifier_bank(W.to(dev), We.to(dev))
    torch.manual_seed(4)
    model = M.senet(512, 4).to(dev)
    args = H.make_args(C, j, 10, discard)
    outs = []
    keep = E.CAND_FROM_STATS
    try:
        for on in (False, True):
            E.CAND_FROM_STATS = on
            res = M.ResidentBags(bags, labels, dev, dtype=dtype)
            pooled, _, losses = M._eval_pass(res, dev, args, "eval", model=model)
            batch = res.eval_plan(C, C + 4, j, 10, list(discard))["batch"]
            from moc_amd import _lib
            assert bool(batch.c.flags & _lib.MOC_CAND_FROM_STATS) == (on and C > 4)
            outs.append((pooled.clone(), list(losses)))
    finally:
        E.CAND_FROM_STATS = keep
    assert torch.equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]
    assert torch.isfinite(outs[0][0]).all()


def test_cu_census_and_reserved_table(dev):
    """moc_cu_census finds every compute unit of the device (MI355X: 8 XCDs x 32), and the reserved table is an equal
    share of every XCD."""
    E = _engine()
    slots = E.cu_slots(dev)
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    assert len(slots) == n_cu, (len(slots), n_cu)
    xccs = sorted({x for x, _ in slots})
    assert len(xccs) == 8 and all(sum(1 for x, _ in slots if x == k) == n_cu // 8 for k in xccs)
    for n in (8, 32, 64):
        t = E.reserved_cus(dev, n).cpu()
        bits = [(w, b) for w in range(128) for b in range(32) if (int(t[w]) >> b) & 1]
        assert len(bits) == n
        chosen = {((w * 32 + b) >> 8, (w * 32 + b) & 255) for w, b in bits}
        assert chosen <= set(slots)
        assert all(sum(1 for x, _ in chosen if x == k) == n // 8 for k in xccs)
    assert E.reserved_cus(dev, 0) is None


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C,D,sizes", [(2, 512, [15000, 9000, 12001, 7]), (3, 512, [5000] * 6 + [33]), (30, 512, [6000, 4100, 300]),
                                       (2, 256, [4000, 3000]), (40, 512, [3000, 2200])])
def test_ticketed_score_pass_and_reserved_cus_give_the_same_bits(dev, dtype, C, D, sizes):
    """The streaming score pass hands its tiles out by ticket (moc_batch_t.tile_ticket) and may stay off a set of compute
    units (cu_reserved: its workgroups there end at once).  Which wave computes a tile changes nothing in the tile:
    statistics and union flags are compared bit for bit with the static walk, masked and unmasked, twice in a row on
    the same batch (the counter is cleared by every launch)."""
    E = _engine()
    if dtype != torch.float32 and C == 40:
        pytest.skip("16-bit storage beyond three n-tiles takes the K-split kernels (no persistent workgroups)")
    W, We = synth.make_bank(77 + C, D, C)
    bags, _ = synth.make_slide_set(910 + C, sizes, D, We, C)
    X = torch.cat(bags).to(dev).to(dtype).contiguous()
    bank = E.Bank.get(W.to(dev), We.to(dev), dtype, dev)
    g = torch.Generator().manual_seed(11)
    mask = (torch.rand(sum(sizes), generator=g) > 0.5).to(torch.uint8)
    for m in (None, mask):
        outs = []
        for ticket, reserve in ((False, 0), (True, 0), (True, 32), (True, 120)):
            b = E.SlideBatch(X, sizes, C, C + 4, 100, 10, (), mask=m)
            b.reserve_cus(reserve, ticket=ticket)
            assert (b.c.cu_reserved is not None) == (reserve > 0) and (b.c.tile_ticket is not None) == ticket
            for _ in range(2):
                b.stats.fill_(float("nan"))
                b.phase_a(bank)
            torch.cuda.synchronize()
            outs.append((b.stats.clone(), b.sel_flag.clone(), b.n_sel.clone(), b.sel_idx.clone(),
                         b.n_kept.clone() if m is not None else None, int(b.c.flags)))
        ref = outs[0]
        off = [0]
        for n in sizes:
            off.append(off[-1] + n)
        rows = (C + 5) if (ref[5] & 1) else (2 * C + 3)
        for o in outs[1:]:
            assert torch.equal(o[2], ref[2]), "n_sel differs"
            for s_i in range(len(sizes)):
                nk = int(ref[4][s_i]) if m is not None else sizes[s_i]
                lo = off[s_i]
                assert torch.equal(o[0][:rows, lo:lo + nk], ref[0][:rows, lo:lo + nk]), f"statistics of slide {s_i} differ"
                assert torch.equal(o[1][lo:lo + nk], ref[1][lo:lo + nk]), f"union flags of slide {s_i} differ"
                S = int(ref[2][s_i])
                assert torch.equal(o[3][lo:lo + S], ref[3][lo:lo + S]), f"selected_index of slide {s_i} differs"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_ticketed_score_pass_with_every_compute_unit_reserved(dev, dtype):
    """The table names EVERY compute unit: all workgroups but the first eight of the launch end at once, and those eight
    (which never leave) draw from eight of the sixty-four counters -- the tiles of the other fifty-six reach them only
    through the sweep behind the pipelined loop (one tile at a time from whichever counter still holds one).  Same bits
    as the static walk, so no tile is lost or done with another tile's rows."""
    E = _engine()
    C, D, sizes = 2, 512, [9000, 41, 7000, 3, 5000]
    W, We = synth.make_bank(123, D, C)
    bags, _ = synth.make_slide_set(321, sizes, D, We, C)
    X = torch.cat(bags).to(dev).to(dtype).contiguous()
    bank = E.Bank.get(W.to(dev), We.to(dev), dtype, dev)
    g = torch.Generator().manual_seed(3)
    mask = (torch.rand(sum(sizes), generator=g) > 0.5).to(torch.uint8)
    ref = E.SlideBatch(X, sizes, C, C + 4, 50, 10, (), mask=mask)
    ref.phase_a(bank)
    b = E.SlideBatch(X, sizes, C, C + 4, 50, 10, (), mask=mask)
    b.reserve_cus(0, ticket=True)
    everything = torch.full((128,), -1, dtype=torch.int32, device=dev)
    b.cu_reserved = everything
    b.c.cu_reserved = everything.data_ptr()
    b.stats.fill_(float("nan"))
    b.phase_a(bank)
    torch.cuda.synchronize()
    assert torch.equal(b.n_sel, ref.n_sel) and torch.equal(b.n_kept, ref.n_kept)
    off = 0
    for s_i, n in enumerate(sizes):
        nk = int(ref.n_kept[s_i])
        assert torch.equal(b.stats[:, off:off + nk], ref.stats[:, off:off + nk]), f"statistics of slide {s_i} differ"
        assert torch.equal(b.sel_flag[off:off + nk], ref.sel_flag[off:off + nk])
        off += n


@pytest.mark.parametrize("K,smallest,same", [(10, False, True), (16, False, False), (1, True, False), (10, True, True), (3, False, True)])
def test_wave_per_task_topk_gives_the_workgroup_kernels_bits(dev, K, smallest, same):
    """Launches of thousands of (segment, class) tasks take one WAVE per task (topk_mean_wave_kernel); fewer take the
    1,024-thread kernel.  Same pooled values, same chosen rows in the same order (key, then lower row first), same
    counts: ragged segments (1 ... 15,000 rows, shorter than K, exactly 64 / 256 / 257), ties everywhere (quantised keys),
    flat segments (every key equal: the candidate list overflows and the exact slow path runs), keys != values."""
    E = _engine()
    g = torch.Generator().manual_seed(100 + K)
    lens = [1, 2, 5, 15, 16, 17, 63, 64, 65, 255, 256, 257, 300, 1000, 4097, 15000, 7, 9000] + [int(v) for v in torch.randint(1, 3000, (110,), generator=g)]
    Cc = 20                                                        # 128 segments x 20 classes = 2,560 tasks: the wave kernel
    off = [0]
    for n in lens:
        off.append(off[-1] + n)
    N = off[-1]
    keys = torch.randn(Cc, N, generator=g)
    keys[3] = torch.round(keys[3] * 4) / 4                         # heavy ties
    keys[4] = 0.25                                                 # flat: every key equal
    keys[5, : off[16]] = -1.5                                      # flat prefix incl. the 15,000-row segment
    keys[6] = torch.round(keys[6])                                 # a few distinct values only
    keys[7, off[15] + 400: off[15] + 1800] += 20.0                 # the 15,000-row segment's top keys lie between the sampled runs
    keys[8, off[15]: off[16]] = torch.sort(keys[8, off[15]: off[16]])[0]      # ascending
    keys[9, off[15]: off[16]] = torch.sort(keys[9, off[15]: off[16]], descending=True)[0]
    vals = keys if same else torch.randn(Cc, N, generator=g)
    keys_d, vals_d = keys.to(dev), (keys.to(dev) if same else vals.to(dev))
    if same:
        vals_d = keys_d
    seg = torch.tensor(off, dtype=torch.int64, device=dev)
    p_w, i_w, c_w = E.topk_mean(keys_d, vals_d, K, smallest=smallest, want_idx=True, seg_off=seg)
    torch.cuda.synchronize()
    # the same segments in launches of 64 (1,280 tasks: the workgroup kernel)
    p_r, i_r, c_r = [], [], []
    for s0 in range(0, len(lens), 64):
        sub = seg[s0:s0 + 65]
        p, i, c = E.topk_mean(keys_d, vals_d, K, smallest=smallest, want_idx=True, seg_off=sub)
        p_r.append(p); i_r.append(i); c_r.append(c)
    p_r, i_r, c_r = torch.cat(p_r), torch.cat(i_r), torch.cat(c_r)
    assert torch.equal(c_w, c_r)
    assert torch.equal(i_w, i_r), "chosen rows / their order differ"
    assert torch.equal(p_w.view(torch.int32), p_r.view(torch.int32)), "pooled values differ in bits"
    # and against plain torch on one ragged segment
    s_i, c_i = 15, 0
    col = keys[c_i, off[s_i]:off[s_i + 1]]
    order = torch.argsort(-col if not smallest else col, stable=True)[:K]
    assert i_w[s_i, c_i].cpu().tolist() == order.tolist()



# ------------------------------------------------------------------ round 4: the step over the forward's tile records
def _train_epochs(dev, tile_records, C, D, dtype, sizes, j, K, epochs, seed, discard=(), plant=None, cache_scores=False):
    """`epochs` passes of main_moc.train over a resident split with the tile-record step on or off; -> everything a
    step leaves behind, as host arrays."""
    M, E = _mm(), _engine()
    keep = E.TILE_RECORDS
    E.TILE_RECORDS = tile_records
    try:
        W, We = synth.make_bank(seed, D, C)
        bags, labels = synth.make_slide_set(seed + 100, sizes, D, We, C)
        if plant is not None:
            bags = [plant(b_, We) for b_ in bags]
        M.set_classifier_bank(W.to(dev), We.to(dev))
        torch.manual_seed(seed)
        model = M.senet(D, 4).to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
        res = M.ResidentBags([b_.to(dtype) for b_ in bags], labels, dev, cache_scores=cache_scores)
        args = H.make_args(C, j, K, discard)
        torch.manual_seed(seed + 7)
        out = {"loss": [], "pooled": [], "topk": [], "path": []}
        for _ in range(epochs):
            M.train(model, res, opt, dev, args)
            torch.cuda.synchronize()
            t = M.train.last[0].meta_ws()[0]
            out["loss"].append(t["loss"].cpu().numpy().copy())
            out["pooled"].append(t["pooled"].cpu().numpy().copy())
            out["topk"].append(t["topk_idx"].cpu().numpy().copy())
            out["path"].append(int(t["n_pair"].cpu()[0]))
        out["params"] = H.flat_params(model)
        out["m"], out["v"] = H.flat_state(opt, "exp_avg"), H.flat_state(opt, "exp_avg_sq")
        return out
    finally:
        E.TILE_RECORDS = keep


@pytest.mark.parametrize("C,D,dtype,sizes,j,K", [
    (2, 512, torch.float32, [3000, 2500, 4100, 2800, 3333], 400, 10),
    (3, 512, torch.bfloat16, [2000, 2600, 1500, 2200], 300, 10),
    (2, 512, torch.float32, [40, 18, 9, 300], 400, 10),              # slides of a tile or two, S < K, S = 0 possible
    (30, 512, torch.bfloat16, [1800, 2200], 100, 10),                 # the wide step behind the sixteen-row forward
])
def test_host_known_row_counts_change_no_bit(dev, C, D, dtype, sizes, j, K, monkeypatch):
    """moc_batch_t.n_sel_host (round 4): from its second pass on, train() hands the steps a pinned copy of n_sel whose copy
    event has completed -- the forward takes S as an argument (one dependent load less), launches exactly the workgroups
    that have rows, the step reads fewer record keys per lane.  Four passes with it and four with the field left NULL:
    the same losses, pooled rows, parameters and moments, bit for bit; and the copy really was handed over."""
    E = _engine()
    used = []
    orig = E.SlideBatch.publish_n_sel
    monkeypatch.setattr(E.SlideBatch, "publish_n_sel", lambda self, allow=True: used.append(orig(self, allow)) or used[-1])
    a = _train_epochs(dev, True, C, D, dtype, sizes, j, K, 4, 6161 + C)
    assert any(used), "the host copy of n_sel was never handed to the train steps"
    monkeypatch.setattr(E.SlideBatch, "publish_n_sel", lambda self, allow=True: orig(self, False))
    b = _train_epochs(dev, True, C, D, dtype, sizes, j, K, 4, 6161 + C)
    for e in range(4):
        np.testing.assert_array_equal(a["loss"][e], b["loss"][e])
        np.testing.assert_array_equal(a["pooled"][e], b["pooled"][e])
        np.testing.assert_array_equal(a["topk"][e], b["topk"][e])
    for k in ("params", "m", "v"):
        np.testing.assert_array_equal(a[k], b[k])


@pytest.mark.parametrize("C,D,dtype,sizes,j,K,discard", [
    (2, 512, torch.float32, [3000, 2500, 4100, 2800, 3333], 400, 10, ()),
    (2, 512, torch.bfloat16, [3000, 2500, 4100], 400, 10, ("delta_diff",)),
    (3, 512, torch.float32, [2000, 2600, 1500, 2200], 300, 10, ()),
    (2, 256, torch.float16, [900, 700, 1100], 100, 1, ()),
    (4, 768, torch.float32, [1500, 1200, 1700], 150, 5, ("topk",)),
    (2, 512, torch.float32, [40, 18, 9, 300], 400, 10, ()),          # slides smaller than a few tiles, S < K
    (2, 1024, torch.float32, [2500, 1800], 400, 10, ()),
])
def test_tile_record_step_gives_the_full_score_steps_bits(dev, C, D, dtype, sizes, j, K, discard):
    """the step kernel pools among the records the forward leaves per 16-row tile instead of
    re-reading and ranking every mixed score.  Same pooled rows in the same (value desc, row asc) order, the same sums:
    losses, pooled logits, pooled rows, parameters and both Adam moments are BIT-identical to the round-3 step after
    three passes, and the records path (not its fall-back) is what ran."""
    a = _train_epochs(dev, False, C, D, dtype, sizes, j, K, 3, 4242 + C, discard)
    b = _train_epochs(dev, True, C, D, dtype, sizes, j, K, 3, 4242 + C, discard)
    for e in range(3):
        np.testing.assert_array_equal(a["loss"][e], b["loss"][e])
        np.testing.assert_array_equal(a["pooled"][e], b["pooled"][e])
        np.testing.assert_array_equal(a["topk"][e], b["topk"][e])
    for k in ("params", "m", "v"):
        np.testing.assert_array_equal(a[k], b[k])
    assert all(p not in (1000001, 1000002) for p in a["path"])      # round-3 kernel: no marker
    assert b["path"][-1] in (1000001, 1000002)                      # the tile-record kernel ran the last step ...
    if min(sizes) > 1000:
        assert b["path"][-1] == 1000001                             # ... on its records


def test_tile_record_step_falls_back_when_a_tile_holds_the_candidates(dev):
    """Sixteen consecutive rows that all carry the class signal: one tile then holds more than four of the pooled rows,
    the records cannot prove exactness (rho >= T0) and the kernel ranks the full scores instead -- same bits."""
    def plant(bag, We):
        bag = bag.clone()
        bag[32:48] = torch.nn.functional.normalize(bag[32:48] * 0.2 + 3.0 * We[:, 0], dim=1)
        return bag
    sizes = [3000, 2500, 2800]
    a = _train_epochs(dev, False, 2, 512, torch.float32, sizes, 400, 10, 2, 777, plant=plant)
    b = _train_epochs(dev, True, 2, 512, torch.float32, sizes, 400, 10, 2, 777, plant=plant)
    for e in range(2):
        np.testing.assert_array_equal(a["loss"][e], b["loss"][e])
        np.testing.assert_array_equal(a["topk"][e], b["topk"][e])
    for k in ("params", "m", "v"):
        np.testing.assert_array_equal(a[k], b[k])
    assert b["path"][-1] == 1000002


@pytest.mark.parametrize("C,D,dtype,sizes,j,K", [
    (2, 512, torch.float32, [3000, 2500, 4100, 2800], 400, 10),
    (3, 512, torch.bfloat16, [2000, 2600, 1500], 300, 10),
    (30, 512, torch.bfloat16, [1800, 2200], 100, 10),                 # compact statistics, the wide step
    (2, 256, torch.float16, [900, 700, 1100], 100, 5),
])
def test_cached_statistics_give_the_score_pass_bits(dev, C, D, dtype, sizes, j, K):
    """Opt-in `cache_scores` (moc_scores_from_cache): the statistics of every row from ONE unmasked score pass, every train
    pass copying its kept rows' statistics instead of reading the bags (main_moc.py:336-337 recomputes them per visit; the
    bank is frozen).  A row's statistics do not depend on the mask or on the rows sharing its MFMA tile: three passes end in
    the same bits -- losses, pooled rows, parameters, moments."""
    a = _train_epochs(dev, True, C, D, dtype, sizes, j, K, 3, 5151 + C)
    b = _train_epochs(dev, True, C, D, dtype, sizes, j, K, 3, 5151 + C, cache_scores=True)
    for e in range(3):
        np.testing.assert_array_equal(a["loss"][e], b["loss"][e])
        np.testing.assert_array_equal(a["topk"][e], b["topk"][e])
    for k in ("params", "m", "v"):
        np.testing.assert_array_equal(a[k], b[k])
