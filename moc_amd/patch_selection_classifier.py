"""Drop-in for the pooling functions of the reference's
utils/patch_selection_classifier.py that the MOC path can reach (:18-80, :127-171):
each ranks rows by some key per class and averages the ORIGINAL logits of the best
`j` rows.  All four are one kernel here (moc_topk_mean: key column, value column).

Return structure as in the reference: (preds {j: [1] int64}, pooled {j: [1, C]})
and, with return_indices, the value-ordered indices [maxj, C].
"""
from __future__ import annotations

import torch

from . import engine
from .patch_selection_classifier_index import row_stats, _require_gpu


def _pool(keys, vals, topj, maxj, smallest=False, shared=False, return_indices=False):
    pooled = {}
    for j in topj:
        pooled[j] = engine.topk_mean(keys, vals, min(j, maxj), smallest=smallest, key_shared=shared)
    preds = {j: v.argmax(dim=1) for j, v in pooled.items()}
    if not return_indices:
        return preds, pooled
    _, idx, _ = engine.topk_mean(keys, vals, maxj, smallest=smallest, key_shared=shared, want_idx=True)
    return preds, pooled, idx[0].t().to(torch.int64).contiguous()


def topj_pooling(logits, topj, return_indices=False, **kwargs):
    """utils/patch_selection_classifier.py:18-32 -- the MIL aggregator of train/eval."""
    _require_gpu(logits, "topj_pooling")
    maxj = min(max(topj), logits.size(0))
    if torch.is_grad_enabled() and logits.requires_grad:
        # a caller-written training loop (main_moc.py:405-409): the pooled rows carry the gradient, 1/k each
        from .pool_autograd import topk_mean_pool
        pooled = {j: topk_mean_pool(logits, min(j, maxj)) for j in topj}
        preds = {j: v.argmax(dim=1) for j, v in pooled.items()}
        if not return_indices:
            return preds, pooled
        lt = logits.detach().to(torch.float32).t().contiguous()
        _, idx, _ = engine.topk_mean(lt, lt, maxj, want_idx=True)
        return preds, pooled, idx[0].t().to(torch.int64).contiguous()
    lt = logits.detach().to(torch.float32).t().contiguous()
    return _pool(lt, lt, topj, maxj, return_indices=return_indices)


def delta_softmax_classifier_pooling(logits, topj, return_indices=False, **kwargs):
    """utils/patch_selection_classifier.py:35-53"""
    maxj = min(max(topj), logits.size(0))
    C_ = logits.size(1)
    st = row_stats(logits, C_)
    return _pool(st[C_:2 * C_], st[:C_], topj, maxj, return_indices=return_indices)


def delta_diff_classifier_pooling(logits, topj, return_indices=False, **kwargs):
    """utils/patch_selection_classifier.py:56-80"""
    maxj = min(max(topj), logits.size(0))
    C_ = logits.size(1)
    if C_ < 2:
        raise RuntimeError("selected index k out of range")
    st = row_stats(logits, C_)
    return _pool(st[2 * C_:2 * C_ + 1], st[:C_], topj, maxj, shared=True, return_indices=return_indices)


def bottomk_irrel_classifier_pooling(logits, topj, return_indices=False, coords_list=None, bottomk=None,
                                     detection=False, **kwargs):
    """utils/patch_selection_classifier.py:127-171"""
    assert coords_list is not None, "coords_list should be provided"
    if type(coords_list) == int:
        assert logits.size(1) > coords_list, "logits should have more bg classes"
        n_fg = coords_list
    elif type(coords_list) == list:
        assert logits.size(1) > len(coords_list), "logits should have more bg classes"
        n_fg = len(coords_list)
    else:
        raise ValueError("coords_list should be int or list")
    maxj = min(max(topj), logits.size(0))
    if bottomk is None:
        bottomk = maxj
    if detection:
        # :146-149, :161-162 -- ONE foreground column; pooled over two key columns: the foreground logit and the largest
        # background logit of the rows with the least background mass
        st = row_stats(logits, 1)
        bg_key = st[3:4]
    else:
        st = row_stats(logits, n_fg)
        bg_key = st[2 * n_fg + 1:2 * n_fg + 2]
    # rows with the smallest background mass, then their foreground logits ranked per class
    _, low, _ = engine.topk_mean(bg_key, bg_key, bottomk, smallest=True, want_idx=True)
    bg_rows = low[0, 0].to(torch.int64)
    fg_src = torch.stack([st[0], st[4]]) if detection else st[:n_fg]
    fg = fg_src.index_select(1, bg_rows).contiguous()
    out = _pool(fg, fg, topj, maxj, return_indices=return_indices)
    if return_indices:
        return out[0], out[1], bg_rows[out[2]]
    return out
