"""Drop-in for the reference's utils/patch_selection_classifier_index.py: the four
patch selectors as value-ordered index tensors [maxj, C] (int64, on the input's
device).  Same names, arguments and assertion messages; the ranking runs in
libmoc_hip (moc_row_stats + moc_topk_mean), there is no CPU path.

Inside slide_process the selectors are not called one by one: the fused
moc_select kernel forms their union directly (moc_amd/main_moc.py).
"""
from __future__ import annotations


import torch

from . import engine
from ._lib import check, lib, ptr


def _require_gpu(t: torch.Tensor, who: str):
    if not t.is_cuda:
        raise RuntimeError(f"{who}: moc_amd runs on the GPU only (got a {t.device} tensor); "
                           "move it with .to('cuda') -- there is no CPU fallback")


def row_stats(logits: torch.Tensor, n_fg: int) -> torch.Tensor:
    """[2C+3, N] fp32: logits[C] | softmax[C] | |top1-top2| | sum(bg) | max(bg)."""
    _require_gpu(logits, "row_stats")
    lg = logits.detach().to(torch.float32).contiguous()
    N, Ct = lg.shape
    out = torch.empty((2 * n_fg + 3, N), dtype=torch.float32, device=lg.device)
    check(lib().moc_row_stats(ptr(lg), N, Ct, n_fg, ptr(out), engine._stream()), "moc_row_stats")
    return out


def _ranked(keys: torch.Tensor, maxj: int, smallest=False, shared=False) -> torch.Tensor:
    """keys [C, N] (or [1, N] if shared) -> int64 [maxj, C] rows ordered by key."""
    _, idx, _ = engine.topk_mean(keys, keys, maxj, smallest=smallest, want_idx=True)
    return idx[0].t().to(torch.int64).contiguous()


def index_topj_classifier(logits, topj, **kwargs):
    """utils/patch_selection_classifier_index.py:17-26"""
    maxj = min(max(topj), logits.size(0))
    C_ = logits.size(1)
    return _ranked(row_stats(logits, C_)[:C_], maxj)


def index_delta_softmax_classifier(logits, topj, **kwargs):
    """utils/patch_selection_classifier_index.py:28-36"""
    maxj = min(max(topj), logits.size(0))
    C_ = logits.size(1)
    return _ranked(row_stats(logits, C_)[C_:2 * C_], maxj)


def index_delta_diff_classifier(logits, topj, **kwargs):
    """utils/patch_selection_classifier_index.py:38-51 (C identical columns)"""
    maxj = min(max(topj), logits.size(0))
    C_ = logits.size(1)
    if C_ < 2:
        raise RuntimeError("selected index k out of range")   # torch.topk(logits, 2, dim=1) on one column
    one = _ranked(row_stats(logits, C_)[2 * C_:2 * C_ + 1], maxj)
    return one.expand(maxj, C_).contiguous()


def index_bottomk_irrel_classifier(logits, topj, n_classes, bottomk=None, detection=False, **kwargs):
    """utils/patch_selection_classifier_index.py:53-87"""
    assert n_classes is not None, "coords_list should be provided"
    assert logits.size(1) > n_classes, "logits should have more bg classes"
    maxj = min(max(topj), logits.size(0))
    if bottomk is None:
        bottomk = maxj
    if bottomk > logits.size(0):
        print("heyhey small", bottomk, logits.size(0))
        bottomk = logits.size(0)
    if detection:
        # :65-68, :83-84 -- ONE foreground column, every other column background; the candidate rows are ranked by the
        # foreground logit AND by their largest background logit: two key columns out of the same statistics
        st = row_stats(logits, 1)                              # logits[:, 0] | softmax | gap | sum(bg) | max(bg)
        bg_rows = _ranked(st[3:4], bottomk, smallest=True)[:, 0]
        fg = torch.stack([st[0], st[4]]).index_select(1, bg_rows).contiguous()
    else:
        st = row_stats(logits, n_classes)
        bg_rows = _ranked(st[2 * n_classes + 1:2 * n_classes + 2], bottomk, smallest=True)[:, 0]   # [bottomk]
        fg = st[:n_classes].index_select(1, bg_rows)                                             # [C, bottomk]
    order = _ranked(fg, min(maxj, bottomk))                                                       # [maxj, C]
    return bg_rows[order]
