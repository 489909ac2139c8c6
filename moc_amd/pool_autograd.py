"""Differentiable top-j mean pooling and top-instance selection on the HIP path (SURVEY.md section 8,
row f3): what the baseline models' forwards end in (models/model_adapters.py:173-183 `topj_pooling`,
models/model_mil.py:41-44 / :89-92 top-instance pick).  The selection runs in moc_topk_mean; autograd
sees a gather (only the pooled rows carry gradient, 1/k each)."""
from __future__ import annotations

import torch

from . import engine


class _TopkMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits: torch.Tensor, k: int):
        assert logits.dim() == 2 and logits.is_cuda, "topk_mean_pool: [N, C] logits on the GPU (no CPU fallback)"
        cols = logits.detach().to(torch.float32).t().contiguous()          # [C, N]: the kernel is column-wise
        pooled, idx, cnt = engine.topk_mean(cols, cols, int(k), want_idx=True)
        ctx.save_for_backward(idx[0], cnt[0])
        ctx.shape, ctx.dtype = tuple(logits.shape), logits.dtype
        return pooled.to(logits.dtype)                                      # [1, C]

    @staticmethod
    def backward(ctx, g):
        idx, cnt = ctx.saved_tensors
        n, c = ctx.shape
        grad = torch.zeros((n, c), dtype=torch.float32, device=g.device)
        kk = int(cnt[0].item())                                             # min(k, N), the same for every class
        if kk > 0:
            rows = idx[:, :kk].reshape(-1).long()
            cls = torch.arange(c, device=g.device).repeat_interleave(kk)
            grad[rows, cls] = (g[0].to(torch.float32) / kk).repeat_interleave(kk)
        return grad.to(ctx.dtype), None


def topk_mean_pool(logits: torch.Tensor, k: int) -> torch.Tensor:
    """[N, C] -> [1, C]: per class, the mean of its min(k, N) largest logits (ties: lowest row first)."""
    return _TopkMean.apply(logits, int(k))


def top_rows(scores: torch.Tensor, k: int = 1) -> torch.Tensor:
    """Rows of the k largest entries of a [N] score vector, in descending order (int64 [k])."""
    assert scores.dim() == 1 and scores.is_cuda
    col = scores.detach().to(torch.float32).view(1, -1).contiguous()
    _, idx, cnt = engine.topk_mean(col, col, int(k), want_idx=True)
    return idx[0, 0, : int(cnt[0, 0].item())].long()


def top_entry(probs: torch.Tensor):
    """(row, class) of the largest entry of a [N, C] matrix -- the first one in row-major order among
    equals, as `probs.view(1, -1).argmax(1)` picks (models/model_mil.py:88-89)."""
    assert probs.dim() == 2 and probs.is_cuda
    cols = probs.detach().to(torch.float32).t().contiguous()
    best, idx, _ = engine.topk_mean(cols, cols, 1, want_idx=True)           # per class: top value and its lowest row
    best, rows = best[0], idx[0, :, 0].long()
    c = probs.size(1)
    flat = torch.where(best == best.max(), rows * c + torch.arange(c, device=probs.device), torch.iinfo(torch.int64).max)
    f = int(flat.min().item())
    return f // c, f % c
