"""Nystrom attention, restated: the one third-party layer the reference's TransMIL needs (models/model_mil.py:6,
:108-116: `from nystrom_attention import NystromAttention`).

The package (lucidrains/nystrom-attention; the TransMIL authors pin 0.0.9, the reference pins nothing -- it has no
requirements file) is absent from this image and from the reference tree, so the reference's own TransMIL cannot be
constructed here and NO fixture can be generated for it: **parity of this module is unpinned**.  What follows is the
published algorithm (Xiong et al., "Nystromformer", AAAI 2021, as that package implements it): segment-mean landmarks,
three softmax kernels, the Moore-Penrose pseudo-inverse by the cubic iteration of the paper, the depth-wise
convolution residual on the values.  Module / parameter names follow the package (`to_qkv`, `to_out.0`, `res_conv`)
so that a TransMIL checkpoint trained with it loads.  Checked here against what CAN be checked: exact softmax
attention in the limit where every token is its own landmark, the pseudo-inverse against `torch.linalg.pinv`, shapes
and padding (tests/test_baselines_cpu.py).  Plain torch operations: this is a signature shim (SURVEY.md section 8,
f3), not part of the HIP hot path."""
from __future__ import annotations

from math import ceil

import torch
import torch.nn as nn
import torch.nn.functional as F


def moore_penrose_iter_pinv(x: torch.Tensor, iters: int = 6) -> torch.Tensor:
    """Pseudo-inverse of the [..., m, m] landmark kernel by Z <- 1/4 Z (13 I - XZ (15 I - XZ (7 I - XZ)))."""
    abs_x = x.abs()
    col, row = abs_x.sum(dim=-1), abs_x.sum(dim=-2)
    z = x.transpose(-1, -2) / (col.max() * row.max())
    eye = torch.eye(x.shape[-1], device=x.device, dtype=x.dtype).unsqueeze(0)
    for _ in range(iters):
        xz = x @ z
        z = 0.25 * z @ (13 * eye - (xz @ (15 * eye - (xz @ (7 * eye - xz)))))
    return z


class NystromAttention(nn.Module):
    def __init__(self, dim, dim_head=64, heads=8, num_landmarks=256, pinv_iterations=6, residual=True,
                 residual_conv_kernel=33, eps=1e-8, dropout=0.0):
        super().__init__()
        self.eps, self.heads = eps, heads
        inner = heads * dim_head
        self.num_landmarks, self.pinv_iterations = num_landmarks, pinv_iterations
        self.scale = dim_head ** -0.5
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(dropout))
        self.residual = residual
        if residual:
            k = residual_conv_kernel
            self.res_conv = nn.Conv2d(heads, heads, (k, 1), padding=(k // 2, 0), groups=heads, bias=False)

    def forward(self, x, mask=None, return_attn=False):
        assert mask is None and not return_attn, "the reference calls attn(x) only (models/model_mil.py:119)"
        b, n, _ = x.shape
        h, m = self.heads, self.num_landmarks
        if n % m > 0:                                     # pad at the FRONT so that the length divides into m landmarks
            x = F.pad(x, (0, 0, m - (n % m), 0), value=0)
        q, k, v = self.to_qkv(x).chunk(3, dim=-1)
        q, k, v = (t.reshape(b, t.shape[1], h, -1).transpose(1, 2) for t in (q, k, v))      # b h n d
        q = q * self.scale
        l = ceil(n / m)                                   # tokens per landmark
        q_l = q.reshape(b, h, -1, l, q.shape[-1]).sum(dim=3) / l
        k_l = k.reshape(b, h, -1, l, k.shape[-1]).sum(dim=3) / l
        attn1 = (q @ k_l.transpose(-1, -2)).softmax(dim=-1)
        attn2 = (q_l @ k_l.transpose(-1, -2)).softmax(dim=-1)
        attn3 = (q_l @ k.transpose(-1, -2)).softmax(dim=-1)
        out = (attn1 @ moore_penrose_iter_pinv(attn2, self.pinv_iterations)) @ (attn3 @ v)
        if self.residual:
            out = out + self.res_conv(v)
        out = out.transpose(1, 2).reshape(b, out.shape[2], -1)
        return self.to_out(out)[:, -n:]
