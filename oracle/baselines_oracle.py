"""CPU restatement of the pooling / instance-selection steps the baseline models end in (SURVEY.md
section 8, row f3) -- TEST INFRASTRUCTURE ONLY, like oracle/moc_oracle.py: nothing under moc_amd/
imports this file.

  topk_mean_pool  models/model_adapters.py:173-183 (`topj_pooling`: per-class mean of the top-j logits)
  top_rows        models/model_mil.py:40 (`torch.topk(y_probs[:, 1], top_k)[1]`)
  top_entry       models/model_mil.py:88-89 (`y_probs.view(1, -1).argmax(1)` -> (row, class))
  gated_attention_pool   models/model_clam.py:58-63, :178-183, :206 (row f4)

`patched()` swaps them into moc_amd.model_mil / moc_amd.model_adapters so that the CPU tests can pin the
rest of those modules (layers, mixing formulas, constructor RNG order, trainer hooks) to the fixtures
tests/golden/baselines.npz, which tests/golden/make_golden.py produced from the reference's own
classes.  The -m gpu tests run the same cases through the real HIP pooling.
"""
import contextlib

import torch


def topk_mean_pool(logits: torch.Tensor, k: int) -> torch.Tensor:
    j = min(int(k), logits.size(0))
    return logits.topk(j, 0, True, True)[0].mean(dim=0, keepdim=True)


def top_rows(scores: torch.Tensor, k: int = 1) -> torch.Tensor:
    return torch.topk(scores, int(k), dim=0)[1]


def top_entry(probs: torch.Tensor):
    m = int(probs.reshape(1, -1).argmax(1))
    return m // probs.size(1), m % probs.size(1)


def gated_attention_pool(h, Wa, ba, Wb, bb, Wc, bc):
    """models/model_clam.py:58-63 (Attn_Net_Gated.forward), :178-183 and :206 (softmax over N, M = A h):
    -> (A_raw [K, N], M [K, L])."""
    a = torch.tanh(torch.nn.functional.linear(h, Wa, ba))
    b = torch.sigmoid(torch.nn.functional.linear(h, Wb, bb))
    A_raw = torch.nn.functional.linear(a.mul(b), Wc, bc).transpose(1, 0)
    return A_raw, torch.mm(torch.softmax(A_raw, dim=1), h)


@contextlib.contextmanager
def patched():
    import moc_amd.model_adapters as A
    import moc_amd.model_clam as Mc
    import moc_amd.model_mil as Mm
    saved = (A.topk_mean_pool, Mm.top_rows, Mm.top_entry, Mc.gated_attention_pool)
    A.topk_mean_pool, Mm.top_rows, Mm.top_entry, Mc.gated_attention_pool = topk_mean_pool, top_rows, top_entry, gated_attention_pool
    try:
        yield
    finally:
        A.topk_mean_pool, Mm.top_rows, Mm.top_entry, Mc.gated_attention_pool = saved
