"""CPU oracle for the MOC hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a torch-CPU (fp32) restatement of the per-slide pipeline of the
reference (xmed-lab/MOC):  classifier-bank scoring -> four patch selectors ->
union -> candidate scores -> meta-learner gating -> top-K mean pooling -> CE ->
Adam.  It exists only so that tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py have something to check / time the HIP path
against.  Nothing under moc_amd/ may import it.

Pinning: every function here is checked against the reference's own Python
(imported from /root/reference, AST-extracted for main_moc.py) by
tests/golden/make_golden.py, whose outputs are committed as tests/golden/*.npz
and re-checked by tests/test_oracle_golden.py.  The reference holds no tests or
golden vectors of its own, so those generated fixtures
are the pin.

Reference lines followed (paths relative to the reference repo root):
  row scores ............ main_moc.py:336-337
  psi_p   (top-j) ....... utils/patch_selection_classifier_index.py:17-26
  psi_sig (softmax) ..... utils/patch_selection_classifier_index.py:28-36
  psi_dlt (|t1-t2|) ..... utils/patch_selection_classifier_index.py:38-51
  psi_bet (bottom bg) ... utils/patch_selection_classifier_index.py:53-87
  union / gather ........ main_moc.py:335-357
  candidate scores ...... main_moc.py:359-366
  meta-learner .......... main_moc.py:299-312
  gated mix ............. main_moc.py:391-403 (train), :482-492 (eval)
  top-K mean pooling .... utils/patch_selection_classifier.py:18-32
  zs pooling variants ... utils/patch_selection_classifier.py:35-80, :127-171
  train step ............ main_moc.py:378-410
  evaluation tails ...... main_moc.py:412-460, :462-520, :523-582
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

CONCH_TEMPERATURE = 56.3477  # main_moc.py:505
SELECTOR_NAMES = ("topk", "delta_softmax", "delta_diff", "bottomk")  # main_moc.py:341-350


# --------------------------------------------------------------------------
# per-row keys
# --------------------------------------------------------------------------
def _row_gap(logits: torch.Tensor) -> torch.Tensor:
    """|largest - second largest| over classes, per row (index.py:46-48)."""
    two = torch.topk(logits, 2, dim=1)[0]
    return torch.abs(two[:, 0] - two[:, 1])


def _cap(topj, n_rows: int) -> int:
    return min(max(topj), n_rows)


# --------------------------------------------------------------------------
# the four selectors: each returns int64 indices [maxj, C], value-ordered
# --------------------------------------------------------------------------
def sel_top(logits, topj):
    return logits.topk(_cap(topj, logits.size(0)), 0, True, True)[1]


def sel_softmax(logits, topj):
    return F.softmax(logits, dim=1).topk(_cap(topj, logits.size(0)), 0, True, True)[1]


def sel_gap(logits, topj):
    gap = _row_gap(logits)
    rep = torch.stack([gap] * logits.size(1), dim=1)
    return rep.topk(_cap(topj, logits.size(0)), 0, True, True)[1]


def _low_background_parts(logits_ext, n_fg, detection):
    """(foreground columns, background mass, extra key column) -- index.py:65-71 / classifier.py:146-152.
    detection=True: ONE foreground column (column 0), every other column is background, and the rows are also ranked by
    their largest background logit (appended as a second key column, index.py:83-84)."""
    if detection:
        bg = logits_ext[:, 1:]
        return logits_ext[:, 0].unsqueeze(1), bg.sum(dim=1), torch.topk(bg, 1, dim=1)[0][:, 0]
    return logits_ext[:, :n_fg], logits_ext[:, n_fg:].sum(dim=1), None


def sel_low_background(logits_ext, topj, n_classes, bottomk=None, detection=False):
    assert n_classes is not None, "coords_list should be provided"
    assert logits_ext.size(1) > n_classes, "logits should have more bg classes"
    maxj = _cap(topj, logits_ext.size(0))
    bottomk = maxj if bottomk is None else min(bottomk, logits_ext.size(0))
    fg, bg_sum, extra = _low_background_parts(logits_ext, n_classes, detection)
    low = bg_sum.topk(bottomk, 0, False, True)[1]       # smallest background mass
    keys = fg[low]
    if extra is not None:
        keys = torch.cat([keys, extra[low].unsqueeze(1)], dim=1)
    order = keys.topk(maxj, 0, True, True)[1]
    return low[order]


# --------------------------------------------------------------------------
# pooling functions (zs_evaluation's pooling_func choices)
# --------------------------------------------------------------------------
def _finish_pool(values, topj, maxj):
    pooled = {j: values[: min(j, maxj)].mean(dim=0, keepdim=True) for j in topj}
    preds = {j: v.argmax(dim=1) for j, v in pooled.items()}
    return preds, pooled


def pool_top(logits, topj, return_indices=False):
    maxj = _cap(topj, logits.size(0))
    values, idx = logits.topk(maxj, 0, True, True)
    out = _finish_pool(values, topj, maxj)
    return (*out, idx) if return_indices else out


def pool_softmax(logits, topj, return_indices=False):
    maxj = _cap(topj, logits.size(0))
    idx = F.softmax(logits, dim=1).topk(maxj, 0, True, True)[1]
    values = torch.stack([logits[idx[:, c], c] for c in range(logits.size(1))], dim=1)
    out = _finish_pool(values, topj, maxj)
    return (*out, idx) if return_indices else out


def pool_gap(logits, topj, return_indices=False):
    maxj = _cap(topj, logits.size(0))
    idx = sel_gap(logits, topj)
    values = logits[idx[:, 0]]
    out = _finish_pool(values, topj, maxj)
    return (*out, idx) if return_indices else out


def pool_low_background(logits_ext, topj, n_classes, return_indices=False, bottomk=None, detection=False):
    assert logits_ext.size(1) > n_classes, "logits should have more bg classes"
    maxj = _cap(topj, logits_ext.size(0))
    bottomk = maxj if bottomk is None else bottomk
    fg, bg_sum, extra = _low_background_parts(logits_ext, n_classes, detection)
    low = bg_sum.topk(bottomk, 0, False, True)[1]
    keys = fg[low]
    if extra is not None:
        keys = torch.cat([keys, extra[low].unsqueeze(1)], dim=1)
    fg_values, order = keys.topk(maxj, 0, True, True)
    out = _finish_pool(fg_values, topj, maxj)
    return (*out, low[order]) if return_indices else out


# --------------------------------------------------------------------------
# slide_process
# --------------------------------------------------------------------------
def draw_mask(n_rows: int) -> torch.Tensor:
    """Row mask exactly as main_moc.py:330 draws it (CPU default generator)."""
    return torch.rand(n_rows) > 0.5


def slide_process(feat, W, W_ext, n_classes, topj=10, mask=None, discard=()):
    """Returns the reference's 6-key dict.  `mask` (bool [N]) replaces the
    reference's internal torch.rand draw so both sides can be fed the same one;
    pass draw_mask(N) to reproduce `random_mask=True`."""
    if mask is not None:
        feat = feat[mask]
    logits = feat @ W
    logits_ext = feat @ W_ext
    tj = [topj]
    chosen = set()
    if "topk" not in discard:
        chosen.update(sel_top(logits, tj).flatten().tolist())
    if "delta_softmax" not in discard:
        chosen.update(sel_softmax(logits, tj).flatten().tolist())
    if "delta_diff" not in discard:
        chosen.update(sel_gap(logits, tj).flatten().tolist())
    if "bottomk" not in discard:
        chosen.update(sel_low_background(logits_ext, tj, n_classes).flatten().tolist())
    chosen = sorted(chosen)
    sub = feat[chosen]
    sub_logits = sub @ W
    sub_ext = sub @ W_ext
    C = sub_logits.size(1)
    gap = _row_gap(sub_logits)
    bg_max = sub_ext[:, n_classes:].max(dim=1)[0]
    return {
        "selected_index": chosen,
        "selected_feat": sub,
        "logits_top_classifier": sub_logits,
        "logits_delta_softmax_classifier": sub_logits.softmax(dim=1),
        "logits_delta_diff_classifier": torch.stack([gap] * C, dim=1),
        "logits_bottomk_irrel_classifier": torch.stack([bg_max] * C, dim=1),
    }


# --------------------------------------------------------------------------
# meta-learner and the per-slide step
# --------------------------------------------------------------------------
class Senet(nn.Module):
    """Same parameter names/shapes as the reference `senet` (state_dict keys
    model.0.weight, model.0.bias, model.2.weight, model.2.bias)."""

    def __init__(self, in_dim=512, out_dim=4):
        super().__init__()
        self.hidden_dim = 64
        self.model = nn.Sequential(
            nn.Linear(in_dim, self.hidden_dim), nn.ReLU(),
            nn.Linear(self.hidden_dim, out_dim), nn.Sigmoid())

    def forward(self, x):
        return self.model(x)


def make_optimizer(model):
    return torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)  # main_moc.py:316


_CAND_KEYS = ("logits_top_classifier", "logits_delta_softmax_classifier",
              "logits_delta_diff_classifier", "logits_bottomk_irrel_classifier")


def mix_train(gates, sr, discard=()):
    parts = [gates[:, i].unsqueeze(1) * sr[k] for i, k in enumerate(_CAND_KEYS)]
    out = torch.zeros_like(parts[0])
    for name, p in zip(SELECTOR_NAMES, parts):
        if name not in discard:
            out = out + p
    return out


def mix_eval(gates, sr, discard=()):
    """Eval-side mix incl. the reference's quirk: psi_p always added, and the
    last test is for the string "delta_bottomk" (main_moc.py:486-492), so the
    background term is added unless that exact (never used) string is given."""
    parts = [gates[:, i].unsqueeze(1) * sr[k] for i, k in enumerate(_CAND_KEYS)]
    out = parts[0]
    if "delta_softmax" not in discard:
        out = out + parts[1]
    if "delta_diff" not in discard:
        out = out + parts[2]
    if "delta_bottomk" not in discard:
        out = out + parts[3]
    return out


def train_step(model, optimizer, feat, label, W, W_ext, n_classes, topj, topk,
               mask, discard=()):
    """One meta-step (main_moc.py:381-410).  Returns (loss, pooled logits)."""
    sr = slide_process(feat, W, W_ext, n_classes, topj, mask=mask, discard=discard)
    gates = model(sr["selected_feat"])
    mixed = mix_train(gates, sr, discard)
    pooled = pool_top(mixed, [topk])[1][topk]
    loss = F.cross_entropy(pooled, label.view(1))
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss.detach(), pooled.detach()


def train_epoch(model, optimizer, bags, labels, W, W_ext, n_classes, topj, topk,
                masks=None, discard=()):
    """`bags`: list of [N_i, D] tensors in loader order.  masks: list of bool
    tensors or None (= draw from the CPU generator like the reference)."""
    model.train()
    losses = []
    for i, (x, y) in enumerate(zip(bags, labels)):
        m = masks[i] if masks is not None else draw_mask(x.size(0))
        loss, _ = train_step(model, optimizer, x, torch.as_tensor(y), W, W_ext,
                             n_classes, topj, topk, m, discard)
        losses.append(float(loss))
    return losses


def _metrics(pooled_all, labels_all, loss_sum, n_div, n_real):
    from sklearn.metrics import roc_auc_score
    probs = F.softmax(pooled_all * CONCH_TEMPERATURE, dim=1)
    correct = int((pooled_all.argmax(dim=1) == labels_all).sum())
    if probs.shape[1] == 2:
        auc = roc_auc_score(labels_all.numpy(), probs[:, 1].numpy())
    else:
        auc = roc_auc_score(labels_all.numpy(), probs.numpy(), multi_class="ovo", average="macro")
    return {"loss": loss_sum / n_div, "acc": correct / n_real, "auc": auc}


def evaluation(model, bags, labels, W, W_ext, n_classes, topj, topk, discard=(),
               len_dataset=None, return_logits=False):
    """main_moc.py:462-520.  `len_dataset` = len(loader.dataset) after repeat_num
    is restored (the loss divisor); defaults to the number of bags."""
    model.eval()
    pooled_all, loss_sum = [], 0.0
    with torch.no_grad():
        for x, y in zip(bags, labels):
            sr = slide_process(x, W, W_ext, n_classes, topj, discard=discard)
            mixed = mix_eval(model(sr["selected_feat"]), sr, discard)
            pooled = pool_top(mixed, [topk])[1][topk]
            loss_sum += F.cross_entropy(pooled, torch.as_tensor(y).view(1)).item()
            pooled_all.append(pooled)
    pooled_all = torch.cat(pooled_all, 0)
    lab = torch.as_tensor(labels).long()
    out = _metrics(pooled_all, lab, loss_sum, len_dataset or len(bags), len(bags))
    return (out, pooled_all) if return_logits else out


def zs_evaluation(bags, labels, W, W_ext, n_classes, topk, pooling="topj",
                  len_dataset=None, return_logits=False):
    """main_moc.py:412-460 with pooling in {topj, delta_softmax, delta_diff, bottomk}."""
    pooled_all, loss_sum = [], 0.0
    with torch.no_grad():
        for x, y in zip(bags, labels):
            if pooling == "bottomk":
                pooled = pool_low_background(x @ W_ext, [topk], n_classes)[1][topk]
            else:
                fn = {"topj": pool_top, "delta_softmax": pool_softmax, "delta_diff": pool_gap}[pooling]
                pooled = fn(x @ W, [topk])[1][topk]
            loss_sum += F.cross_entropy(pooled, torch.as_tensor(y).view(1)).item()
            pooled_all.append(pooled)
    pooled_all = torch.cat(pooled_all, 0)
    lab = torch.as_tensor(labels).long()
    out = _metrics(pooled_all, lab, loss_sum, len_dataset or len(bags), len(bags))
    return (out, pooled_all) if return_logits else out


def ablation_evaluation(bags, labels, W, W_ext, n_classes, topj, topk, mode,
                        len_dataset=None, return_logits=False):
    """main_moc.py:523-582, mode in {avg, sum, max}."""
    pooled_all, loss_sum = [], 0.0
    with torch.no_grad():
        for x, y in zip(bags, labels):
            sr = slide_process(x, W, W_ext, n_classes, topj)
            cands = torch.stack([sr[k] for k in _CAND_KEYS], 0)
            if mode == "avg":
                mixed = 0.25 * cands[0] + 0.25 * cands[1] + 0.25 * cands[2] + 0.25 * cands[3]
            elif mode == "sum":
                mixed = cands[0] + cands[1] + cands[2] + cands[3]
            elif mode == "max":
                mixed = cands.max(dim=0)[0]
            else:
                raise ValueError(mode)
            pooled = pool_top(mixed, [topk])[1][topk]
            loss_sum += F.cross_entropy(pooled, torch.as_tensor(y).view(1)).item()
            pooled_all.append(pooled)
    pooled_all = torch.cat(pooled_all, 0)
    lab = torch.as_tensor(labels).long()
    out = _metrics(pooled_all, lab, loss_sum, len_dataset or len(bags), len(bags))
    return (out, pooled_all) if return_logits else out
