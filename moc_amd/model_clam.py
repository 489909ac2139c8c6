"""Gated-attention MIL (ABMIL / CLAM) under the reference's class names, constructor arguments, parameter
names and forward contract (reference models/model_clam.py).

forward(h [N, size0]) -> (logits [1, C], Y_prob [1, C], Y_hat [1, 1], A_raw [K, N], results_dict).
What makes these models is the aggregation over the N patches -- attention scores from a tanh x sigmoid
gate, softmax over N, attention-weighted sum of the patch features.  That whole step is ONE pass over
the bag on the HIP path (moc_gated_attention_pool: fp32 MFMA projections, gate and scores in registers,
online softmax; engine.gated_attention_pool); the layers around it (the first fc, the bag and instance
classifiers) are plain torch GEMMs.  The backward pass is HIP as well (moc_gated_attention_backward): the forward keeps
neither the [N, D] activations nor the softmax, so one recompute pass (the forward's main loop) turns the arriving
gradients into those at the two pre-activations, with the gate's derivative taken in registers, and only two plain
GEMMs (dW = dab^T h, dh = dab [Wa; Wb; gM]) go to the library.

The un-gated network (`gate=False`, Attn_Net: A = Wc tanh(Wa h + ba) + bc) runs through the same kernel with a gate
that is exactly one: Wb = 0 and bb = 40 give sigmoid(40) = 1 - 4e-18, which IS 1.0f in fp32 (and 1 + e^-40 == 1.0f in
the kernel's own form of the gate).  Dropout inside the attention network (p = 0.25 on the tanh / sigmoid branches) is
active only in training mode; its masks are torch's, so in that one mode the scores are formed by torch operations
(evaluation, and training without dropout, stay on the kernel)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import engine


def initialize_weights(module):
    """utils/utils.py:399-407."""
    for m in module.modules():
        if isinstance(m, nn.Linear):
            nn.init.xavier_normal_(m.weight)
            m.bias.data.zero_()
        elif isinstance(m, nn.BatchNorm1d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


class _GatedAttentionPool(torch.autograd.Function):
    """Forward and backward both in HIP: the backward recomputes the gate in one pass over the bag (nothing but A_raw
    was kept) and leaves two plain GEMMs to the library (engine.gated_attention_backward)."""

    @staticmethod
    def forward(ctx, h, Wa, ba, Wb, bb, Wc, bc):
        A_raw, M = engine.gated_attention_pool(h, Wa, ba, Wb, bb, Wc, bc)
        ctx.save_for_backward(h, Wa, ba, Wb, bb, Wc, A_raw)
        return A_raw, M

    @staticmethod
    def backward(ctx, gA, gM):
        h, Wa, ba, Wb, bb, Wc, A_raw = ctx.saved_tensors
        if gA is None and gM is None:
            return (None,) * 7
        res = engine.gated_attention_backward(h, Wa, ba, Wb, bb, Wc, A_raw, gA, gM)
        return tuple(r if need else None for r, need in zip(res, ctx.needs_input_grad))


def gated_attention_pool(h, Wa, ba, Wb, bb, Wc, bc):
    """(A_raw [K, N], M [K, L]) with autograd."""
    return _GatedAttentionPool.apply(h, Wa, ba, Wb, bb, Wc, bc)


def _pool_with_dropout(h, attn):
    """Training mode with dropout inside the attention network (models/model_clam.py:24-25, :50-52): the masks are
    torch's, so this one mode forms the scores with torch operations -- `attn.scores_torch` applies the module's own
    Dropout layers exactly where the reference has them.  -> (A_raw [K, N], M [K, L])."""
    A_raw = attn.scores_torch(h).t()
    return A_raw, torch.softmax(A_raw, dim=1) @ h


ONE_GATE_BIAS = 40.0      # sigmoid(40) == 1.0f: an un-gated network as a gated one (module docstring)


class Attn_Net(nn.Module):
    """models/model_clam.py:15-33 (attention without gating): forward(x) -> (A [N, n_classes], x).  Same parameter
    names as the reference (`module.0`, `module.2` / `module.3` with dropout)."""

    def __init__(self, L=1024, D=256, dropout=False, n_classes=1):
        super().__init__()
        mod = [nn.Linear(L, D), nn.Tanh()]
        if dropout:
            mod.append(nn.Dropout(0.25))
        mod.append(nn.Linear(D, n_classes))
        self.module = nn.Sequential(*mod)
        self.dropout = dropout
        self._one_gate = None

    def operands(self):
        lin_a, lin_c = self.module[0], self.module[-1]
        g = self._one_gate
        if g is None or g[0].device != lin_a.weight.device or g[0].shape != lin_a.weight.shape:
            g = self._one_gate = (torch.zeros_like(lin_a.weight), torch.full_like(lin_a.bias, ONE_GATE_BIAS))
        return (lin_a.weight, lin_a.bias, g[0], g[1], lin_c.weight, lin_c.bias)

    def scores_torch(self, x):
        return self.module(x)

    def forward(self, x):
        if self.dropout and self.training:
            return self.scores_torch(x), x
        A_raw, _ = gated_attention_pool(x, *self.operands())
        return A_raw.t(), x


class Attn_Net_Gated(nn.Module):
    """models/model_clam.py:41-64: forward(x) -> (A [N, n_classes], x)."""

    def __init__(self, L=1024, D=256, dropout=False, n_classes=1):
        super().__init__()
        a, b = [nn.Linear(L, D), nn.Tanh()], [nn.Linear(L, D), nn.Sigmoid()]
        if dropout:
            a.append(nn.Dropout(0.25))
            b.append(nn.Dropout(0.25))
        self.attention_a, self.attention_b = nn.Sequential(*a), nn.Sequential(*b)
        self.attention_c = nn.Linear(D, n_classes)
        self.dropout = dropout

    def operands(self):
        return (self.attention_a[0].weight, self.attention_a[0].bias, self.attention_b[0].weight,
                self.attention_b[0].bias, self.attention_c.weight, self.attention_c.bias)

    def scores_torch(self, x):
        return self.attention_c(self.attention_a(x).mul(self.attention_b(x)))

    def forward(self, x):
        if self.dropout and self.training:
            return self.scores_torch(x), x
        A_raw, _ = gated_attention_pool(x, *self.operands())
        return A_raw.t(), x


class CLAM_SB(nn.Module):
    """Single attention branch (models/model_clam.py:77-243)."""

    size_dict = {"small": [1024, 512, 256], "big": [1024, 512, 384], "benchmark": [384, 512, 256],
                 "conch": [512, 512, 384], "gigapath": [1536, 512, 256], "virchow": [2560, 512, 256]}

    def __init__(self, gate=True, size_arg="small", dropout=False, k_sample=8, n_classes=2,
                 instance_loss_fn=nn.CrossEntropyLoss(), subtyping=False, conch_init=False, conch_freeze=False):
        super().__init__()
        self._build(gate, size_arg, dropout, n_heads=1)
        size = self.size_dict[size_arg]
        self.classifiers = nn.Linear(size[1], n_classes)
        self.instance_classifiers = nn.ModuleList([nn.Linear(size[1], 2) for _ in range(n_classes)])
        self.k_sample, self.instance_loss_fn, self.n_classes, self.subtyping = k_sample, instance_loss_fn, n_classes, subtyping
        initialize_weights(self)
        assert not conch_init, "conch_init loads a checkpoint from the authors' home directory (model_clam.py:108)"

    def _build(self, gate, size_arg, dropout, n_heads):
        size = self.size_dict[size_arg]
        fc = [nn.Linear(size[0], size[1]), nn.ReLU()]
        if dropout:
            fc.append(nn.Dropout(0.25))
        fc.append((Attn_Net_Gated if gate else Attn_Net)(L=size[1], D=size[2], dropout=dropout, n_classes=n_heads))
        self.attention_net = nn.Sequential(*fc)

    def relocate(self):
        self.to(torch.device("cuda"))

    @staticmethod
    def create_positive_targets(length, device):
        return torch.full((length,), 1, device=device).long()

    @staticmethod
    def create_negative_targets(length, device):
        return torch.full((length,), 0, device=device).long()

    def _k(self, A):
        return A.shape[1] - 1 if A.shape[1] < self.k_sample else self.k_sample

    def inst_eval(self, A, h, classifier):
        """In-the-class branch: the k most and k least attended patches as positives / negatives (:140-155)."""
        A = A.view(1, -1) if A.dim() == 1 else A
        k = self._k(A)
        top_p = torch.index_select(h, 0, torch.topk(A, k)[1][-1])
        top_n = torch.index_select(h, 0, torch.topk(-A, k, dim=1)[1][-1])
        targets = torch.cat([self.create_positive_targets(k, h.device), self.create_negative_targets(k, h.device)])
        logits = classifier(torch.cat([top_p, top_n], dim=0))
        return self.instance_loss_fn(logits, targets), torch.topk(logits, 1, dim=1)[1].squeeze(1), targets

    def inst_eval_out(self, A, h, classifier):
        """Out-of-the-class branch: the k most attended patches as negatives (:158-171)."""
        A = A.view(1, -1) if A.dim() == 1 else A
        k = self._k(A)
        top_p = torch.index_select(h, 0, torch.topk(A, k)[1][-1])
        targets = self.create_negative_targets(k, h.device)
        logits = classifier(top_p)
        return self.instance_loss_fn(logits, targets), torch.topk(logits, 1, dim=1)[1].squeeze(1), targets

    def _features(self, h):
        net = self.attention_net
        return net[:-1](h), net[-1]

    def forward_patch_level(self, h):
        h, _ = self._features(h)
        return self.classifiers(h)

    def _instance_terms(self, A_soft, h, label, per_head):
        total, preds, targets = 0.0, [], []
        inst_labels = F.one_hot(label, num_classes=self.n_classes).squeeze()
        for i, clf in enumerate(self.instance_classifiers):
            Ai = A_soft[i] if per_head else A_soft
            if inst_labels[i].item() == 1:
                loss, p, t = self.inst_eval(Ai, h, clf)
            elif self.subtyping:
                loss, p, t = self.inst_eval_out(Ai, h, clf)
            else:
                continue
            preds.extend(p.cpu().numpy())
            targets.extend(t.cpu().numpy())
            total += loss
        if self.subtyping:
            total /= len(self.instance_classifiers)
        return total, preds, targets

    def _bag_logits(self, M):
        return self.classifiers(M)

    def forward_single(self, h, label=None, instance_eval=False, return_features=False, attention_only=False):
        h, attn = self._features(h)
        if attn.dropout and self.training:
            A_raw, M = _pool_with_dropout(h, attn)                    # torch's masks: the one mode off the kernel
        else:
            A_raw, M = gated_attention_pool(h, *attn.operands())      # [K, N], [K, L]: the whole aggregation
        if attention_only:
            return A_raw
        results = {}
        if instance_eval:
            total, preds, targets = self._instance_terms(F.softmax(A_raw, dim=1), h, label, per_head=A_raw.size(0) > 1)
            results = {"instance_loss": total, "inst_labels": np.array(targets), "inst_preds": np.array(preds)}
        logits = self._bag_logits(M)
        Y_hat = torch.topk(logits, 1, dim=1)[1]
        Y_prob = F.softmax(logits, dim=1)
        if return_features:
            results.update({"features": M})
        return logits, Y_prob, Y_hat, A_raw, results

    def forward_batch(self, h, label=None, instance_eval=False, return_features=False, attention_only=False):
        outs = [self.forward_single(h[i], label[i] if instance_eval else None, instance_eval, return_features, attention_only)
                for i in range(h.shape[0])]
        logits, Y_prob, Y_hat, A_raw = (torch.cat([o[j] for o in outs], dim=0) for j in range(4))
        results = {}
        if instance_eval:
            results["instance_loss"] = torch.stack([o[4]["instance_loss"] for o in outs]).sum()
            results["inst_labels"] = np.concatenate([o[4]["inst_labels"] for o in outs])
            results["inst_preds"] = np.concatenate([o[4]["inst_preds"] for o in outs])
        return logits, Y_prob, Y_hat, A_raw, results

    def forward(self, h, label=None, instance_eval=False, return_features=False, attention_only=False):
        if h.dim() == 2:
            return self.forward_single(h, label, instance_eval, return_features, attention_only)
        return self.forward_batch(h, label, instance_eval, return_features, attention_only)


class CLAM_MB(CLAM_SB):
    """One attention head and one 1-logit bag classifier per class (models/model_clam.py:245-326)."""

    def __init__(self, gate=True, size_arg="small", dropout=False, k_sample=8, n_classes=2,
                 instance_loss_fn=nn.CrossEntropyLoss(), subtyping=False, conch_init=False, conch_freeze=False):
        nn.Module.__init__(self)
        self._build(gate, size_arg, dropout, n_heads=n_classes)
        size = self.size_dict[size_arg]
        self.classifiers = nn.ModuleList([nn.Linear(size[1], 1) for _ in range(n_classes)])
        self.instance_classifiers = nn.ModuleList([nn.Linear(size[1], 2) for _ in range(n_classes)])
        self.k_sample, self.instance_loss_fn, self.n_classes, self.subtyping = k_sample, instance_loss_fn, n_classes, subtyping
        initialize_weights(self)

    def forward_patch_level(self, h):
        h, _ = self._features(h)
        return torch.cat([clf(h) for clf in self.classifiers], dim=1)

    def _bag_logits(self, M):
        return torch.cat([self.classifiers[c](M[c]) for c in range(self.n_classes)]).view(1, -1)

    def forward(self, h, label=None, instance_eval=False, return_features=False, attention_only=False):
        return self.forward_single(h, label, instance_eval, return_features, attention_only)
