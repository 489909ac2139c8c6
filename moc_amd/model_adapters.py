"""CLIP-Adapter / Tip-Adapter / MoE-adapter / AMU heads over CONCH patch features, under the
reference's class names, constructor arguments, parameter names and forward signatures (SURVEY.md
section 8, row f3; reference models/model_adapters.py).  Every forward ends in the same top-j mean
pooling over the N patches (reference :173-183 and its copies): that runs on the HIP path
(pool_autograd.topk_mean_pool -> moc_topk_mean) with autograd through the pooled rows; the adapter
layers themselves are plain torch GEMMs.  forward(feat [N, c_in]) -> pooled logits [1, C]."""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .pool_autograd import topk_mean_pool


def _unit(x):
    return x / x.norm(dim=-1, keepdim=True)


def _kaiming(net):
    for i in (0, 2):
        nn.init.kaiming_normal_(net[i].weight, a=math.sqrt(5))


def _bottleneck(c_in, reduction, init=True):
    """Linear -> ReLU -> Linear -> ReLU, bias-free.  `init=False` leaves the re-initialisation to the
    caller: the constructors below draw from the RNG in the reference's order, so the same seed gives the
    same parameters."""
    net = nn.Sequential(nn.Linear(c_in, c_in // reduction, bias=False), nn.ReLU(inplace=True),
                        nn.Linear(c_in // reduction, c_in, bias=False), nn.ReLU(inplace=True))
    if init:
        _kaiming(net)
    return net


class _PooledHead(nn.Module):
    """Shared tail: cosine logits against the frozen classifier tensor and top-j mean pooling."""

    def topj_pooling(self, logits, topj=10):
        return topk_mean_pool(logits, topj)

    def _zero_shot(self, feat, topj):
        return self.topj_pooling(_unit(feat) @ self.classifier, topj=topj)


class Linear_Adapter(nn.Module):
    """One bias-free linear map feat -> class logits, optionally initialised from labelled sample
    features (the Tip-Adapter cache as a weight matrix; reference :77-97)."""

    def __init__(self, feat_dim, class_num, sample_features=None):
        super().__init__()
        self.fc = nn.Linear(feat_dim, class_num, bias=False)
        if sample_features is None:
            nn.init.kaiming_normal_(self.fc.weight, a=math.sqrt(5))
            return
        feats, labels = sample_features[0], sample_features[1]
        feats = (feats - feats.mean()) / feats.std()
        w = torch.zeros(feat_dim, class_num, device=feats.device)
        for f, y in zip(feats, labels):
            w[:, int(y)] += f
        self.fc.weight.data = (w / (len(labels) / class_num)).t()

    def forward(self, feat):
        return self.fc(feat)


def uncertainty(logits, type, power):
    """Per-row confidence factor of the zero-shot logits (reference :100-145)."""
    p = F.softmax(logits, dim=-1)
    if type == "none":
        return torch.tensor(1.0)
    if type == "entropy":
        ent = -(p * torch.log2(p)).sum(-1, keepdim=True) / torch.log2(torch.tensor(p.shape[-1]).float())
        return (ent * power).exp()
    top = p.max(dim=-1, keepdim=True).values
    if type == "energy":
        tau = 2
        return 1.0 / (tau * (torch.log(torch.exp((p - top) / tau).sum(-1, keepdim=True)) + top)) ** power
    if type == "max":
        return 1.0 / top ** power
    if type == "max-min":
        return 1.0 / (top - p.min(dim=-1, keepdim=True).values) ** power
    if type == "var":
        return torch.std(p, dim=-1, keepdim=True)
    if type == "top5":
        t5 = p.topk(5, dim=-1).values
        return 1.0 / (t5[:, 0] - t5[:, -1]).unsqueeze(-1) ** power
    if type == "moment":
        z = (p - p.mean(-1, keepdim=True)) / torch.std(p, dim=-1, keepdim=True)
        return 1 / ((z ** 4).mean(-1, keepdim=True) / 250) ** power
    raise RuntimeError("Invalid uncertainty type.")


class Conch_CLIP_Ada(_PooledHead):
    """Residual bottleneck adapter on the features (reference :148-215)."""

    def __init__(self, c_in=512, reduction=4, num_classes=2, classifier_tensor=None, clip_ratio=0.1, topj=10):
        super().__init__()
        self.adapter = _bottleneck(c_in, reduction)
        self.topj, self.classifier, self.num_classes, self.clip_ratio = topj, classifier_tensor, num_classes, clip_ratio

    def forward(self, feat):
        mixed = self.adapter(feat) * self.clip_ratio + feat * (1 - self.clip_ratio)
        return self.topj_pooling(_unit(mixed) @ self.classifier, topj=self.topj)

    def forward_disable_ada(self, feat):
        return self._zero_shot(feat, self.topj)


class Conch_TIP_Ada(_PooledHead):
    """Linear adapter on the logits (reference :218-250).  `forward` normalises `feat` IN PLACE, as the
    reference does (:238)."""

    def __init__(self, c_in=512, num_classes=2, classifier_tensor=None, sample_features=None, clip_ratio=0.1):
        super().__init__()
        self.adapter = Linear_Adapter(c_in, num_classes, sample_features)
        self.classifier, self.num_classes, self.clip_ratio = classifier_tensor, num_classes, clip_ratio

    def forward(self, feat):
        feat /= feat.norm(dim=-1, keepdim=True)
        logits = self.adapter(feat) * self.clip_ratio + (feat @ self.classifier) * (1 - self.clip_ratio)
        return self.topj_pooling(logits, topj=10)

    def forward_disable_ada(self, feat):
        return self._zero_shot(feat, 10)


def load_balancing_loss_func(router_probs: torch.Tensor, expert_indices: torch.Tensor):
    """Switch-Transformer auxiliary loss, eqs. (4)-(6) of arXiv:2101.03961 (reference :253-289):
    E^2 * mean_e( fraction of tokens routed to e  *  mean router probability of e )."""
    n_exp = router_probs.shape[-1]
    idx = expert_indices.to(torch.int64)
    if idx.dim() == 2:
        idx = idx.unsqueeze(2)
    routed = F.one_hot(idx, n_exp).max(dim=-2).values.to(torch.float32)
    return (routed.mean(dim=-2) * router_probs.mean(dim=-2)).mean() * n_exp ** 2


class SwitchGate(nn.Module):
    """Softmax router over the experts, optionally hard top-1 (reference :292-327)."""

    def __init__(self, c_in=512, num_experts=3, use_switch_gate=False, use_balance_loss=False,
                 init_tensor=None, router_trainable=True):
        super().__init__()
        self.dim, self.num_experts = c_in, num_experts
        self.gate = nn.Linear(c_in, num_experts, bias=False)
        self.use_switch_gate, self.use_balance_loss = use_switch_gate, use_balance_loss
        self.init_tensor, self.router_trainable = init_tensor, router_trainable
        if init_tensor is not None:
            self.gate.weight.data = init_tensor.transpose(0, 1)
        else:
            nn.init.kaiming_normal_(self.gate.weight, a=math.sqrt(5))
        if not router_trainable:
            self.gate.weight.requires_grad = False

    def forward(self, x):
        scores = F.softmax(self.gate(x), dim=-1)
        if not self.use_switch_gate:
            return scores, None
        _, top = scores.topk(1, dim=-1)
        scores = scores * torch.zeros_like(scores).scatter_(-1, top, 1)
        if not self.use_balance_loss:
            return scores, None
        return scores, load_balancing_loss_func(scores.unsqueeze(0), top.squeeze().unsqueeze(0))


class Conch_MOE_CLIP_Ada(_PooledHead):
    """`ada_num` bottleneck adapters mixed per patch by a router (reference :330-405)."""

    def __init__(self, c_in=512, reduction=4, ada_num=5, topj=10, classifier_tensor=None, clip_ratio=0.1,
                 use_switch_gate=False, use_balance_loss=False, router_tensor=None, router_trainable=True):
        super().__init__()
        assert ada_num > 1
        self.ada_num, self.topj, self.init_router = ada_num, topj, router_tensor
        self.use_switch_gate, self.use_balance_loss, self.router_trainable = use_switch_gate, use_balance_loss, router_trainable
        for i in range(ada_num):
            setattr(self, f"adapter_{i}", _bottleneck(c_in, reduction, init=False))
        if router_tensor is None:
            self.ada_router = SwitchGate(c_in, ada_num, use_switch_gate, use_balance_loss, None, router_trainable)
        else:
            assert not use_balance_loss and not use_switch_gate
            self.ada_router = SwitchGate(c_in, ada_num, False, False, router_tensor, router_trainable)
        self.classifier = classifier_tensor
        self.clip_ratio = clip_ratio / ada_num
        self.reset_adapter_weight()

    def reset_adapter_weight(self):
        for i in range(self.ada_num):
            _kaiming(getattr(self, f"adapter_{i}"))

    def forward(self, feat):
        feat = _unit(feat)
        weight, balance = self.ada_router(feat)                                        # [N, E]
        experts = torch.stack([getattr(self, f"adapter_{i}")(feat) for i in range(self.ada_num)], dim=-1)
        mixed = _unit((experts * weight.unsqueeze(-2)).sum(-1))
        pooled = self.topj_pooling(_unit(mixed * self.clip_ratio + feat * (1 - self.clip_ratio)) @ self.classifier,
                                   topj=self.topj)
        return (pooled, balance) if self.use_balance_loss else pooled

    def forward_disable_ada(self, feat):
        return self._zero_shot(feat, self.topj)


class Conch_AMUVanilla_Ada(_PooledHead):
    """Feature adapter + auxiliary-feature linear adapter weighted by an uncertainty factor
    (reference :408-497).  forward(feat, aux_feat) -> (pooled, pooled_aux)."""

    def __init__(self, c_in=512, c_in_aux=1024, reduction=4, num_classes=2, classifier_tensor=None,
                 clip_ratio=0.1, aux_ratio=0.1, uncertainty_type="none", uncertainty_power=1.0):
        super().__init__()
        self.adapter = _bottleneck(c_in, reduction, init=False)
        self.aux_adapter = Linear_Adapter(c_in_aux, num_classes, None)
        self.classifier, self.num_classes = classifier_tensor, num_classes
        self.clip_ratio, self.aux_ratio = clip_ratio, aux_ratio
        self.uncertainty_type, self.uncertainty_power = uncertainty_type, uncertainty_power
        _kaiming(self.adapter)

    def forward(self, feat, aux_feat):
        feat = _unit(feat)
        clip_logits = feat @ self.classifier
        ada_logits = _unit(self.adapter(feat).squeeze()) @ self.classifier
        aux_logits = self.aux_adapter(_unit(aux_feat)).squeeze()
        factor = uncertainty(clip_logits.float(), self.uncertainty_type, self.uncertainty_power)
        logits = (ada_logits * self.clip_ratio + aux_logits * self.aux_ratio * factor
                  + clip_logits * (1 - self.clip_ratio - self.aux_ratio))
        return self.topj_pooling(logits, topj=10), self.topj_pooling(aux_logits, topj=10)

    def forward_disable_ada(self, feat, aux_feat):
        return self._zero_shot(feat, 10)


class Conch_AMUTip_Ada(_PooledHead):
    """Two linear adapters (features, auxiliary features) on the logits (reference :500-545).
    `forward` normalises both inputs IN PLACE, as the reference does (:530-531)."""

    def __init__(self, c_in=512, c_in_aux=1024, num_classes=2, classifier_tensor=None, sample_features=None,
                 aux_sample_features=None, clip_ratio=0.1, aux_ratio=0.1):
        super().__init__()
        self.adapter = Linear_Adapter(c_in, num_classes, sample_features)
        self.aux_adapter = Linear_Adapter(c_in_aux, num_classes, aux_sample_features)
        self.classifier, self.num_classes = classifier_tensor, num_classes
        self.clip_ratio, self.aux_ratio = clip_ratio, aux_ratio

    def forward(self, feat, aux_feat):
        feat /= feat.norm(dim=-1, keepdim=True)
        aux_feat /= aux_feat.norm(dim=-1, keepdim=True)
        logits = (self.adapter(feat) * self.clip_ratio + self.aux_adapter(aux_feat) * self.aux_ratio
                  + (feat @ self.classifier) * (1 - self.clip_ratio - self.aux_ratio))
        return self.topj_pooling(logits, topj=10)

    def forward_disable_ada(self, feat, aux_feat):
        return self._zero_shot(feat, 10)
