"""Max-instance MIL baselines under the reference's names and forward contract (SURVEY.md section 8,
row f3; reference models/model_mil.py).  forward(h) -> (top_instance logits [1, C], Y_prob [1, C],
Y_hat, y_probs [N, C], results_dict).  The per-instance classifier is torch (a plain library GEMM);
picking the top instance over the N patches runs on the HIP path (pool_autograd) and autograd flows
through the picked row only, as in the reference.  Parameter names match (`classifier.{0,2}` /
`fc.0`, `classifiers.{c}`), so reference checkpoints load.  TransMIL (plain torch + the restated Nystrom attention of
moc_amd/nystrom.py; parity unpinned: the reference's own class cannot be built without the absent third-party
package) keeps the contract (logits, Y_prob, Y_hat, None, None)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .pool_autograd import top_entry, top_rows


def _plain(m):
    return getattr(m, "module", m)            # the reference indexes `.module` (DataParallel) in one branch


class MIL_fc(nn.Module):
    """Binary max-instance MIL (models/model_mil.py:11-51)."""

    size_dict = {"small": [1024, 512], "benchmark": [384, 512]}

    def __init__(self, gate=True, size_arg="benchmark", dropout=False, n_classes=2, top_k=1):
        super().__init__()
        assert n_classes == 2
        d_in, d_hid = self.size_dict[size_arg]
        layers = [nn.Linear(d_in, d_hid), nn.ReLU()]
        if dropout:
            layers.append(nn.Dropout(0.25))
        layers.append(nn.Linear(d_hid, n_classes))
        self.classifier = nn.Sequential(*layers)
        self.top_k = top_k

    def relocate(self):
        self.classifier.to(torch.device("cuda"))

    def forward(self, h, return_features=False):
        net = _plain(self.classifier)
        if return_features:
            h = net[:-1](h)
            logits = net[-1](h)
        else:
            logits = net(h)                                   # [N, 2]
        y_probs = F.softmax(logits, dim=1)
        pick = top_rows(y_probs[:, 1], self.top_k).view(1,)   # top_k == 1 in every caller (:40)
        top_instance = torch.index_select(logits, 0, pick)
        Y_hat = torch.topk(top_instance, 1, dim=1)[1]
        Y_prob = F.softmax(top_instance, dim=1)
        results = {}
        if return_features:
            results["features"] = torch.index_select(h, 0, pick)
        return top_instance, Y_prob, Y_hat, y_probs, results


class MIL_fc_mc(nn.Module):
    """Multi-class max-instance MIL: one 1-logit head per class, the instance holding the single
    largest class probability represents the bag (models/model_mil.py:54-101)."""

    size_dict = {"small": [1024, 512]}

    def __init__(self, gate=True, size_arg="small", dropout=False, n_classes=2, top_k=1):
        super().__init__()
        assert n_classes > 2
        d_in, d_hid = self.size_dict[size_arg]
        layers = [nn.Linear(d_in, d_hid), nn.ReLU()]
        if dropout:
            layers.append(nn.Dropout(0.25))
        self.fc = nn.Sequential(*layers)
        self.classifiers = nn.ModuleList([nn.Linear(d_hid, 1) for _ in range(n_classes)])
        self.top_k, self.n_classes = top_k, n_classes
        assert self.top_k == 1

    def relocate(self):
        dev = torch.device("cuda")
        self.fc, self.classifiers = self.fc.to(dev), self.classifiers.to(dev)

    def forward(self, h, return_features=False):
        h = self.fc(h)
        heads = _plain(self.classifiers)
        logits = torch.cat([heads[c](h) for c in range(self.n_classes)], dim=1)      # [N, C]
        y_probs = F.softmax(logits, dim=1)
        row, cls = top_entry(y_probs)
        pick = torch.tensor([row], device=h.device)
        top_instance = logits[pick]
        Y_hat = torch.tensor([cls], device=h.device)
        Y_prob = y_probs[pick]
        results = {}
        if return_features:
            results["features"] = torch.index_select(h, 0, pick)
        return top_instance, Y_prob, Y_hat, y_probs, results


class TransLayer(nn.Module):
    """models/model_mil.py:105-122."""

    def __init__(self, norm_layer=nn.LayerNorm, dim=512):
        super().__init__()
        from .nystrom import NystromAttention
        self.norm = norm_layer(dim)
        self.attn = NystromAttention(dim=dim, dim_head=dim // 8, heads=8, num_landmarks=dim // 2, pinv_iterations=6,
                                     residual=True, dropout=0.1)

    def forward(self, x):
        return x + self.attn(self.norm(x))


class PPEG(nn.Module):
    """models/model_mil.py:125-139: depth-wise 7 / 5 / 3 convolutions over the patch tokens laid out as a square."""

    def __init__(self, dim=512):
        super().__init__()
        self.proj = nn.Conv2d(dim, dim, 7, 1, 7 // 2, groups=dim)
        self.proj1 = nn.Conv2d(dim, dim, 5, 1, 5 // 2, groups=dim)
        self.proj2 = nn.Conv2d(dim, dim, 3, 1, 3 // 2, groups=dim)

    def forward(self, x, H, W):
        B, _, C = x.shape
        cls_token, feat_token = x[:, 0], x[:, 1:]
        cnn_feat = feat_token.transpose(1, 2).view(B, C, H, W)
        x = self.proj(cnn_feat) + cnn_feat + self.proj1(cnn_feat) + self.proj2(cnn_feat)
        x = x.flatten(2).transpose(1, 2)
        return torch.cat((cls_token.unsqueeze(1), x), dim=1)


class TransMIL(nn.Module):
    """models/model_mil.py:142-273: forward(data) -> (logits [B, C], Y_prob, Y_hat, None, None), same modules and
    parameter names (`pos_layer`, `_fc1`, `cls_token`, `layer1`, `layer2`, `norm`, `_fc2`).  Its attention is the
    third-party `nystrom_attention` package, absent here AND in the reference tree (the reference's own import fails in
    this image): the layer is restated in moc_amd/nystrom.py from the published algorithm, and since no reference
    output can be produced, **parity of this class is unpinned** (checked: shapes, the 5-tuple, the padding rule, the
    attention's exact-softmax limit; not checked: a reference number).  Plain torch: a signature shim, row f3."""

    size_dict = {"small": 1024, "big": 1024, "benchmark": 384, "conch": 512, "gigapath": 1536, "virchow": 2560}

    def __init__(self, n_classes, size_arg="small", **kwargs):
        super().__init__()
        size = self.size_dict[size_arg]
        self.pos_layer = PPEG(dim=512)
        self._fc1 = nn.Sequential(nn.Linear(size, 512), nn.ReLU())
        self.cls_token = nn.Parameter(torch.randn(1, 1, 512))
        self.n_classes = n_classes
        self.layer1 = TransLayer(dim=512)
        self.layer2 = TransLayer(dim=512)
        self.norm = nn.LayerNorm(512)
        self._fc2 = nn.Linear(512, self.n_classes)

    def _tokens(self, data):
        """fc1, wrap-around padding to a square, class token, layer 1, PPEG, layer 2 (:236-266)."""
        if len(data.shape) == 2:
            data = data.unsqueeze(0)
        h = self._fc1(data.float())                                       # [B, n, 512]
        H = h.shape[1]
        side = int(np.ceil(np.sqrt(H)))
        h = torch.cat([h, h[:, :side * side - H, :]], dim=1)              # pad with the first rows again
        h = torch.cat((self.cls_token.expand(h.shape[0], -1, -1).to(h.device), h), dim=1)
        h = self.layer1(h)
        h = self.pos_layer(h, side, side)
        return self.layer2(h), data.shape[1]

    def forward_patch_level(self, data):
        h, n = self._tokens(data)
        return self._fc2(h).squeeze(0)[1:n + 1]                           # [n, n_classes], no final norm (:198-200)

    def forward(self, data, **kwargs):
        h, _ = self._tokens(data)
        logits = self._fc2(self.norm(h)[:, 0])                            # [B, n_classes]
        return logits, F.softmax(logits, dim=1), torch.argmax(logits, dim=1), None, None
