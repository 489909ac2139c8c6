"""Max-instance MIL baselines under the reference's names and forward contract (SURVEY.md section 8,
row f3; reference models/model_mil.py).  forward(h) -> (top_instance logits [1, C], Y_prob [1, C],
Y_hat, y_probs [N, C], results_dict).  The per-instance classifier is torch (a plain library GEMM);
picking the top instance over the N patches runs on the HIP path (pool_autograd) and autograd flows
through the picked row only, as in the reference.  Parameter names match (`classifier.{0,2}` /
`fc.0`, `classifiers.{c}`), so reference checkpoints load."""
from __future__ import annotations

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


class TransMIL(nn.Module):
    """models/model_mil.py:142-273: forward(data) -> (logits, Y_prob, Y_hat, None, None).  Its layers are
    Nystrom attention from the third-party `nystrom_attention` package, which neither this image nor
    the reference tree holds (the reference's own import of it fails here); the contract is recorded
    and construction fails loudly rather than substituting a different attention."""

    def __init__(self, n_classes, *args, **kwargs):
        super().__init__()
        try:
            import nystrom_attention  # noqa: F401
        except ImportError as e:
            raise ImportError("TransMIL needs the `nystrom_attention` package (absent here; "
                              "reference: models/model_mil.py:6)") from e
        raise NotImplementedError("TransMIL is outside the MOC hot path (SURVEY.md section 8, f3: signature only)")

    def forward(self, data, **kwargs):  # pragma: no cover
        raise NotImplementedError
