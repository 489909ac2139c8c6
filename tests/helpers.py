"""Shared helpers for the parity tests (test-side only)."""
from __future__ import annotations

import os
import types

import numpy as np
import torch

from moc_amd import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SELECTORS = ("topk", "delta_softmax", "delta_diff", "bottomk")

# fp32 tolerance stated by BASELINE.json:north_star ("within 1e-4 fp32")
ATOL = 1e-4
# a row may legitimately enter/leave a top-j set when its key is this close to the
# j-th key: fp32 dot products summed in a different order differ by ~1e-7 relative
KEY_BAND = 2e-6


def golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


def discard_from_mask(dmask: int):
    return [n for k, n in enumerate(SELECTORS) if dmask >> k & 1]


def unpack_mask(bits, n):
    return torch.from_numpy(np.unpackbits(bits)[:n].astype(bool))


def unpack_masks(bits, sizes):
    out, off = [], 0
    for n in sizes:
        nb = (n + 7) // 8
        out.append(unpack_mask(bits[off:off + nb], n))
        off += nb
    return out


def selector_keys(logits_ext: torch.Tensor, C: int):
    """Key columns the four selectors rank rows by (larger = selected first).
    Returns dict name -> [N, ncols] fp32."""
    lg = logits_ext[:, :C]
    two = torch.topk(lg, 2, dim=1)[0]
    return {
        "top": lg,
        "softmax": torch.softmax(lg, dim=1),
        "gap": (two[:, 0] - two[:, 1]).abs().unsqueeze(1),
        "lowbg": (-logits_ext[:, C:].sum(dim=1)).unsqueeze(1),
    }


def assert_topj_set(got_idx, exp_idx, key_col: torch.Tensor, band=KEY_BAND, what=""):
    """got/exp: 1-D index collections for ONE key column.  They must agree except
    for rows whose key lies within `band` of the selection boundary."""
    got, exp = set(int(i) for i in got_idx), set(int(i) for i in exp_idx)
    assert len(got) == len(exp), f"{what}: |got|={len(got)} |exp|={len(exp)}"
    if got == exp:
        return
    boundary = min(float(key_col[i]) for i in exp)
    scale = max(1.0, abs(boundary))
    for i in got ^ exp:
        d = abs(float(key_col[i]) - boundary)
        assert d <= band * scale, f"{what}: row {i} differs, key margin {d:.3e}"


def ambiguous_rows(keys: dict, j: int, band=KEY_BAND):
    """Rows whose membership in some selector's top-j is within the tie band."""
    amb = set()
    for name, K in keys.items():
        n = K.size(0)
        if n <= j:
            continue
        for c in range(K.size(1)):
            col = K[:, c]
            srt = torch.sort(col, descending=True)[0]
            b = float(srt[j - 1])
            near = (col - b).abs() <= band * max(1.0, abs(b))
            amb.update(torch.nonzero(near).flatten().tolist())
    return amb


def make_args(C, j, K, discard=()):
    return types.SimpleNamespace(disable_tqdm=True, n_classes=C, topj=j, topk=K,
                                 discard_classifiers=list(discard), pretrain="conch",
                                 ablation_study="none")


class ListDataset:
    """Dataset protocol of the reference's split objects (dataset_generic.py:380-393)."""

    def __init__(self, bags, labels, repeat_num=None):
        self.bags, self.labels, self.repeat_num = bags, labels, repeat_num

    def real_len(self):
        return len(self.bags)

    def __len__(self):
        return self.repeat_num if self.repeat_num else len(self.bags)

    def __getitem__(self, idx):
        if idx >= len(self):
            raise IndexError
        k = idx % len(self.bags)
        x = self.bags[k]
        return x, self.labels[k], np.zeros((x.size(0), 2), dtype=np.int64), f"slide_{k}.h5"


class ListLoader:
    """batch_size=1 default-collate stand-in for torch DataLoader (main_moc.py:290-293)."""

    def __init__(self, bags, labels, repeat_num=None):
        self.dataset = ListDataset(bags, labels, repeat_num)

    def __iter__(self):
        ds = self.dataset
        for i in range(len(ds)):
            x, y, coords, path = ds[i]
            yield x.unsqueeze(0), torch.tensor([y]), torch.from_numpy(coords).unsqueeze(0), [path]

    def __len__(self):
        return len(self.dataset)


def seeded_senet(oracle, seed):
    torch.manual_seed(seed)
    return oracle.Senet(512, 4)


def flat_params(model):
    return torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy()


def flat_grads(model):
    return torch.cat([p.grad.detach().reshape(-1) for p in model.parameters()]).cpu().numpy()


def flat_state(opt, key):
    return torch.cat([opt.state[p][key].reshape(-1) for g in opt.param_groups
                      for p in g["params"]]).cpu().numpy()


def bank_and_bag(seed, N, C, label, D=512):
    W, We = synth.make_bank(seed, D, C)
    return W, We, synth.make_bag(seed + 500, N, D, We, C, label=label)


def assert_adam_params_close(got, exp, v_exp, step, grad_noise, lr=1e-3, base=2e-6, what=""):
    """Parameters after `step` Adam steps.  Adam's update is lr*m_hat/(sqrt(v_hat)+eps):
    for an element whose gradient is comparable to the gradient noise between two
    correct fp32 implementations (`grad_noise`, absolute), the update is sign-like and
    may differ by up to lr per step; elsewhere the difference scales with
    grad_noise/|g|.  Bound each element accordingly."""
    got, exp, v_exp = (np.asarray(a, dtype=np.float64) for a in (got, exp, v_exp))
    v_hat = v_exp / (1.0 - 0.999 ** step)
    allowed = base + lr * step * np.minimum(1.0, 4.0 * grad_noise / (np.sqrt(v_hat) + 1e-8))
    bad = np.abs(got - exp) > allowed
    assert not bad.any(), (f"{what}: {int(bad.sum())} params outside the Adam noise bound; worst "
                           f"{np.abs(got - exp)[bad].max():.3e} (allowed {allowed[bad].min():.3e})")
