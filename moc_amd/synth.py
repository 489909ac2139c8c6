"""Synthetic slides for tests and bench.

Real CONCH embeddings / zero-shot weights are not available offline, so bags
are unit-norm Gaussian rows with a planted class signal, and the classifier
bank is random unit-norm columns with W_ext[:, :C] == W (as the reference's
prompt files imply, main_moc.py:161-202).

Two generators:
  * numpy Philox (counter based, host) -- bit-stable, used by tests and the
    committed golden fixtures, which store only seeds + expected outputs;
  * torch on-device -- used by bench.py for the large configs (bags are then
    copied to the host for the CPU baseline, so both legs see the same bytes).
"""
from __future__ import annotations

import numpy as np
import torch

PLANT_FRACTION = 0.05      # rows carrying the slide's class direction
BACKGROUND_FRACTION = 0.20  # rows carrying one of the 4 background directions
PLANT_GAIN = 0.3


def _rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=int(seed)))


def make_bank(seed: int, D: int, C: int, n_bg: int = 4):
    """(W [D,C], W_ext [D,C+n_bg]) fp32, unit-norm columns, shared first C."""
    g = _rng(seed).standard_normal((D, C + n_bg)).astype(np.float32)
    g /= np.linalg.norm(g, axis=0, keepdims=True)
    W_ext = torch.from_numpy(np.ascontiguousarray(g))
    return W_ext[:, :C].contiguous(), W_ext


def make_bag(seed: int, N: int, D: int, W_ext: torch.Tensor, C: int, label: int,
             scale: float = 1.0, confusion: float = 0.0, gain: float = PLANT_GAIN) -> torch.Tensor:
    """One slide [N,D] fp32 on the host.  `confusion` > 0 gives that share of the planted rows
    the direction of a random OTHER class (a harder task: AUC < 1); 0 leaves the stream of random
    numbers exactly as it was, so fixtures made without it do not change."""
    r = _rng(seed)
    x = r.standard_normal((N, D)).astype(np.float32)
    x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12)
    u = r.random(N)
    bank = W_ext.numpy()
    fg = u < PLANT_FRACTION
    if confusion > 0.0:
        r2 = _rng(seed + 7919)
        wrong = fg & (r2.random(N) < confusion)
        other = (label + 1 + r2.integers(0, C - 1, size=N)) % C
        x[fg & ~wrong] += gain * bank[:, label][None, :]
        x[wrong] += gain * bank[:, other[wrong]].T
    else:
        x[fg] += gain * bank[:, label][None, :]
    bg = (u >= PLANT_FRACTION) & (u < PLANT_FRACTION + BACKGROUND_FRACTION)
    which = r.integers(C, bank.shape[1], size=N)
    x[bg] += PLANT_GAIN * bank[:, which[bg]].T
    x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12)
    if scale != 1.0:
        x *= np.float32(scale)
    return torch.from_numpy(x)


def bag_sizes(seed: int, n_slides: int, mean_n: int, fixed: bool = True,
              sigma: float = 0.4, lo: int = 2000, hi: int = 60000):
    if fixed:
        return [int(mean_n)] * n_slides
    r = _rng(seed)
    n = np.exp(np.log(mean_n) + sigma * r.standard_normal(n_slides))
    return [int(v) for v in np.clip(n, lo, hi)]


def make_slide_set(base_seed: int, sizes, D: int, W_ext: torch.Tensor, C: int, confusion: float = 0.0,
                   gain: float = PLANT_GAIN):
    """Host bags + round-robin labels (every class present when len>=C)."""
    labels = [i % C for i in range(len(sizes))]
    bags = [make_bag(base_seed + i, n, D, W_ext, C, labels[i], confusion=confusion, gain=gain)
            for i, n in enumerate(sizes)]
    return bags, labels


def make_bag_device(seed: int, N: int, D: int, W_ext: torch.Tensor, C: int, label: int,
                    device, dtype=torch.float32) -> torch.Tensor:
    """Same recipe generated on `device` (not bit-equal to make_bag)."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    x = torch.randn(N, D, device=device, generator=g)
    x = x / x.norm(dim=1, keepdim=True).clamp_min(1e-12)
    u = torch.rand(N, device=device, generator=g)
    bank = W_ext.to(device)
    which = torch.randint(C, bank.shape[1], (N,), device=device, generator=g)
    fg = (u < PLANT_FRACTION).unsqueeze(1)
    bg = ((u >= PLANT_FRACTION) & (u < PLANT_FRACTION + BACKGROUND_FRACTION)).unsqueeze(1)
    x = x + PLANT_GAIN * fg * bank[:, label].unsqueeze(0)
    x = x + PLANT_GAIN * bg * bank[:, which].t()
    x = x / x.norm(dim=1, keepdim=True).clamp_min(1e-12)
    return x.to(dtype)
