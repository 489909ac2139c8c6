"""Multi-GPU for the MOC path: one process per GPU, torch.distributed over RCCL
("nccl" backend on ROCm) -- or gloo on CPU for the tests.

What shards:
  * evaluation / zs_evaluation: slides are independent -> each rank evaluates its
    shard, one all_gather of the [n_local, C] pooled logits (+ losses), AUC on
    every rank from the gathered matrix.  No other collective.
  * train: phase A (everything in slide_process) shards with the slides.  The
    reference's one-Adam-step-per-slide recurrence does not; the data-parallel
    form north_star names is synchronous minibatch SGD over G slides, one per
    rank: every rank computes the gradient of its slide's loss, ONE all-reduce
    of the flat [33,092] fp32 gradient (132 KB; latency-bound on xGMI, so a
    single fused buffer, never per-tensor), mean, identical Adam step on every
    rank.  With G == 1 this is exactly the reference's loop.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_indices(n_items: int, rank: int, world: int, sizes=None):
    """Items of this rank.  With `sizes` (rows per slide) the split is greedy
    longest-first so ranks get similar bytes; otherwise round robin.  Every item
    lands on exactly one rank and the result is deterministic."""
    if sizes is None:
        return list(range(rank, n_items, world))
    order = sorted(range(n_items), key=lambda i: (-int(sizes[i]), i))
    load = [0] * world
    mine = []
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        load[r] += int(sizes[i])
        if r == rank:
            mine.append(i)
    return sorted(mine)


class FlatGrads:
    """The four gradient tensors as views of ONE flat buffer, so the meta-gradient
    all-reduce is a single collective."""

    def __init__(self, params, device):
        self.numels = [p.numel() for p in params]
        self.flat = torch.zeros(sum(self.numels), dtype=torch.float32, device=device)
        self.views, o = [], 0
        for p, n in zip(params, self.numels):
            self.views.append(self.flat[o:o + n].view_as(p))
            o += n


def allreduce_mean_(flat: torch.Tensor, group=None) -> float:
    """Sum-all-reduce in place; returns the scale (1/world) the caller applies
    (fused into the Adam kernel's grad_scale on the GPU path)."""
    if not dist.is_initialized():
        return 1.0
    world = dist.get_world_size(group)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)      # also at world 1: same call path as N > 1
    return 1.0 / world


def gather_rows(local: torch.Tensor, counts, group=None) -> torch.Tensor:
    """all_gather of per-rank [n_r, C] matrices with ragged n_r (counts known on
    every rank): returns the concatenation in rank order."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    width = local.shape[1:]
    m = max(counts)
    pad = torch.zeros((m, *width), dtype=local.dtype, device=local.device)
    pad[: local.size(0)] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[:c] for o, c in zip(out, counts)], 0)


def unshard(values_by_rank, index_lists, n_items):
    """Inverse of shard_indices for gathered per-item rows: rank r's k-th row is
    item index_lists[r][k]."""
    total = torch.cat(values_by_rank, 0) if isinstance(values_by_rank, (list, tuple)) else values_by_rank
    perm = [i for lst in index_lists for i in lst]
    assert sorted(perm) == list(range(n_items)), "shards do not partition the items"
    out = torch.empty_like(total)
    out[torch.tensor(perm, device=total.device)] = total
    return out


# --------------------------------------------------------------------------- GPU path
def train_dp(model, loader, optimizer, device, args, group=None):
    """Synchronous data-parallel pass: this rank's loader holds ITS slides; step t uses the
    t-th slide of every rank.  All ranks must hold the same number of visits."""
    from . import engine, main_moc as M
    model.train()
    use = engine.train_use_bits(args.discard_classifiers)
    if isinstance(loader, M.ResidentBags):
        batch, lab, bank = M._resident_pass_setup(loader, device, args)
        sizes = batch.sizes
    else:
        X, sizes, x_starts, labels = M._collect(loader, device, args)
        mask_all, _ = engine.draw_row_masks(sum(sizes))
        masks, o = [], 0
        for n in sizes:
            masks.append(mask_all[o:o + n])
            o += n
        bank = M._bank_for(X, device)
        batch = M._sub_batch(X, sizes, x_starts, list(range(len(sizes))), bank.C, bank.Ce, args.topj, args.topk,
                             args.discard_classifiers, masks)
        lab = torch.tensor(labels, dtype=torch.int64).to(device, non_blocking=True)
    meta = engine.MetaState(model, optimizer, need_grads=True)
    fg = FlatGrads(meta.params, device)
    # point the C ABI's gradient outputs at the flat buffer: the all-reduce is ONE collective
    for name, v in zip(("g_W1", "g_b1", "g_W2", "g_b2"), fg.views):
        setattr(meta.c, name, v.data_ptr())
    batch.phase_a(bank)
    for t in range(len(sizes)):
        engine.train_grad(batch, meta, lab, t, use)
        scale = allreduce_mean_(fg.flat, group)
        engine.adam_step(meta, scale)
    train_dp.last = (batch, lab, fg)


def evaluation_dp(model, loader, device, args, all_labels, my_ids, index_lists, group=None):
    """Slide-sharded evaluation: `loader` iterates this rank's slides (my_ids of the global
    list); returns the reference's metrics dict over ALL slides on every rank."""
    from . import main_moc as M
    model.eval()
    with torch.no_grad():
        pooled, labels, losses = M._eval_pass(loader, device, args, "eval", model=model)
    counts = [len(l) for l in index_lists]
    both = torch.cat([pooled.to(device), torch.tensor(losses, dtype=torch.float32, device=device).unsqueeze(1)], 1)
    allv = unshard(gather_rows(both, counts, group), index_lists, len(all_labels)).cpu()
    n = len(all_labels)
    return M._metrics(allv[:, :-1].contiguous(), list(all_labels), allv[:, -1].tolist(), n, n, args)
