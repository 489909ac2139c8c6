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


class DirectRccl:
    """ncclAllReduce called straight from the host loop (ctypes into the librccl.so torch already
    loaded), on the caller's stream.  torch.distributed.all_reduce costs ~25 us of Python/dispatcher
    time per call; with one 132 KB collective per meta-step that IS the step time.  The communicator
    is this class's own (ncclCommInitRank; the unique id travels through torch.distributed), is
    checked once against torch's all-reduce, and anything going wrong at set-up falls back to torch."""

    NCCL_FLOAT32, NCCL_SUM = 7, 0

    def __init__(self, device, group=None):
        import ctypes as C
        import os
        self.C = C
        self.ok = False
        self.comm = C.c_void_p()
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        try:
            path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            self.lib = C.CDLL(path)
            class UniqueId(C.Structure):                      # ncclUniqueId: 128 opaque bytes, passed BY VALUE
                _fields_ = [("internal", C.c_byte * 128)]
            uid = UniqueId()
            if self.rank == 0:
                assert self.lib.ncclGetUniqueId(C.byref(uid)) == 0
            box = [bytes(uid)]
            dist.broadcast_object_list(box, src=0, group=group)
            uid = UniqueId.from_buffer_copy(box[0])
            self.lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
            self.lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
            torch.cuda.set_device(device)
            rc = self.lib.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank)
            assert rc == 0, f"ncclCommInitRank rc={rc}"
            # self-check against torch's collective on a known vector
            a = torch.arange(1, 1025, dtype=torch.float32, device=device) * (self.rank + 1)
            b = a.clone()
            self.all_reduce_(a)
            dist.all_reduce(b, group=group)
            torch.cuda.synchronize()
            assert torch.equal(a, b), "direct RCCL all-reduce disagrees with torch.distributed"
            self.ok = True
        except Exception as e:  # noqa: BLE001  (set-up only; the data path never swallows errors)
            import warnings
            warnings.warn(f"direct RCCL unavailable ({e}); using torch.distributed.all_reduce")

    def all_reduce_(self, flat: torch.Tensor):
        rc = self.lib.ncclAllReduce(flat.data_ptr(), flat.data_ptr(), flat.numel(), self.NCCL_FLOAT32, self.NCCL_SUM,
                                    self.comm, torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError(f"ncclAllReduce failed rc={rc}")

    def close(self):
        if self.ok and self.comm:
            self.lib.ncclCommDestroy.argtypes = [self.C.c_void_p]
            self.lib.ncclCommDestroy(self.comm)
            self.comm, self.ok = self.C.c_void_p(), False


class P2pExchange:
    """The node-local one-shot exchange of include/moc_hip.h (moc_p2p_*): every rank maps its peers'
    fine-grained receive buffers (hipIpc handles travel through torch.distributed) and the step
    kernel itself pushes, flags, waits and sums.  Set-up is collective and ends with a self-check
    against torch.distributed's all-reduce; `ok` is the same on every rank (all must agree)."""

    def __init__(self, device, n_par: int, group=None):
        import ctypes as C
        import socket
        from ._lib import lib, check
        self.C, self.lib, self.check = C, lib(), check
        self.handle = C.c_void_p()
        self.ok = False
        self.why = ""
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_par = int(n_par)
        mine_ok, blob = 1, b""
        try:
            assert self.world <= 8, "more than 8 ranks"
            torch.cuda.set_device(device)
            check(self.lib.moc_p2p_create(self.world, self.rank, self.n_par, C.byref(self.handle)), "moc_p2p_create")
            nb = self.lib.moc_p2p_handle_bytes()
            buf = C.create_string_buffer(nb)
            check(self.lib.moc_p2p_export(self.handle, buf), "moc_p2p_export")
            blob = buf.raw
        except Exception as e:  # noqa: BLE001 (set-up only)
            mine_ok, self.why = 0, f"create/export: {e}"
        box = [None] * self.world
        dist.all_gather_object(box, (mine_ok, socket.gethostname(), blob, self.why), group=group)
        if not all(b[0] for b in box):
            self.why = "; ".join(f"rank {i}: {b[3]}" for i, b in enumerate(box) if not b[0])
        elif len({b[1] for b in box}) != 1:
            self.why = "ranks span more than one node"
        else:
            try:
                check(self.lib.moc_p2p_connect(self.handle, b"".join(b[2] for b in box)), "moc_p2p_connect")
                mine_ok = 1
            except Exception as e:  # noqa: BLE001
                mine_ok, self.why = 0, f"connect: {e}"
            # self-check on a known vector (also proves every rank reached this point with a live mapping)
            flag = torch.tensor([mine_ok], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if int(flag.item()) == 1:
                a = (torch.arange(1, self.n_par + 1, dtype=torch.float32, device=device) % 1021) * (self.rank + 1)
                b = a.clone()
                self.all_reduce_(a)
                dist.all_reduce(b, group=group)
                torch.cuda.synchronize()
                good = int(torch.equal(a, b)) and self.error() == 0
                flag.fill_(good)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                self.ok = int(flag.item()) == 1
                if not self.ok:
                    self.why = "self-check against torch.distributed failed on some rank"
            elif not self.why:
                self.why = "a peer could not map the buffers"
        if not self.ok:
            import warnings
            warnings.warn(f"peer-to-peer gradient exchange unavailable ({self.why}); using the collective")

    def all_reduce_(self, flat: torch.Tensor):
        self.check(self.lib.moc_p2p_allreduce(self.handle, flat.data_ptr(), flat.numel(),
                                              torch.cuda.current_stream().cuda_stream), "moc_p2p_allreduce")

    def error(self) -> int:
        return int(self.lib.moc_p2p_error(self.handle))

    def close(self):
        if self.handle:
            self.lib.moc_p2p_destroy(self.handle)
            self.handle, self.ok = self.C.c_void_p(), False


_direct = {}
_p2p = {}


def p2p_exchange(device, n_par: int, group=None):
    """The communicator for (group, device, n_par), created on first use (collective!).  None when
    MOC_DP_EXCHANGE=rccl, outside a process group, at world 1 or when the set-up self-check failed."""
    import os
    if os.environ.get("MOC_DP_EXCHANGE", "auto") == "rccl" or not dist.is_initialized():
        return None
    if dist.get_world_size(group) == 1 or dist.get_world_size(group) > 8:
        return None
    key = (id(group), torch.device(device).index, int(n_par))
    x = _p2p.get(key)
    if x is None:
        x = _p2p[key] = P2pExchange(device, n_par, group)
    return x if x.ok else None


def meta_grad_allreduce(flat: torch.Tensor, group=None):
    """Sum-all-reduce of the flat meta-gradient on the current stream (direct RCCL when it came up)."""
    if not dist.is_initialized():
        return
    if flat.is_cuda and dist.get_backend(group) == "nccl":
        key = (id(group), flat.device.index)
        d = _direct.get(key)
        if d is None:
            d = _direct[key] = DirectRccl(flat.device, group)
        if d.ok:
            d.all_reduce_(flat)
            return
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)


def allreduce_entry(flat: torch.Tensor, group=None):
    """-> (function address, comm, keepalive) for moc_train_steps_dp: ncclAllReduce itself and this
    module's communicator when direct RCCL came up; otherwise a ctypes callback with the same
    signature that runs torch.distributed's all-reduce of `flat` (gloo, or nccl without the direct
    path).  (None, None, None) outside a process group."""
    import ctypes as C
    if not dist.is_initialized():
        return None, None, None
    if flat.is_cuda and dist.get_backend(group) == "nccl":
        key = (id(group), flat.device.index)
        d = _direct.get(key)
        if d is None:
            d = _direct[key] = DirectRccl(flat.device, group)
        if d.ok:
            return C.cast(d.lib.ncclAllReduce, C.c_void_p), d.comm, d
    proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p)

    def _cb(send, recv, count, dtype, op, comm, stream):
        try:
            assert send == recv == flat.data_ptr() and count == flat.numel()
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            return 0
        except Exception:  # noqa: BLE001 -- reported to the C caller as a failed collective
            import traceback
            traceback.print_exc()
            return 1

    fn = proto(_cb)
    return C.cast(fn, C.c_void_p), None, fn


def exchange_error() -> int:
    """0, or 1 + the rank whose push did not arrive within the time-out in some exchange so far
    (meaningful after a synchronize; the pass it happened in must be discarded)."""
    for x in _p2p.values():
        if x.handle and x.error():
            return x.error()
    return 0


def shutdown():
    for d in _direct.values():
        d.close()
    _direct.clear()
    if _p2p:
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()          # nobody unmaps while a peer may still push
    for x in _p2p.values():
        x.close()
    _p2p.clear()


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
    resident = isinstance(loader, M.ResidentBags)
    if not resident:
        batch.phase_a(bank)                 # (a resident split's phase A is already issued, possibly a pass ahead)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    # the whole pass is ONE call, the host loop lives in C.  On one node the exchange happens inside
    # the step kernel (moc_train_steps_p2p: forward + step, two launches); otherwise per step
    # forward, pool+loss+gradients, the ONE collective (the flat gradient), Adam with 1/world.
    x = p2p_exchange(device, fg.flat.numel(), group) if (fg.flat.is_cuda and engine.fused_step_shape(batch)) else None
    if x is not None:
        if x.error():
            raise RuntimeError(f"peer-to-peer exchange: rank {x.error() - 1} stayed silent past the time-out in an earlier pass")
        engine.train_steps_p2p(batch, meta, lab, 0, len(sizes), use, x.handle)
        train_dp.exchange = "p2p"
    else:
        fn, comm, keep = allreduce_entry(fg.flat, group)
        engine.train_steps_dp(batch, meta, lab, 0, len(sizes), use, fg.flat, fn, comm, world)
        del keep
        train_dp.exchange = "collective"
    meta.advance(len(sizes))            # the optimizer's own step counters, once per pass
    if resident:
        M.resident_pass_done(loader, device, args)
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
