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
        import warnings
        self.C = C
        self.ok = False
        self.comm = C.c_void_p()
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.why = ""

        def agree(mine_ok: bool) -> bool:
            """Collective: True only when EVERY rank got through the stage just finished.  A rank that failed still
            joins every agreement, so nobody is left waiting in a collective its peer skipped."""
            flag = torch.tensor([1 if mine_ok else 0], dtype=torch.int32, device=device if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            return int(flag.item()) == 1

        class UniqueId(C.Structure):                          # ncclUniqueId: 128 opaque bytes, passed BY VALUE
            _fields_ = [("internal", C.c_byte * 128)]

        # stage 1 (not collective): the library and its symbols
        good = True
        try:
            path = os.environ.get("MOC_RCCL_LIB") or os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            self.lib = C.CDLL(path)
            self.lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
            self.lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
            self.lib.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
        except Exception as e:  # noqa: BLE001  (set-up only; the data path never swallows errors)
            good, self.why = False, f"rank {self.rank}: librccl: {e}"
        if not agree(good):
            warnings.warn(f"direct RCCL unavailable ({self.why or 'on another rank'}); using torch.distributed.all_reduce")
            return
        # stage 2: the unique id travels through torch.distributed (every rank takes part, whatever rank 0 got)
        uid = UniqueId()
        if self.rank == 0:
            try:
                good = self.lib.ncclGetUniqueId(C.byref(uid)) == 0
            except Exception as e:  # noqa: BLE001
                good, self.why = False, f"ncclGetUniqueId: {e}"
        box = [bytes(uid)]
        dist.broadcast_object_list(box, src=0, group=group)
        uid = UniqueId.from_buffer_copy(box[0])
        if not agree(good):
            warnings.warn(f"direct RCCL unavailable ({self.why or 'rank 0 got no unique id'}); using torch.distributed.all_reduce")
            return
        # stage 3 (collective inside RCCL: every rank is here, all agreed so far)
        try:
            torch.cuda.set_device(device)
            rc = self.lib.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank)
            good = rc == 0
            if not good:
                self.why = f"rank {self.rank}: ncclCommInitRank rc={rc}"
        except Exception as e:  # noqa: BLE001
            good, self.why = False, f"rank {self.rank}: ncclCommInitRank: {e}"
        if not agree(good):
            warnings.warn(f"direct RCCL unavailable ({self.why or 'communicator failed on another rank'}); using torch.distributed.all_reduce")
            self._destroy_comm()
            return
        # stage 4: self-check against torch's collective on a known vector (both collectives on every rank)
        try:
            a = torch.arange(1, 1025, dtype=torch.float32, device=device) * (self.rank + 1)
            b = a.clone()
            self.all_reduce_(a)
            dist.all_reduce(b, group=group)
            torch.cuda.synchronize()
            good = bool(torch.equal(a, b))
            if not good:
                self.why = f"rank {self.rank}: direct RCCL all-reduce disagrees with torch.distributed"
        except Exception as e:  # noqa: BLE001
            good, self.why = False, f"rank {self.rank}: self-check: {e}"
        self.ok = agree(good)
        if not self.ok:
            warnings.warn(f"direct RCCL unavailable ({self.why or 'self-check failed on another rank'}); using torch.distributed.all_reduce")
            self._destroy_comm()

    def _destroy_comm(self):
        try:
            if self.comm:
                self.lib.ncclCommDestroy.argtypes = [self.C.c_void_p]
                self.lib.ncclCommDestroy(self.comm)
        except Exception:  # noqa: BLE001
            pass
        self.comm = self.C.c_void_p()

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
        self.step_checked = False
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

    def step_self_check(self, device, bank, topj: int, topk: int, group=None) -> bool:
        """Collective.  The set-up check above runs p2p_allreduce_kernel; what training runs is the STEP kernel's own
        push / flag / sum path (pool_w1_step_kernel or its wide form, chosen from the run's constants).  So before
        train_dp trusts the exchange: two synchronous steps of the real step kernel -- both buffer parities -- on two
        small synthetic slides per rank, the last rank deliberately late at the second step, compared with the same two
        steps taken the slow way: every rank's gradient (moc_train_grad) all-gathered through torch.distributed, summed
        in rank order (the order p2p_sum uses) and applied by moc_adam_step.  All ranks must hold the same bits and
        match that expectation to 1e-6; any doubt on any rank sends ALL ranks to the collective path."""
        import time
        from . import engine, main_moc as M
        world, rank = self.world, self.rank
        D, C_, Ce = bank.D, bank.C, bank.Ce
        verdict = 0
        try:
            g = torch.Generator(device="cpu")
            g.manual_seed(97)                                   # same meta-learner on every rank, torch's own RNG untouched

            def fresh():
                m = M.senet(D, 4)
                with torch.no_grad():
                    for p_ in m.parameters():
                        p_.copy_(torch.randn(p_.shape, generator=g) * 0.05)
                m = m.to(device)
                return m, torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-4)
            model_a, opt_a = fresh()
            g.manual_seed(97)
            model_b, opt_b = fresh()
            gx = torch.Generator(device="cpu")
            gx.manual_seed(1000 + rank)                         # different slides on every rank
            sizes = [700 + 37 * rank, 650 + 11 * rank]
            X = torch.randn(sum(sizes), D, generator=gx)
            X = (X / X.norm(dim=1, keepdim=True)).to(bank.dtype).to(device)
            batch = engine.SlideBatch(X, sizes, C_, Ce, topj, topk)
            batch.phase_a(bank)
            lab = torch.tensor([rank % C_, (rank + 1) % C_], dtype=torch.int64, device=device)
            use = 15
            # (a) the step kernel's exchange, one step per call so that a rank can be late between them
            meta_a = engine.MetaState(model_a, opt_a)
            engine.train_steps_p2p(batch, meta_a, lab, 0, 1, use, self.handle)
            meta_a.advance(1)
            torch.cuda.synchronize()
            if rank == world - 1:
                time.sleep(0.25)                                # the others' step kernels wait in their bounded polls
            engine.train_steps_p2p(batch, meta_a, lab, 1, 1, use, self.handle)
            meta_a.advance(1)
            # (b) the same two steps through torch.distributed
            meta_b = engine.MetaState(model_b, opt_b, need_grads=True)
            fg = FlatGrads(meta_b.params, device)
            for name, v in zip(("g_W1", "g_b1", "g_W2", "g_b2"), fg.views):
                setattr(meta_b.c, name, v.data_ptr())
            for t in range(2):
                engine.train_grad(batch, meta_b, lab, t, use)
                parts = [torch.empty_like(fg.flat) for _ in range(world)]
                dist.all_gather(parts, fg.flat, group=group)
                total = torch.zeros_like(fg.flat)
                for q in range(world):                          # rank order, as p2p_sum
                    total += parts[q]
                fg.flat.copy_(total)
                engine.adam_step(meta_b, grad_scale=1.0 / world)
            torch.cuda.synchronize()
            pa = torch.cat([p_.detach().reshape(-1) for p_ in model_a.parameters()])
            pb = torch.cat([p_.detach().reshape(-1) for p_ in model_b.parameters()])
            close = bool(torch.isfinite(pa).all()) and float((pa - pb).abs().max()) <= 1e-6
            # the same bits on every rank: compare a checksum of the raw words
            h = pa.view(torch.int32).to(torch.int64)
            sig = torch.stack([h.sum(), (h * torch.arange(1, h.numel() + 1, device=device)).sum()])
            lo, hi = sig.clone(), sig.clone()
            if dist.get_backend(group) != "nccl":
                lo, hi = lo.cpu(), hi.cpu()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
            verdict = int(close and torch.equal(lo, hi) and self.error() == 0)
            if not verdict:
                self.why = (f"step-kernel self-check: max |p2p - gathered| = {float((pa - pb).abs().max()):.3e}, "
                            f"ranks agree = {bool(torch.equal(lo, hi))}, error word = {self.error()}")
        except Exception as e:  # noqa: BLE001 (set-up only)
            self.why = f"step-kernel self-check raised: {e}"
            verdict = 0
        flag = torch.tensor([verdict], dtype=torch.int32, device=device)
        if dist.get_backend(group) != "nccl":
            flag = flag.cpu()
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        self.step_checked = True
        if int(flag.item()) != 1:
            self.ok = False
            import warnings
            warnings.warn(f"peer-to-peer gradient exchange failed its step-kernel self-check ({self.why or 'on another rank'}); "
                          "using the collective")
        return self.ok

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


def drop_p2p():
    """Collective: close every peer-to-peer exchange and forget it (and its sticky error word), e.g. before a
    re-run on the collective path after a time-out.  Nobody unmaps while a peer may still push: drain, barrier, close."""
    if not _p2p:
        return
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    for x in _p2p.values():
        x.close()
    _p2p.clear()


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
        M._loader_seed_draw(loader)
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
    if x is not None and not x.step_checked:
        # first use with this shape: the step kernel's own exchange path must reproduce a gathered sum (collective)
        if not x.step_self_check(device, bank, args.topj, args.topk, group):
            x = None
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


# --------------------------------------------------------------------------- splits spread over the ranks
def block_lists(n_items: int, world: int):
    """Contiguous blocks of the loader order, one per rank, BALANCED: n // world items each and one more on the first
    n % world ranks, so no rank is empty unless world > n (ceil-sized blocks left trailing ranks empty: 12 slides on
    8 ranks gave 2, 2, 2, 2, 2, 2, 0, 0)."""
    base, extra = divmod(n_items, world)
    out, lo = [], 0
    for r in range(world):
        k = base + (1 if r < extra else 0)
        out.append(list(range(lo, lo + k)))
        lo += k
    return out


def seq_layout(counts_by_rank):
    """Where the gathered pieces of a pass land.  counts_by_rank[r] = n_sel of rank r's slides of this pass, in loader
    order.  Every rank sends `mx` rows (its own sum S, padded to the largest sum of any rank); rank r's piece starts at
    row r * mx of the gathered array, its slides back to back inside it.  -> (mx, row_off over ALL slides in loader
    order + the end, n_sel in loader order).  Pure: the same on every rank, testable without a GPU."""
    totals = [int(sum(c)) for c in counts_by_rank]
    mx = max(1, max(totals) if totals else 1)
    row_off, n_sel = [], []
    for r, cs in enumerate(counts_by_rank):
        o = r * mx
        for c in cs:
            assert int(c) >= 1, "a slide of the pass selected no row"
            row_off.append(o)
            n_sel.append(int(c))
            o += int(c)
    row_off.append(row_off[-1] + n_sel[-1] if n_sel else 0)
    return mx, row_off, n_sel


class ShardedSplit:
    """One split of a run spread over the ranks of a process group: every rank knows every slide's label (and which
    rank holds it); the bags of `my_ids` live here, packed in HBM (`local`, a main_moc.ResidentBags, or None when this
    rank holds none).  Offers what main_moc's loops ask of `loader.dataset` (real_len(), repeat_num, len()).  The
    evaluation loops below take it where main_moc's take a loader."""

    def __init__(self, my_bags, my_ids, all_labels, index_lists, device, dtype=None, paths=None):
        from . import main_moc as M
        self.all_labels = [int(v) for v in all_labels]
        self.my_ids = [int(i) for i in my_ids]
        self.index_lists = [[int(i) for i in lst] for lst in index_lists]
        assert sorted(i for lst in self.index_lists for i in lst) == list(range(len(self.all_labels))), "shards do not partition the slides"
        assert len(my_bags) == len(self.my_ids)
        self.device = torch.device(device)
        self.local = (M.ResidentBags(my_bags, [self.all_labels[i] for i in self.my_ids], device, dtype=dtype, paths=paths)
                      if len(my_bags) else None)
        self.repeat_num = None
        self.dataset = self

    def real_len(self):
        return len(self.all_labels)

    def __len__(self):
        return self.repeat_num if self.repeat_num else len(self.all_labels)


def _eval_sharded(split: ShardedSplit, device, args, mode, model=None, pooling_func=None, group=None):
    """Every rank runs main_moc's batched evaluation pass over ITS slides; one all_gather of the [n_local, C + 1]
    (pooled logits | loss) rows; the reference's metrics over all slides, the same dict on every rank."""
    from . import main_moc as M
    real_len = split.real_len()
    n_div = len(split)                      # main_moc.py:499-501: the loss is divided by len(dataset) as the caller left it
    C_ = args.n_classes
    with torch.no_grad():
        if split.local is not None:
            split.local.repeat_num = None
            if mode == "zs_custom":
                pooled, _, losses = M._eval_pass_custom(split.local, device, args, pooling_func)
            else:
                pooled, _, losses = M._eval_pass(split.local, device, args, mode, model=model)
            both = torch.cat([pooled.to(device), torch.tensor(losses, dtype=torch.float32, device=device).unsqueeze(1)], 1)
        else:
            both = torch.zeros((0, C_ + 1), dtype=torch.float32, device=device)
    counts = [len(lst) for lst in split.index_lists]
    allv = unshard(gather_rows(both, counts, group), split.index_lists, real_len).cpu()
    return M._metrics(allv[:, :-1].contiguous(), list(split.all_labels), allv[:, -1].tolist(), n_div, real_len, args)


def evaluation(model, split: ShardedSplit, device, args, group=None):
    """main_moc.evaluation (main_moc.py:462-520) over a split spread across the ranks."""
    if model.training:
        model.eval()
    return _eval_sharded(split, device, args, "eval", model=model, group=group)


def zs_evaluation(split: ShardedSplit, device, args, pooling_func=None, group=None):
    """main_moc.zs_evaluation (main_moc.py:412-460) over a split spread across the ranks."""
    from . import patch_selection_classifier as P
    kinds = {None: "topj", P.topj_pooling: "topj", P.delta_softmax_classifier_pooling: "delta_softmax",
             P.delta_diff_classifier_pooling: "delta_diff", P.bottomk_irrel_classifier_pooling: "bottomk"}
    if pooling_func in kinds:
        return _eval_sharded(split, device, args, "zs_" + kinds[pooling_func], group=group)
    return _eval_sharded(split, device, args, "zs_custom", pooling_func=pooling_func, group=group)


def ablation_evaluation(split: ShardedSplit, device, args, group=None):
    """main_moc.ablation_evaluation (main_moc.py:523-582) over a split spread across the ranks."""
    return _eval_sharded(split, device, args, "ablation", group=group)


# --------------------------------------------------------------------------- exact-sequential multi-GPU (section 8e mode 1)
class SeqShardedBags(ShardedSplit):
    """A train split whose slides live on G GPUs -- rank r holds the r-th contiguous block of the loader order
    (block_lists: balanced) -- for train_seq.  Every rank knows every slide's size and label (the mask stream and the
    labels are global); only the bag rows are sharded.  Being a ShardedSplit it can also be evaluated (main_moc.py:615
    evaluates the train split every epoch).

    More ranks than slides cannot work (a rank with no bag has no phase A to contribute and no bank dtype to go by):
    that is decided from (n, world) alone, so EVERY rank raises the same ValueError here, before any collective."""

    def __init__(self, my_bags, all_sizes, all_labels, device, rank: int, world: int, dtype=None, paths=None):
        self.all_sizes = [int(v) for v in all_sizes]
        n = len(self.all_sizes)
        assert len(all_labels) == n and n >= 1
        self.rank, self.world = int(rank), int(world)
        if world > n:
            raise ValueError(f"exact-sequential training: the train split has {n} slide(s) but the job has {world} ranks -- "
                             f"every rank must hold at least one slide; run with at most {n} rank(s)")
        blocks = block_lists(n, world)
        self.blocks = blocks
        self.per = max(len(b_) for b_ in blocks)                 # slides of the fullest rank
        self.lo, self.hi = blocks[rank][0], blocks[rank][-1] + 1
        assert len(my_bags) == self.hi - self.lo, f"rank {rank} holds slides [{self.lo}, {self.hi}) of the loader order"
        assert [int(b.size(0)) for b in my_bags] == self.all_sizes[self.lo:self.hi]
        super().__init__(my_bags, blocks[rank], all_labels, blocks, device, dtype=dtype, paths=paths)
        self.next_pass_len = None
        self._plans = {}


def _seq_plan(sh: SeqShardedBags, m: int, bank, args):
    import ctypes
    from . import engine
    from .engine import CompactBatch, SlideBatch
    key = (m, bank.C, bank.Ce, args.topj, args.topk, tuple(sorted(args.discard_classifiers or ())))
    plan = sh._plans.get(key)
    if plan is not None:
        return plan
    assert m <= sh.real_len()                                   # (longer passes are split into rounds by train_seq)
    dev, per, world = sh.device, sh.per, sh.world
    # the slides of this pass (the first m of the loader order) that each rank holds
    n_by_rank = [max(0, min(m, blk[-1] + 1) - blk[0]) for blk in sh.blocks]
    n_loc = n_by_rank[sh.rank]
    cap = min(args.topj * (2 * bank.C + 2), max(sh.all_sizes))  # a slide selects at most this many rows
    nk = 2 * bank.C + 2
    starts = [0]
    for v in sh.all_sizes:
        starts.append(starts[-1] + v)
    X = sh.local.X if sh.local is not None else None
    dtype, D = bank.dtype, bank.D
    pin = (lambda t: t.pin_memory()) if torch.device(dev).type == "cuda" else (lambda t: t)
    sets = []
    for _ in range(2):
        local = None
        if n_loc:
            sizes = sh.local.sizes[:n_loc]
            local = SlideBatch(X, sizes, bank.C, bank.Ce, args.topj, args.topk, args.discard_classifiers,
                               mask=torch.ones(sum(sizes), dtype=torch.uint8), x_starts=sh.local.starts[:n_loc])
        # what this rank sends: its slides' selected rows back to back (moc_pack_selected_rows), at most per * cap of them
        send_feat = torch.zeros((per * cap, D), dtype=dtype, device=dev)
        # what every rank ends up with: world pieces of `mx` rows each (mx = the pass's largest per-rank sum, known
        # once the counts have been exchanged); at world 1 the piece IS the send buffer -- nothing is copied
        recv_feat = send_feat if world == 1 else torch.zeros((world * per * cap, D), dtype=dtype, device=dev)
        sets.append({
            "local": local,
            "send_feat": send_feat,
            "send_cand": torch.zeros((per * cap, nk), dtype=torch.float32, device=dev),
            "send_nsel": torch.zeros(per, dtype=torch.int32, device=dev),
            "all_nsel": torch.zeros(world * per, dtype=torch.int32, device=dev),
            "nsel_host": pin(torch.zeros(world * per, dtype=torch.int32)),
            "recv_cand": torch.zeros((world * per * cap, nk), dtype=torch.float32, device=dev),
            "compact": CompactBatch(m, world * per * cap, cap, D, dtype, bank.C, bank.Ce, args.topj, args.topk, dev, X=recv_feat),
            "stage": torch.empty(max(1, starts[min(m, sh.hi)] - starts[sh.lo]) if n_loc else 1, dtype=torch.uint8).pin_memory(),
            "stage_free": None, "steps_done": None,
        })
    side = torch.cuda.Stream(device=dev)
    for st in sets:
        cb = st["compact"]
        ts = [st["send_feat"], st["send_cand"], st["send_nsel"], st["all_nsel"], st["recv_cand"], cb.X, cb.cand, cb.n_sel, cb.row_off]
        if st["local"] is not None:
            b = st["local"]
            if bank.Ce <= 16:
                b.reserve_cus()                  # its phase A runs beside the meta-steps of the pass before (main_moc.train_plan)
            ts += [t for t in (b.kept, b.n_kept, b.stats, b.sel_flag, b.sel_idx, b.sel_row, b.n_sel, b.cand, b.row_off, b.x_off, b.ticket) if t is not None]
            X.record_stream(side)
        for t in ts:
            t.record_stream(side)
    # position of (rank r, local slide i) in the gathered count vector, for the pass's slides in loader order
    order = [r * per + i for r in range(world) for i in range(n_by_rank[r])]
    plan = sh._plans[key] = {
        "sets": sets, "turn": 0, "ahead": None, "side": side, "m": m, "n_loc": n_loc, "n_by_rank": n_by_rank, "cap": cap,
        "row_lo": starts[sh.lo], "row_hi": starts[min(m, sh.hi)] if n_loc else starts[sh.lo], "rows_total": starts[m],
        "labels": torch.tensor(sh.all_labels[:m], dtype=torch.int64).to(dev),
        "order": torch.tensor(order, dtype=torch.int64).to(dev),
        "all_masks": torch.empty(starts[m], dtype=torch.uint8),
        "drawer": engine.MaskDrawer(starts[m], (ctypes.c_int64 * (m + 1))(*starts[:m + 1]), m),
        "sent_rows": 0, "padded_rows": 0,                       # statistics: rows this rank sent / would have sent in cap-sized blocks
    }
    return plan


def _all_gather_into(out: torch.Tensor, inp: torch.Tensor, group=None):
    """out = concatenation over the ranks of `inp` (equal shapes), on the current stream: ONE RCCL all-gather
    ("nccl" backend); with gloo (the one-device tests) the list form over views of `out`."""
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out, inp, group=group)
        return
    world = dist.get_world_size(group)
    parts = out.view(world, *inp.shape)
    dist.all_gather([parts[q] for q in range(world)], inp, group=group)


def seq_exchange(st, n_by_rank, per, world, group=None, sync=None):
    """The hand-over of a pass (collective): counts first, then the UNPADDED pieces.  `st` holds this rank's packed
    selections (`send_feat` [.., D], `send_cand` [.., 2C+2] row-major, `send_nsel` [per]) and the receive side
    (`all_nsel` [world * per], `nsel_host`, `recv_cand`, `compact`).  Afterwards `compact` is the pass's phase-A
    result for ALL slides in loader order: X = world pieces of `mx` rows, cand transposed to [2C+2, rows], n_sel and
    row_off per slide.  Device-agnostic (the CPU tests run it over gloo); `sync(tensor)` makes a device->host copy
    that is safe to read (default: the copy is synchronous).  -> (mx, rows this rank sent)."""
    cb = st["compact"]
    if world > 1:
        _all_gather_into(st["all_nsel"], st["send_nsel"], group)
    else:
        st["all_nsel"].copy_(st["send_nsel"])
    if sync is None:
        st["nsel_host"].copy_(st["all_nsel"])
    else:
        sync(st["nsel_host"], st["all_nsel"])
    counts = st["nsel_host"].tolist()
    mx, row_off, n_sel = seq_layout([counts[r * per:r * per + n_by_rank[r]] for r in range(world)])
    assert world * mx <= cb.total
    if world > 1:
        _all_gather_into(cb.X[:world * mx], st["send_feat"][:mx], group)
        _all_gather_into(st["recv_cand"][:world * mx], st["send_cand"][:mx], group)
        cand_rows = st["recv_cand"][:world * mx]
    else:
        cand_rows = st["send_cand"][:mx]                   # (cb.X is the send buffer itself)
    cb.cand[:, :world * mx].copy_(cand_rows.t())
    return mx, row_off, n_sel


def _seq_issue(sh, plan, turn, bank, rng_before, group, host_wait=None):
    """Masks of the whole pass from generator state `rng_before` (every rank draws the same stream and keeps its own
    rows), phase A of this rank's slides, their selections packed back to back, then the hand-over (seq_exchange: one
    small all-gather of the counts, two of the unpadded pieces): afterwards set `turn`'s CompactBatch holds the pass's
    phase-A result for ALL slides, in loader order.  On the CURRENT stream; the host waits once, for the counts.
    -> generator state after the draws (None: torch drew for us)."""
    from . import engine
    st = plan["sets"][turn]
    n_loc, cap = plan["n_loc"], plan["cap"]
    drawn = plan["drawer"].take(rng_before)             # the whole pass's flags, usually drawn a pass ahead (helper thread)
    if drawn is not None:
        allm, _, _, rng_after, buf = drawn
        mine = allm[plan["row_lo"]:plan["row_hi"]]      # this rank's rows: a view of the pinned buffer, read in place
    else:                                               # generator layout unknown to the replay: torch draws, in line
        if st["stage_free"] is not None:
            st["stage_free"].synchronize()
        _, _, rng_after = engine.draw_row_masks_from(rng_before, plan["rows_total"], plan["all_masks"])
        buf = None
        if n_loc:
            st["stage"].copy_(plan["all_masks"][plan["row_lo"]:plan["row_hi"]])
        mine = st["stage"]
    if host_wait is not None:
        host_wait.synchronize()
    if n_loc:
        b = st["local"]
        b.use_host_mask(mine, int(mine.sum()))
        b.phase_a(bank)
        engine.pack_selected_rows(b, 0, n_loc, cap, st["send_feat"], st["send_cand"])
        st["send_nsel"][:n_loc].copy_(b.n_sel)

    def sync(host, devt):
        host.copy_(devt, non_blocking=True)
        e = torch.cuda.Event()
        e.record()
        e.synchronize()
    mx, row_off, _ = seq_exchange(st, plan["n_by_rank"], sh.per, sh.world, group, sync)
    cb = st["compact"]
    cb.n_sel.copy_(st["all_nsel"][plan["order"]])       # loader order
    cb.set_layout(row_off)
    plan["sent_rows"] += mx
    plan["padded_rows"] += sh.per * cap
    ev = torch.cuda.Event()
    ev.record()
    if buf is not None:
        plan["drawer"].attach(buf, ev)
    else:
        st["stage_free"] = ev
    return rng_after


def train_seq(model, shard: SeqShardedBags, optimizer, device, args, group=None):
    """Exact-sequential multi-GPU training pass (SURVEY.md section 8e mode 1): the reference's recurrence -- one Adam
    step per slide, in loader order (main_moc.py:380-410) -- with phase A (everything in slide_process, which has no
    trainable parameter) sharded over the GPUs that hold the bags.  Each rank runs phase A on its block of slides,
    the compact results (selected rows, candidate scores, counts: moc_pack_selected) are all-gathered, and EVERY rank
    then runs the same meta-steps over all slides, so all ranks hold bit-identical parameters -- the ones a single
    GPU computes -- with no parameter broadcast.  Phase A + gather of the next pass are issued a pass ahead on a side
    stream, as in main_moc.train.  Unlike train_dp this does not change the optimisation trajectory."""
    from . import engine, main_moc as M
    if not model.training:
        model.train()
    m_all, n_real = len(shard), shard.real_len()
    if m_all > n_real:
        # repeat_num beyond the split (datasets/dataset_generic.py:380-393: visit v is slide v mod n): whole rounds over
        # the slides, then the remainder -- each round a pass of its own, in order, the mask stream running on
        rounds = [n_real] * (m_all // n_real) + ([m_all % n_real] if m_all % n_real else [])
        keep_rep, keep_next = shard.repeat_num, shard.next_pass_len
        try:
            for i, L in enumerate(rounds):
                shard.repeat_num = None if L == n_real else L
                if i + 1 < len(rounds):
                    shard.next_pass_len = rounds[i + 1]
                elif keep_next == 0:
                    shard.next_pass_len = 0
                else:                                   # the next call's first round
                    shard.next_pass_len = min(n_real, keep_next if keep_next is not None else m_all)
                train_seq(model, shard, optimizer, device, args, group)
        finally:
            shard.repeat_num, shard.next_pass_len = keep_rep, keep_next
        return
    use = engine.train_use_bits(args.discard_classifiers)
    assert shard.local is not None                     # (SeqShardedBags refuses world > slides, on every rank alike)
    bank = M._bank_for(shard.local.X, device)
    assert bank.C == args.n_classes
    m = len(shard)
    plan = _seq_plan(shard, m, bank, args)
    now = torch.get_rng_state()
    ahead, plan["ahead"] = plan["ahead"], None
    if ahead is not None and ahead["bank"] is bank and ahead["after"] is not None and torch.equal(ahead["before"], now):
        torch.set_rng_state(ahead["after"])
        torch.cuda.current_stream().wait_event(ahead["done"])
        plan["turn"] = ahead["turn"]
    else:
        if ahead is not None:
            ahead["done"].synchronize()
        plan["turn"] = 1 - plan["turn"]
        after = _seq_issue(shard, plan, plan["turn"], bank, now, group)
        if after is not None:
            torch.set_rng_state(after)
    cb = plan["sets"][plan["turn"]]["compact"]
    meta = engine.MetaState.cached(model, optimizer)
    engine.train_steps(cb, meta, plan["labels"], 0, m, use)
    train_seq.last = (cb, plan["labels"])
    train_seq.last_local = plan["sets"][plan["turn"]]["local"]
    # ---- the next pass's phase A + gather, a pass ahead on the side stream (main_moc.resident_pass_done)
    mark = torch.cuda.Event()
    mark.record(torch.cuda.current_stream())
    plan["sets"][plan["turn"]]["steps_done"] = mark
    hint = shard.next_pass_len
    if not M.PREFETCH_PHASE_A or hint == 0:
        return
    if hint is not None:
        hint = min(hint, shard.real_len())              # (a longer pass starts with a whole round)
    nplan = plan
    if hint is not None and hint != m:
        nplan = _seq_plan(shard, hint, bank, args)
        if nplan["ahead"] is not None:
            nplan["ahead"]["done"].synchronize()
            nplan["ahead"] = None
    other = 1 - nplan["turn"]
    before = torch.get_rng_state()
    with torch.cuda.stream(nplan["side"]):
        after = _seq_issue(shard, nplan, other, bank, before, group, host_wait=nplan["sets"][other]["steps_done"])
        done = torch.cuda.Event()
        done.record(nplan["side"])
    if after is None:
        torch.set_rng_state(before)
    nplan["ahead"] = {"turn": other, "before": before, "after": after, "done": done, "bank": bank}


def train_minibatch(model, loader, optimizer, device, args, G: int):
    """ONE process, one GPU: the optimisation trajectory of train_dp at world size G -- synchronous minibatches of G
    consecutive slides (step t = slides [tG, (t+1)G) of the loader order, which is how a G-rank run shards them),
    mean gradient, one Adam step per minibatch -- by gradient accumulation.  It exists to answer, without G GPUs,
    whether minibatch data parallelism keeps the reference's AUC (SURVEY.md section 8e mode 2: "must be validated");
    masks come from this process's generator in loader order (a G-rank run draws them per rank)."""
    from . import engine, main_moc as M
    assert G >= 1
    model.train()
    use = engine.train_use_bits(args.discard_classifiers)
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
    for name, v in zip(("g_W1", "g_b1", "g_W2", "g_b2"), fg.views):
        setattr(meta.c, name, v.data_ptr())
    acc = torch.zeros_like(fg.flat)
    batch.phase_a(bank)
    for t0 in range(0, len(sizes), G):
        group = range(t0, min(t0 + G, len(sizes)))
        acc.zero_()
        for b in group:                                   # rank order: the same summation order as the exchange
            engine.train_grad(batch, meta, lab, b, use)
            acc += fg.flat
        fg.flat.copy_(acc)
        engine.adam_step(meta, grad_scale=1.0 / len(group))
    train_minibatch.last = (batch, lab, fg)


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
