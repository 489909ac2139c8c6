"""Host orchestration over libmoc_hip: packed slide batches, the prepared
classifier bank, the meta-learner state shared with torch.optim.Adam, and the
batched phase-A / phase-B drivers.  PyTorch here is plumbing only: it owns the
device buffers and the stream; every kernel is ours.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Sequence

import torch

from . import _lib
from ._lib import MocBatch, MocMeta, MocMetaWs, check, lib, ptr

HIDDEN = 64
# MOC_STEP_GRAPH=1: a pass of meta-steps as one hipGraph launch (moc_train_steps_graph) instead of 2 n + 1 stream
# launches (moc_train_steps).  Bit-identical (tests/test_gpu_graph.py) and it frees the host (38 against 170-250 us per
# 32-step pass), but on the GPU the replayed chain is 1-3 % SLOWER than the stream launches (fp32 bags: 27.7 against
# 26.9 us per step; steady state 36.6 k against 37.0 k meta-steps/s, bf16 42.5 k against 43.4 k) and its first kernel
# starts later (the whole graph is enqueued before the doorbell), so stream launches stay the default.
STEP_GRAPH = os.environ.get("MOC_STEP_GRAPH", "0") == "1"
GRAPH_TABLE_STEPS = 4096        # Adam steps of coefficients kept on the device per table build
CAND_FROM_STATS = os.environ.get("MOC_CAND_FROM_STATS", "1") != "0"   # evaluation: no materialised candidate columns (0: as in training)
COMPACT_STATS = os.environ.get("MOC_COMPACT_STATS", "1") != "0"     # wide banks: C + 5 statistics per row (0: always 2C + 3)

# The look-ahead phase A of a train pass stays off MOC_RESERVE_CUS compute units, which the sequential meta-steps of the
# pass in progress then find free (moc_batch_t.cu_reserved + tile_ticket, include/moc_hip.h; 0: the whole chip, static
# walk).  Measured on the default workload (scripts/sweep_reserve_cus.sh, profiles/NOTES.md): 0 -> 42.9 k, 48 -> 44.2 k,
# 64 -> 45.6 k, 96 -> 42.9 k meta-steps/s; the ticketed walk alone costs the score pass 7 % (fp32) / 19 % (bf16), so
# passes with nothing beside them (evaluation) keep the static walk.
RESERVE_CUS = int(os.environ.get("MOC_RESERVE_CUS", "64"))
# the training step pools among the forward's tile records (moc_meta_ws_t.tile_ws; 0: re-reads every mixed score, round 3)
TILE_RECORDS = os.environ.get("MOC_TILE_RECORDS", "1") != "0"
N_SEL_HOST = os.environ.get("MOC_N_SEL_HOST", "1") != "0"            # moc_batch_t.n_sel_host for the train steps (0: never, diagnostic)

# bench.py sets this to a list: every batched score-pass launch then appends
# (start_event, stop_event, algorithmic_bytes) recorded on the launch stream.
SCORE_EVENTS = None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The current stream's hipStream_t.  torch.cuda.current_stream() builds a Stream object through four layers of device-index
    look-ups (4-9 us a call, several calls per pass, two of them in front of a pass's first launch); the raw accessor is the
    same handle in 0.3 us."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_census = {}      # device index -> sorted list of (xcc, hw_id_bits) slots
_reserved = {}    # (device index, n) -> device int32[128] bitmap


def cu_slots(device):
    """The compute units of `device` as (xcc, HW_ID[15:8]) slots, found by moc_cu_census (once per process)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _census:
        n_cu = torch.cuda.get_device_properties(idx).multi_processor_count
        found = []
        with torch.cuda.device(idx):
            for hold_us in (20, 100, 400):          # (another process may hold whole CUs for a while: look again, longer)
                hist = torch.zeros(16 * 256, dtype=torch.int32, device=dev)
                check(lib().moc_cu_census(ptr(hist), 16384, hold_us, _stream()), "moc_cu_census")
                h = hist.cpu().view(16, 256)
                found = sorted(set(found) | {(x, s) for x in range(16) for s in range(256) if int(h[x, s]) > 0})
                if len(found) >= n_cu:
                    break
        _census[idx] = found
    return _census[idx]


def reserved_cus(device, n):
    """Device bitmap (int32[128], bit xcc * 256 + HW_ID[15:8]) of `n` compute units the look-ahead score pass stays off:
    an equal share of every XCD, dealt over its shader engines / arrays (highest CU id of each first).  None for n <= 0."""
    n = int(n)
    if n <= 0:
        return None
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, n)
    if key not in _reserved:
        chosen = choose_reserved_slots(cu_slots(dev), n)
        if not chosen:                      # too small a device to give any compute unit away
            import warnings
            warnings.warn(f"moc_amd: {len(cu_slots(dev))} compute units visible -- the look-ahead score pass keeps the "
                          "static whole-chip walk (MOC_RESERVE_CUS ignored)")
            _reserved[key] = None
        else:
            words = reserved_words(chosen)
            t = torch.tensor([w - (1 << 32) if w >= (1 << 31) else w for w in words], dtype=torch.int32)
            _reserved[key] = t.to(dev)
    return _reserved[key]


def choose_reserved_slots(slots, n):
    """`n` of the (xcc, HW_ID[15:8]) slots: an equal share of every XCD (the first n % XCDs get one more), inside an XCD
    dealt round-robin over its shader engines / arrays (HW_ID[15:12]), the highest CU id of each first.  Pure."""
    n = int(n)
    assert n > 0, f"{n} reserved compute units"
    # a smaller part or partition (CPX / DPX, HSA_CU_MASK), or a census that saw less: never more than three quarters of
    # what exists; an empty result tells the caller to keep the static whole-chip walk
    n = min(n, len(slots) * 3 // 4)
    if n <= 0:
        return []
    xccs = sorted({x for x, _ in slots})
    chosen = []
    per = [n // len(xccs) + (1 if i < n % len(xccs) else 0) for i in range(len(xccs))]
    for x, want in zip(xccs, per):
        groups = {}
        for _, s_ in (t for t in slots if t[0] == x):
            groups.setdefault(s_ >> 4, []).append(s_)                # HW_ID[15:12] = SE | SH; [11:8] = CU
        order = [sorted(g, reverse=True) for _, g in sorted(groups.items())]
        want = min(want, sum(len(g) for g in order) * 3 // 4)      # (an incomplete census: never more than three quarters of what was seen)
        k = 0
        while want > 0:
            g = order[k % len(order)]
            if g:
                chosen.append((x, g.pop(0)))
                want -= 1
            k += 1
    return chosen


def reserved_words(chosen):
    """The 128-word bitmap moc_batch_t.cu_reserved points at: bit xcc * 256 + HW_ID[15:8]."""
    words = [0] * 128
    for x, s_ in chosen:
        assert 0 <= x < 16 and 0 <= s_ < 256
        bit = x * 256 + s_
        words[bit >> 5] |= 1 << (bit & 31)
    return words


_stream_objs = {}


def stream_obj():
    """torch's Stream object of the current stream, cached by raw handle: Event.record() / Event.wait() without a stream
    argument go through torch.cuda.current_stream(), 7-9 us a call (device-index look-ups, an availability check that reads the
    environment) -- three of them per pass, one in front of the pass's first launch."""
    if _raw_stream is None:
        return torch.cuda.current_stream()
    dev = torch.cuda.current_device()
    key = (dev, _raw_stream(dev))
    obj = _stream_objs.get(key)
    if obj is None:
        if len(_stream_objs) > 64:
            _stream_objs.clear()
        obj = _stream_objs[key] = torch.cuda.current_stream()
    return obj


def timed_scores(batch, bank):
    """moc_scores_timed on the current stream: -> (start, stop) torch events holding the score kernel's own time stamps
    (start.elapsed_time(stop) after a synchronisation = the kernel's duration as rocprofv3 reports it)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()                                   # (torch creates the HIP event at its first record)
    e1.record()
    h0, h1 = e0.cuda_event, e1.cuda_event
    assert h0 and h1, "torch did not hand out the HIP event handles"
    check(lib().moc_scores_timed(C.byref(batch.c), ptr(bank.image), _stream(), C.c_void_p(h0), C.c_void_p(h1)), "moc_scores_timed")
    return e0, e1


def _dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return _lib.MOC_F32
    if dt == torch.bfloat16:
        return _lib.MOC_BF16
    if dt == torch.float16:
        return _lib.MOC_F16
    raise AssertionError(f"bag dtype {dt} unsupported (float32, bfloat16 or float16)")


class Bank:
    """The classifier bank [zeroshot_weights | zeroshot_weights_ext[:, C:]] re-laid
    for the score kernel (moc_prepare_bank).  Cached per (tensors, versions, dtype)."""

    _cache: dict = {}

    def __init__(self, W: torch.Tensor, W_ext: torch.Tensor, dtype: torch.dtype, device, fg_from_ext=False):
        assert W.dim() == 2 and W_ext.dim() == 2 and W.size(0) == W_ext.size(0)
        assert W_ext.size(1) > W.size(1), "logits should have more bg classes"
        self.D, self.C, self.Ce = W.size(0), W.size(1), W_ext.size(1)
        self.dtype = dtype
        Wd = W.detach().to(device=device, dtype=torch.float32).contiguous()
        Wed = W_ext.detach().to(device=device, dtype=torch.float32).contiguous()
        code = _dtype_code(dtype)
        nbytes = lib().moc_bank_bytes(self.D, self.Ce, code)
        self.image = torch.empty(nbytes, dtype=torch.uint8, device=device)
        check(lib().moc_prepare_bank(ptr(Wd), ptr(Wed), self.D, self.C, self.Ce, code, int(fg_from_ext),
                                     ptr(self.image), _stream()), "moc_prepare_bank")
        self._keep = (Wd, Wed)

    @classmethod
    def get(cls, W, W_ext, dtype, device, fg_from_ext=False) -> "Bank":
        device = torch.device(device)
        key = (W.data_ptr(), W._version, tuple(W.shape), W_ext.data_ptr(), W_ext._version,
               tuple(W_ext.shape), dtype, str(device), bool(fg_from_ext))
        hit = cls._cache.get(key)
        if hit is None:
            if len(cls._cache) > 16:
                cls._cache.clear()
            hit = cls._cache[key] = (cls(W, W_ext, dtype, device, fg_from_ext), W, W_ext)
        return hit[0]


class SlideBatch:
    """Slides packed back to back in one device array + every work array of phase A.

    X: [total_rows, D] device tensor (float32 or bfloat16); sizes: rows per slide.
    mask: optional bool/uint8 [total_rows] keep flags (host or device)."""

    def __init__(self, X: torch.Tensor, sizes: Sequence[int], C_: int, Ce: int, topj: int, topk: int,
                 discard=(), mask: torch.Tensor | None = None, x_starts: Sequence[int] | None = None):
        assert X.is_cuda and X.dim() == 2 and X.is_contiguous()
        sizes = [int(s) for s in sizes]
        assert len(sizes) > 0 and min(sizes) > 0, "empty slide"
        if x_starts is None:
            assert sum(sizes) == X.size(0), "sizes must partition X's rows"
        else:
            assert len(x_starts) == len(sizes) and all(0 <= st and st + n <= X.size(0) for st, n in zip(x_starts, sizes))
        dev = X.device
        self.X, self.sizes, self.device = X, sizes, dev
        self.n_slides, self.total, self.D = len(sizes), sum(sizes), X.size(1)
        self.C, self.Ce, self.topj, self.topk = int(C_), int(Ce), int(topj), int(topk)
        self.discard_bits = _lib.discard_bits(discard)
        off = [0]
        for s in sizes:
            off.append(off[-1] + s)
        self.row_off_host = off
        self._row_off_c = (C.c_int64 * len(off))(*off)        # host copy handed to the C ABI
        self.row_off = torch.tensor(off, dtype=torch.int64).to(dev, non_blocking=True)
        self.x_off = None
        if x_starts is not None:
            self.x_off = torch.tensor([int(v) for v in x_starts], dtype=torch.int64).to(dev, non_blocking=True)
        T, n = self.total, self.n_slides
        self.mask = None
        self.kept_rows_host = T          # rows the score pass has to read (algorithmic)
        if mask is not None:
            assert mask.numel() == T
            if not mask.is_cuda:
                self.kept_rows_host = int(mask.sum())
            self.mask = mask.to(torch.uint8).to(dev, non_blocking=True).contiguous()
        i32 = dict(dtype=torch.int32, device=dev)
        self.kept = torch.empty(T + 16, **i32) if mask is not None else None   # +16: see moc_hip.h
        self.n_kept = torch.empty(n, **i32) if mask is not None else None
        self.stats = torch.empty((2 * self.C + 3, T), dtype=torch.float32, device=dev)
        self.sel_flag = torch.empty(T, dtype=torch.uint8, device=dev)
        self.sel_idx = torch.empty(T, **i32)
        self.sel_row = torch.empty(T, dtype=torch.int64, device=dev)
        self.n_sel = torch.empty(n, **i32)
        self.cand = torch.empty((2 * self.C + 2, T), dtype=torch.float32, device=dev)
        self.c = MocBatch(
            X=ptr(X), dtype=_dtype_code(X.dtype), D=self.D, total_rows=T, n_slides=n, max_rows=max(sizes),
            row_off=ptr(self.row_off), row_off_host=C.cast(self._row_off_c, C.c_void_p), x_off=ptr(self.x_off),
            mask=ptr(self.mask), C=self.C, Ce=self.Ce, topj=self.topj,
            topk=self.topk, discard_bits=self.discard_bits, flags=0, kept=ptr(self.kept),
            n_kept=ptr(self.n_kept), stats=ptr(self.stats), sel_flag=ptr(self.sel_flag),
            sel_idx=ptr(self.sel_idx), sel_row=ptr(self.sel_row), n_sel=ptr(self.n_sel), cand=ptr(self.cand))
        self.ticket = None
        self.cu_reserved = None
        self._ws = None
        self.stats_cache = None          # (statistics of every row of X, layout flag): phase A copies instead of reading the bags
        # moc_batch_t.n_sel_host: a pinned copy of n_sel requested behind every phase A, handed to the train steps once its
        # event has completed (the look-ahead phase A of a pass ended long before the pass's steps are issued) -- the forward
        # then takes a slide's S as an argument instead of loading it, launches exactly the workgroups that have rows, and
        # the step reads fewer record keys per lane.  Masked (training) batches only.
        self._n_sel_pin = torch.empty(n, dtype=torch.int32).pin_memory() if (N_SEL_HOST and mask is not None) else None
        self._n_sel_ev = None

    def reserve_cus(self, n: int | None = None, ticket: bool | None = None):
        """This batch's score passes stay off `n` compute units (default MOC_RESERVE_CUS) and hand their tiles out by
        ticket: for a phase A that runs beside the meta-steps of another pass.  0 undoes it (static walk, whole chip);
        ticket=True with n = 0: the ticketed walk alone (tests)."""
        n = RESERVE_CUS if n is None else int(n)
        use_ticket = (n > 0) if ticket is None else bool(ticket)
        assert use_ticket or n <= 0, "reserved compute units need the ticketed walk"
        self.cu_reserved = reserved_cus(self.device, n) if n > 0 else None
        if n > 0 and self.cu_reserved is None and ticket is None:
            use_ticket = False                  # nothing could be reserved on this device: static walk, whole chip
        if use_ticket and self.ticket is None:
            self.ticket = torch.zeros(_lib.TICKET_WORDS, dtype=torch.int32, device=self.device)
        self.c.tile_ticket = ptr(self.ticket) if use_ticket else None
        self.c.cu_reserved = ptr(self.cu_reserved)

    def set_mask(self, host_mask_u8: torch.Tensor, kept_rows: int):
        """New keep flags for the same visits (next epoch): async H2D into the resident mask array."""
        assert self.mask is not None and host_mask_u8.numel() == self.total and host_mask_u8.dtype == torch.uint8
        self.mask.copy_(host_mask_u8, non_blocking=True)
        self.c.mask = ptr(self.mask)
        self.kept_rows_host = int(kept_rows)
        self.c.max_rows = max(self.sizes)       # (use_host_mask may have tightened it to ANOTHER mask's kept rows)

    def use_host_mask(self, pinned_mask_u8: torch.Tensor, kept_rows: int, max_kept: int | None = None):
        """Keep flags read by the compaction kernel straight from PINNED host memory (device-mapped by
        hipHostMalloc): no copy command at all.  Measured on the GPU box: the 480 KB asynchronous upload it
        replaces blocked the issuing thread for ~7 ms once every few dozen epochs (scripts/diag_stall.py).
        The caller keeps the buffer untouched until the pass that reads it has run."""
        assert pinned_mask_u8.is_pinned() and pinned_mask_u8.numel() == self.total and pinned_mask_u8.dtype == torch.uint8
        self._host_mask = pinned_mask_u8
        self.c.mask = pinned_mask_u8.data_ptr()
        self.kept_rows_host = int(kept_rows)
        # the flags are host bytes: the exact largest kept-row count of any slide is a tighter max_rows than the bag
        # sizes (about half) -- it sizes the grids of everything after the compaction and decides kernel shapes
        mk = max_kept if max_kept is not None else \
            int(lib().moc_host_max_kept(pinned_mask_u8.data_ptr(), C.cast(self._row_off_c, C.c_void_p), self.n_slides))
        assert 0 <= mk <= max(self.sizes)
        self.c.max_rows = max(1, mk)

    # ---- phase A ----
    def _layout(self, compact: bool, cand_from_stats: bool = False):
        """Statistics layout of the next score pass and of what reads it (include/moc_hip.h MOC_STATS_COMPACT);
        cand_from_stats: an evaluation pass -- the forward reads the candidate scores from the statistics, the
        [2C+2, S] candidate columns of wide banks are never written (MOC_CAND_FROM_STATS)."""
        self.c.flags = ((_lib.MOC_STATS_COMPACT if compact else 0) | (_lib.MOC_CAND_FROM_STATS if cand_from_stats else 0) |
                        (self.c.flags & (_lib.MOC_SELECT_PER_COLUMN | _lib.MOC_FORWARD_ROWS64 | _lib.MOC_FORWARD_FOUR_WAVES | _lib.MOC_FORWARD_ROWS16)))

    def phase_a(self, bank: Bank, for_eval: bool = False):
        assert bank.D == self.D and bank.C == self.C and bank.Ce == self.Ce and bank.dtype == self.X.dtype
        # wide banks: C + 5 statistics per row instead of 2C + 3 (the selector and the candidate gather re-form the
        # softmax columns); phase A is the only reader of its own statistics, so the layout is its private choice.
        # for_eval: only meta_forward follows (no train step, no ablation mix): candidates straight from the statistics
        self._layout(COMPACT_STATS and self.Ce > 16, cand_from_stats=bool(for_eval) and CAND_FROM_STATS and self.C > 4)
        self._n_sel_stale()
        if SCORE_EVENTS is None and self.stats_cache is None:
            check(lib().moc_phase_a(C.byref(self.c), ptr(bank.image), _stream()), "moc_phase_a")
            self._n_sel_request()
            return
        # same four launches, with events around the score pass
        check(lib().moc_mask_compact(C.byref(self.c), _stream()), "moc_mask_compact")
        self.phase_a_tail(bank)

    _n_sel_pin = None                   # (class defaults: CompactBatch builds its own fields)
    _n_sel_ev = None

    def _n_sel_stale(self):
        """n_sel is about to be rewritten: the host copy no longer describes it."""
        self.c.n_sel_host = None
        self._n_sel_ev = None

    def _n_sel_request(self):
        """Behind the launches that write n_sel, on their stream: the asynchronous copy into the pinned buffer and its event."""
        if self._n_sel_pin is None:
            return
        self.c.n_sel_host = None
        self._n_sel_pin.copy_(self.n_sel, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(stream_obj())
        self._n_sel_ev = ev

    def publish_n_sel(self, allow: bool = True):
        """Called in front of the train steps: moc_batch_t.n_sel_host = the pinned copy when its event has completed (never
        waited for), else NULL."""
        ev = self._n_sel_ev
        ok = allow and ev is not None and ev.query()
        self.c.n_sel_host = ptr(self._n_sel_pin) if ok else None
        return ok

    def phase_a_head(self, bank: Bank):
        """The first launch of phase A alone -- the kept-row lists from the keep flags (read over PCIe when they are host
        flags) -- for a caller that runs it ahead of the rest (moc_amd.runs)."""
        assert bank.D == self.D and bank.C == self.C and bank.Ce == self.Ce and bank.dtype == self.X.dtype
        self._layout(COMPACT_STATS and self.Ce > 16)
        self._n_sel_stale()
        check(lib().moc_mask_compact(C.byref(self.c), _stream()), "moc_mask_compact")

    def phase_a_tail(self, bank: Bank):
        """... and the rest: score pass (timed when bench.py collects SCORE_EVENTS), selection, union, candidates."""
        if self.stats_cache is not None:
            # opt-in (include/moc_hip.h moc_scores_from_cache): the kept rows' statistics from those of an earlier, unmasked
            # score pass over the same rows and bank -- the same bits, no read of the bags
            cache, compact = self.stats_cache
            assert compact == bool(self.c.flags & _lib.MOC_STATS_COMPACT) and cache.size(1) == self.X.size(0)
            check(lib().moc_scores_from_cache(C.byref(self.c), ptr(cache), cache.size(1), _stream()), "moc_scores_from_cache")
        elif SCORE_EVENTS is None:
            check(lib().moc_scores(C.byref(self.c), ptr(bank.image), _stream()), "moc_scores")
        else:
            e0, e1 = timed_scores(self, bank)
            SCORE_EVENTS.append((e0, e1, self.kept_rows_host * self.D * self.X.element_size()))
        self._n_sel_stale()
        self.select()
        self.gather_candidates()
        self._n_sel_request()

    def scores(self, bank: Bank):
        self._layout(False)                      # callers of scores() read `stats` themselves: the full layout
        check(lib().moc_mask_compact(C.byref(self.c), _stream()), "moc_mask_compact")
        check(lib().moc_scores(C.byref(self.c), ptr(bank.image), _stream()), "moc_scores")

    def select(self):
        self._n_sel_stale()
        check(lib().moc_select(C.byref(self.c), _stream()), "moc_select")

    def gather_candidates(self, with_feat=False):
        feat = torch.empty((self.total, self.D), dtype=self.X.dtype, device=self.device) if with_feat else None
        check(lib().moc_gather_candidates(C.byref(self.c), ptr(feat), _stream()), "moc_gather_candidates")
        return feat

    # ---- phase B work arrays ----
    def meta_ws(self):
        if self._ws is None:
            dev, T, n, Cc, K = self.device, self.total, self.n_slides, self.C, self.topk
            f32 = dict(dtype=torch.float32, device=dev)
            i32 = dict(dtype=torch.int32, device=dev)
            t = dict(
                H1=torch.empty((T, HIDDEN), **f32), gates=torch.empty((T, 4), **f32),
                mixed=torch.empty((Cc, T), **f32), pooled=torch.empty((n, Cc), **f32),
                topk_idx=torch.empty((n, Cc, K), **i32), topk_cnt=torch.empty((n, Cc), **i32),
                loss=torch.empty(n, **f32), pred=torch.empty(n, **i32),
                pair_dh=torch.empty((Cc * K, HIDDEN), **f32),
                W2_alt=torch.empty((4, HIDDEN), **f32),
                pair_row=torch.empty(Cc * K, dtype=torch.int64, device=dev), n_pair=torch.zeros(1, **i32))
            # tile records (include/moc_hip.h moc_meta_ws_t.tile_ws): what the training forward leaves for the step kernel
            # (only where the tile-record step can run -- narrow banks -- and only for batches that train: masked ones and
            # the gathered batches of the exact-sequential mode; an evaluation batch of 3 M rows x 30 classes would carry 1 GB)
            trains = self.mask is not None or isinstance(self, CompactBatch)
            nb = lib().moc_tile_ws_bytes(T, n, Cc) if (TILE_RECORDS and trains and Cc <= 16 and K <= 16 and Cc * K <= 64) else 0
            t["tile_ws"] = torch.empty(nb, dtype=torch.uint8, device=dev) if nb else None
            self._ws = (t, MocMetaWs(**{k: ptr(v) for k, v in t.items()}, tile_ws_bytes=nb))
        return self._ws


def build_stats_cache(X: torch.Tensor, sizes, starts, bank: "Bank", topj: int, topk: int):
    """The statistics of EVERY row of X (one unmasked score pass over the slides `sizes` at rows `starts`, which must cover
    what later passes visit) -> (stats_all [rows of stats, X rows] indexed by the row's position in X, compact flag) -- what
    SlideBatch.stats_cache takes.  28 bytes per row at two classes."""
    tmp = SlideBatch(X, sizes, bank.C, bank.Ce, topj, topk, (), x_starts=starts)
    compact = COMPACT_STATS and bank.Ce > 16
    tmp._layout(compact)
    check(lib().moc_mask_compact(C.byref(tmp.c), _stream()), "moc_mask_compact")
    check(lib().moc_scores(C.byref(tmp.c), ptr(bank.image), _stream()), "moc_scores")
    ns = (bank.C + 5) if compact else (2 * bank.C + 3)
    out = torch.zeros((ns, X.size(0)), dtype=torch.float32, device=X.device)
    off = tmp.row_off_host
    for b, (st, n) in enumerate(zip(starts, sizes)):      # slot order of an unmasked pass = row order of the slide
        out[:, st:st + n] = tmp.stats[:ns, off[b]:off[b] + n]
    return out, compact


class CompactBatch(SlideBatch):
    """Phase A's RESULT for n slides with no bag behind it: the selected rows themselves (`X` [rows, D]), their
    candidate scores (`cand` [2C+2, rows]) and `n_sel` [n] -- what moc_pack_selected_rows writes on every rank and
    the exact-sequential multi-GPU mode all-gathers (dist.train_seq).  The phase-B entry points take it like any
    batch: sel_row is the identity, slide b's rows are X[row_off[b] : row_off[b] + n_sel[b]] -- and row_off is set
    per pass (set_layout): the gathered pieces are as long as the pass's selections make them."""

    def __init__(self, n_slides: int, rows: int, cap: int, D: int, dtype: torch.dtype, C_: int, Ce: int, topj: int, topk: int,
                 device, X: torch.Tensor | None = None):
        T = int(rows)
        self.device, self.n_slides, self.total, self.D = device, int(n_slides), T, int(D)
        self.C, self.Ce, self.topj, self.topk, self.cap = int(C_), int(Ce), int(topj), int(topk), int(cap)
        self.sizes = [cap] * n_slides
        self.X = X if X is not None else torch.zeros((T, D), dtype=dtype, device=device)
        assert tuple(self.X.shape) == (T, D) and self.X.dtype == dtype
        self.cand = torch.zeros((2 * self.C + 2, T), dtype=torch.float32, device=device)
        self.n_sel = torch.zeros(n_slides, dtype=torch.int32, device=device)
        self.sel_row = torch.arange(T, dtype=torch.int64, device=device)
        self.row_off_host = [0] * (n_slides + 1)
        self._row_off_c = (C.c_int64 * (n_slides + 1))()
        self._row_off_pin = torch.zeros(n_slides + 1, dtype=torch.int64)
        if torch.device(device).type == "cuda":
            self._row_off_pin = self._row_off_pin.pin_memory()
        self.row_off = torch.zeros(n_slides + 1, dtype=torch.int64, device=device)
        self.discard_bits, self.mask, self.kept_rows_host = 0, None, T
        self.c = MocBatch(
            X=ptr(self.X), dtype=_dtype_code(dtype), D=self.D, total_rows=T, n_slides=self.n_slides, max_rows=cap,
            row_off=ptr(self.row_off), row_off_host=C.cast(self._row_off_c, C.c_void_p), x_off=None, mask=None,
            C=self.C, Ce=self.Ce, topj=self.topj, topk=self.topk, discard_bits=0, flags=0, kept=None, n_kept=None,
            stats=None, sel_flag=None, sel_idx=None, sel_row=ptr(self.sel_row), n_sel=ptr(self.n_sel), cand=ptr(self.cand))
        self._ws = None

    def set_layout(self, row_off):
        """First row of every slide (+ the end) for the pass whose pieces were just gathered: host copy for the kernel
        arguments of the single-slide launches, device copy (stream-ordered, from a pinned buffer the caller does not
        touch again before this set's steps have run) for the kernels that read row_off themselves."""
        assert len(row_off) == self.n_slides + 1 and row_off[-1] <= self.total
        self.row_off_host = [int(v) for v in row_off]
        for i, v in enumerate(self.row_off_host):
            self._row_off_c[i] = v
        self._row_off_pin.copy_(torch.tensor(self.row_off_host, dtype=torch.int64))
        self.row_off.copy_(self._row_off_pin, non_blocking=True)


def pack_selected(batch: SlideBatch, slide0: int, n: int, cap: int, feat_out: torch.Tensor, cand_out: torch.Tensor):
    """moc_pack_selected: slides [slide0, slide0+n) of `batch` (phase A done) -> feat_out [n, cap, D], cand_out [n, 2C+2, cap]."""
    assert feat_out.is_contiguous() and cand_out.is_contiguous() and feat_out.dtype == batch.X.dtype
    assert feat_out.numel() >= n * cap * batch.D and cand_out.numel() >= n * (2 * batch.C + 2) * cap
    check(lib().moc_pack_selected(C.byref(batch.c), slide0, n, cap, ptr(feat_out), ptr(cand_out), _stream()), "moc_pack_selected")


def pack_selected_rows(batch: SlideBatch, slide0: int, n: int, cap: int, feat_out: torch.Tensor, cand_out: torch.Tensor):
    """moc_pack_selected_rows: slides [slide0, slide0+n) of `batch` (phase A done), unpadded -> feat_out [rows, D] (slide
    b's rows behind those of the slides before it), cand_out [rows, 2C+2] row-major."""
    assert feat_out.is_contiguous() and cand_out.is_contiguous() and feat_out.dtype == batch.X.dtype
    rows = feat_out.size(0)
    assert feat_out.size(1) == batch.D and tuple(cand_out.shape) == (rows, 2 * batch.C + 2)
    check(lib().moc_pack_selected_rows(C.byref(batch.c), slide0, n, cap, ptr(feat_out), ptr(cand_out), rows, _stream()),
          "moc_pack_selected_rows")


class MetaState:
    """Views of a senet's parameters and its torch.optim.Adam state as the C ABI
    wants them.  The tensors are the optimizer's own (updated in place), so
    `optimizer.state_dict()` / `model.state_dict()` stay what the reference saves
    (main_moc.py:628)."""

    def __init__(self, model, optimizer=None, need_grads=False, bag_dtype=torch.bfloat16):
        lin1, lin2 = model.model[0], model.model[2]
        self.params = [lin1.weight, lin1.bias, lin2.weight, lin2.bias]
        assert lin1.out_features == HIDDEN and lin2.out_features == 4 and lin2.in_features == HIDDEN
        for p in self.params:
            assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous(), "senet must be fp32 on the GPU"
        self.D = lin1.in_features
        self.optimizer = optimizer
        self._group, self._graph, self._graph_ws, self._graph_ok = None, None, None, False
        kw = dict(lr=0.0, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, H=HIDDEN, D=self.D, step=0)
        names = ("W1", "b1", "W2", "b2")
        ptrs = {n: ptr(p.data) for n, p in zip(names, self.params)}
        self.grads = None
        self._state_refs = []
        if optimizer is not None:
            assert isinstance(optimizer, torch.optim.Adam), "the fused step implements torch.optim.Adam only"
            groups = [g for g in optimizer.param_groups if any(p is q for p in g["params"] for q in self.params)]
            assert len(groups) == 1, "senet parameters must sit in one param group"
            g = self._group = groups[0]
            assert not g.get("amsgrad", False) and not g.get("maximize", False), "amsgrad/maximize unsupported"
            steps = set()
            for n, p in zip(names, self.params):
                st = optimizer.state[p]
                if len(st) == 0:   # same lazy init as torch.optim.Adam._init_group
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                ptrs["m_" + n], ptrs["v_" + n] = ptr(st["exp_avg"]), ptr(st["exp_avg_sq"])
                self._state_refs.append((st["exp_avg"], st["exp_avg_sq"]))      # meta.c holds their raw addresses
                steps.add(int(float(st["step"])))
            assert len(steps) == 1, "parameters disagree on the Adam step count"
            b1, b2 = g["betas"]
            kw.update(lr=float(g["lr"]), beta1=float(b1), beta2=float(b2), eps=float(g["eps"]),
                      weight_decay=float(g["weight_decay"]), step=steps.pop())
        if need_grads:
            self.grads = [torch.zeros_like(p) for p in self.params]
            for n, gt in zip(names, self.grads):
                ptrs["g_" + n] = ptr(gt)
        # scratch for the forward pass's operand-ordered copy of W1 (sized for the larger, bf16x3 form)
        self.w1_image = torch.empty(max(lib().moc_w1_image_bytes(self.D, _lib.MOC_BF16),
                                        lib().moc_w1_image_bytes(self.D, _lib.MOC_F32)),
                                    dtype=torch.uint8, device=self.params[0].device)
        ptrs["W1_image"] = ptr(self.w1_image)
        self.c = MocMeta(**ptrs, **kw)

    # ---- one MetaState per (model, optimizer), reused from pass to pass: building the views costs 20-40 us of
    # Python, and the pass graphs (moc_train_steps_graph) are keyed on the tensors it points at
    _cache = weakref.WeakKeyDictionary()

    @classmethod
    def cached(cls, model, optimizer):
        ent = cls._cache.get(model)
        if ent is not None:
            meta, opt_ref, sig = ent
            if opt_ref() is optimizer and sig == cls._signature(meta.params, optimizer):
                meta.refresh()
                return meta
        meta = cls(model, optimizer)
        meta._graph_ok = True          # lives from pass to pass: worth capturing its passes
        cls._cache[model] = (meta, weakref.ref(optimizer), cls._signature(meta.params, optimizer))
        return meta

    @staticmethod
    def _signature(params, optimizer):
        sig = []
        for p in params:
            st = optimizer.state.get(p) or {}
            m, v = st.get("exp_avg"), st.get("exp_avg_sq")
            # storage addresses, not object identities: `.data = ` / `set_()` keep the id and move the storage, and the
            # fused kernels write through the raw pointers in meta.c
            sig.append((p.data_ptr(), m.data_ptr() if m is not None else 0, v.data_ptr() if v is not None else 0,
                        id(st.get("step"))))
        return tuple(sig)

    def refresh(self):
        """Hyper-parameters and step count as the optimizer holds them NOW (someone may have changed the learning
        rate, stepped it, or loaded a state dict in place)."""
        g = self._group
        b1, b2 = g["betas"]
        c = self.c
        c.lr, c.beta1, c.beta2, c.eps, c.weight_decay = float(g["lr"]), float(b1), float(b2), float(g["eps"]), float(g["weight_decay"])
        st = self.optimizer.state
        s0 = int(st[self.params[0]]["step"])
        assert all(int(st[p]["step"]) == s0 for p in self.params[1:]), "parameters disagree on the Adam step count"
        c.step = s0

    def step_graph(self):
        """The moc_step_graph_t of this meta-learner (created on first use).  None when switched off, and for a
        MetaState built for one call (not through `cached`): its graphs would be captured and thrown away."""
        if not (STEP_GRAPH and self._graph_ok):
            return None
        if self._graph is None:
            nbytes = lib().moc_step_graph_workspace_bytes(GRAPH_TABLE_STEPS)
            self._graph_ws = torch.empty(nbytes, dtype=torch.uint8, device=self.params[0].device)
            h = C.c_void_p()
            check(lib().moc_step_graph_create(ptr(self._graph_ws), nbytes, C.byref(h)), "moc_step_graph_create")
            self._graph = h
            weakref.finalize(self, lib().moc_step_graph_destroy, h).atexit = False   # (at exit the runtime goes first)
        return self._graph

    def graph_stats(self):
        """(captures, replays, passes that fell back to stream launches) of this meta-learner's pass graphs."""
        if self._graph is None:
            return (0, 0, 0)
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        check(lib().moc_step_graph_stats(self._graph, C.byref(a), C.byref(b), C.byref(c)), "moc_step_graph_stats")
        return a.value, b.value, c.value

    def advance(self, n_steps: int):
        """Record n fused Adam steps in the optimizer's own step counters."""
        self.c.step += n_steps
        for p in self.params:
            st = self.optimizer.state[p]["step"]
            st += n_steps   # tensor in-place (host or device scalar)


def draw_row_masks(total: int, out: torch.Tensor | None = None):
    """`total` keep flags from the CPU default generator, bit for bit what consecutive
    `torch.rand(n_i) > 0.5` calls (sum n_i == total) produce (main_moc.py:330), leaving the generator
    where they would.  Uses the library's mt19937 replay (several times faster than torch.rand);
    falls back to torch.rand itself when the generator state is not the layout it knows.
    Returns (uint8 host tensor [total], number of kept rows)."""
    if out is None:
        out = torch.empty(total, dtype=torch.uint8)
    if torch.get_default_dtype() == torch.float32:
        st = torch.get_rng_state()
        kept = lib().moc_host_draw_masks(ptr(st), st.numel(), total, ptr(out))
        if kept >= 0:
            torch.set_rng_state(st)
            return out, int(kept)
    m = torch.rand(total) > 0.5
    out.copy_(m)
    return out, int(m.sum())


def draw_row_masks_from(rng_state: torch.Tensor, total: int, out: torch.Tensor):
    """As draw_row_masks, but from the given generator state (a `torch.get_rng_state()` tensor), without
    touching torch's generator: -> (out, kept rows, state after the draws).  When the state's layout is not
    the one the library knows, torch itself draws from ITS generator and the third value is None."""
    if torch.get_default_dtype() == torch.float32:
        st = rng_state.clone()
        kept = lib().moc_host_draw_masks(ptr(st), st.numel(), total, ptr(out))
        if kept >= 0:
            return out, int(kept), st
    m = torch.rand(total) > 0.5
    out.copy_(m)
    return out, int(m.sum()), None


class MaskDrawer:
    """The keep flags of a pass, drawn one pass AHEAD on a helper thread (the reference has a DataLoader worker process
    beside its main thread; here the second host thread replays torch's mt19937: moc_host_draw_masks releases the GIL).

    The stream is sequential, so the flags of the pass after next follow from the generator state the last draw left:
    as soon as a draw for (state S, row layout L) is handed out, the draw for (S', L) starts in the background into a
    spare pinned buffer.  `take(S', L)` then returns it at once; any other request (someone else drew from the
    generator, another pass length) discards the speculation and draws in line -- the bits are always the ones
    `torch.rand(N) > 0.5` would give from the requested state.  Buffers are pinned host memory read in place by the
    compaction kernel: one is handed out again only after the event the caller attached to it has completed."""

    def __init__(self, total: int, row_off_c, n_slides: int, n_buffers: int = 4, pinned: bool = True):
        from concurrent.futures import ThreadPoolExecutor
        self.total, self.row_off_c, self.n_slides = int(total), row_off_c, int(n_slides)
        self.bufs = [torch.empty(self.total, dtype=torch.uint8) for _ in range(n_buffers)]
        if pinned:                                # (False: the CPU tests of the draw-ahead logic)
            self.bufs = [b.pin_memory() for b in self.bufs]
        self.busy = [None] * n_buffers            # event after the phase A that reads buffer i (None: free)
        self.pool = ThreadPoolExecutor(1)
        self.ahead = None                         # (state_before, buffer index, future)

    def _draw(self, state_before: torch.Tensor, i: int):
        st = state_before.clone()
        kept = lib().moc_host_draw_masks(ptr(st), st.numel(), self.total, ptr(self.bufs[i]))
        if kept < 0:
            return None
        mk = lib().moc_host_max_kept(self.bufs[i].data_ptr(), C.cast(self.row_off_c, C.c_void_p), self.n_slides)
        return int(kept), int(mk), st

    def _free_buffer(self, exclude=()):
        for i, ev in enumerate(self.busy):
            if i in exclude:
                continue
            if ev is None or ev.query():
                self.busy[i] = None
                return i
        for i, ev in enumerate(self.busy):        # none free: wait for the oldest the caller is done with
            if i not in exclude:
                ev.synchronize()
                self.busy[i] = None
                return i
        raise AssertionError("MaskDrawer: no buffer")

    def take(self, state_before: torch.Tensor, chain: bool = True):
        """-> (pinned flags, kept rows, max kept rows of a slide, generator state after, buffer index) or None when
        the generator state is not the layout the replay knows (the caller lets torch draw).  `chain` False: the pass
        after this one has another row layout (the caller starts ITS drawer): nothing is drawn ahead here."""
        if torch.get_default_dtype() != torch.float32:
            return None
        got = None
        if self.ahead is not None:
            st0, i, fut = self.ahead
            self.ahead = None
            res = fut.result()
            if res is not None and torch.equal(st0, state_before):
                got = (i, res)
            # (a discarded speculation leaves buffer i free: nothing on the GPU reads it)
        if got is None:
            i = self._free_buffer()
            res = self._draw(state_before, i)
            if res is None:
                return None
            got = (i, res)
        i, (kept, mk, st_after) = got
        if chain:                                 # the flags of the pass after this one, in the background
            j = self._free_buffer(exclude=(i,))
            self.ahead = (st_after, j, self.pool.submit(self._draw, st_after, j))
        return self.bufs[i], kept, mk, st_after, i

    def prefetch(self, state_before: torch.Tensor):
        """Start drawing the flags that follow `state_before` now (no-op when that is what is being drawn already):
        for a caller that learns early which pass comes next -- the draw then runs beside its kernel launches."""
        if torch.get_default_dtype() != torch.float32:
            return
        if self.ahead is not None:
            if torch.equal(self.ahead[0], state_before):
                return
            self.ahead[2].result()                # let the stale draw finish: its buffer is free again
            self.ahead = None
        j = self._free_buffer()
        self.ahead = (state_before.clone(), j, self.pool.submit(self._draw, state_before, j))

    def attach(self, i: int, event):
        """`event` completes when the GPU work that reads buffer i has run."""
        self.busy[i] = event


def train_use_bits(discard) -> int:
    return (~_lib.discard_bits(discard)) & 15            # main_moc.py:396-403


def eval_use_bits(discard) -> int:
    """main_moc.py:486-492: psi_p always; the last test is for "delta_bottomk"."""
    d = discard or ()
    return 1 | (0 if "delta_softmax" in d else 2) | (0 if "delta_diff" in d else 4) | (0 if "delta_bottomk" in d else 8)


def meta_forward(batch: SlideBatch, meta: MetaState, slide0: int, n: int, use_bits: int, keep_hidden: bool = True):
    """keep_hidden=False (evaluation): H1 and the gates are not written -- only the backward pass reads them."""
    _, ws = batch.meta_ws()
    if not keep_hidden:
        ws = type(ws).from_buffer_copy(ws)
        ws.H1 = None
        ws.gates = None
    check(lib().moc_meta_forward(C.byref(batch.c), C.byref(meta.c), C.byref(ws), slide0, n, use_bits, _stream()),
          "moc_meta_forward")


def mix_fixed(batch: SlideBatch, slide0: int, n: int, mode: str):
    _, ws = batch.meta_ws()
    code = {"avg": 0, "sum": 1, "max": 2}[mode]
    check(lib().moc_mix_fixed(C.byref(batch.c), C.byref(ws), slide0, n, code, _stream()), "moc_mix_fixed")


def pool_loss(batch: SlideBatch, labels: torch.Tensor, slide0: int, n: int):
    _, ws = batch.meta_ws()
    check(lib().moc_pool_loss(C.byref(batch.c), C.byref(ws), ptr(labels), slide0, n, _stream()), "moc_pool_loss")


def loss_only(batch: SlideBatch, labels: torch.Tensor, slide0: int, n: int):
    """CE + argmax of ws.pooled[slide0:slide0+n] (already filled) into ws.loss / ws.pred."""
    t, _ = batch.meta_ws()
    check(lib().moc_ce_loss(ptr(t["pooled"][slide0:]), ptr(labels[slide0:]), n, batch.C,
                            ptr(t["loss"][slide0:]), ptr(t["pred"][slide0:]), _stream()), "moc_ce_loss")


def train_steps(batch: SlideBatch, meta: MetaState, labels: torch.Tensor, slide0: int, n: int, use_bits: int):
    """n consecutive meta-steps (one slide each) with Adam applied in place: one graph launch per pass when the
    meta-learner has a step graph (MOC_STEP_GRAPH=1, opt-in), 2 n + 1 stream launches otherwise -- same kernels,
    same coefficient floats, bit-identical parameters."""
    _, ws = batch.meta_ws()
    g = meta.step_graph()
    # (a pass graph bakes its grids in: the device's n_sel there)
    batch.publish_n_sel(allow=g is None)
    if g is not None:
        check(lib().moc_train_steps_graph(g, C.byref(batch.c), C.byref(meta.c), C.byref(ws), ptr(labels), slide0, n,
                                          use_bits, _stream()), "moc_train_steps_graph")
    else:
        check(lib().moc_train_steps(C.byref(batch.c), C.byref(meta.c), C.byref(ws), ptr(labels), slide0, n,
                                    use_bits, _stream()), "moc_train_steps")
    meta.advance(n)


def train_steps_dp(batch: SlideBatch, meta: MetaState, labels: torch.Tensor, slide0: int, n: int, use_bits: int,
                   grad_flat: torch.Tensor, allreduce_fn, comm, world: int):
    """n synchronous data-parallel steps on this rank's slides; `allreduce_fn` is the address of a
    function with ncclAllReduce's signature (None at world 1).  Does not advance the step counters."""
    _, ws = batch.meta_ws()
    check(lib().moc_train_steps_dp(C.byref(batch.c), C.byref(meta.c), C.byref(ws), ptr(labels), slide0, n,
                                   use_bits, ptr(grad_flat), grad_flat.numel(), allreduce_fn, comm, world,
                                   _stream()), "moc_train_steps_dp")


def fused_step_shape(batch: SlideBatch) -> bool:
    """Whether the node-local exchange step applies -- decided from C, topk, D, topj only."""
    c = batch.c
    return bool(lib().moc_p2p_step_supported(c.C, c.topk, c.D, c.topj))


def train_steps_p2p(batch: SlideBatch, meta: MetaState, labels: torch.Tensor, slide0: int, n: int, use_bits: int,
                    comm_handle):
    """n synchronous data-parallel steps with the gradient exchange inside the step kernel
    (moc_p2p_*: peer-mapped receive buffers on one node).  Does not advance the step counters."""
    _, ws = batch.meta_ws()
    check(lib().moc_train_steps_p2p(C.byref(batch.c), C.byref(meta.c), C.byref(ws), ptr(labels), slide0, n,
                                    use_bits, comm_handle, _stream()), "moc_train_steps_p2p")


def train_grad(batch: SlideBatch, meta: MetaState, labels: torch.Tensor, slide: int, use_bits: int):
    """Forward + loss + gradients of one slide into meta.grads (no update)."""
    _, ws = batch.meta_ws()
    check(lib().moc_train_grad(C.byref(batch.c), C.byref(meta.c), C.byref(ws), ptr(labels), slide, use_bits,
                               _stream()), "moc_train_grad")


def adam_step(meta: MetaState, grad_scale: float = 1.0, advance: bool = True):
    check(lib().moc_adam_step(C.byref(meta.c), C.c_float(grad_scale), _stream()), "moc_adam_step")
    if advance:
        meta.advance(1)


def topk_mean(keys: torch.Tensor, vals: torch.Tensor, K: int, smallest=False, key_shared=False,
              want_idx=False, seg_off: torch.Tensor | None = None):
    """keys/vals: [C, N] device fp32 (keys may be [1, N] with key_shared).  Returns
    pooled [n_seg, C] (and idx [n_seg, C, K], cnt [n_seg, C])."""
    assert keys.is_cuda and vals.is_cuda and keys.dtype == vals.dtype == torch.float32
    keys, vals = keys.contiguous(), vals.contiguous()
    Cc, N = vals.shape
    dev = vals.device
    if seg_off is None:
        seg_off = torch.tensor([0, N], dtype=torch.int64, device=dev)
    n_seg = seg_off.numel() - 1
    pooled = torch.empty((n_seg, Cc), dtype=torch.float32, device=dev)
    idx = torch.empty((n_seg, Cc, K), dtype=torch.int32, device=dev) if want_idx else None
    cnt = torch.empty((n_seg, Cc), dtype=torch.int32, device=dev) if want_idx else None
    check(lib().moc_topk_mean(ptr(keys), 0 if key_shared else N, ptr(vals), N, ptr(seg_off), None, n_seg, Cc,
                              int(K), int(bool(smallest)), ptr(pooled), ptr(idx), ptr(cnt), _stream()),
          "moc_topk_mean")
    return (pooled, idx, cnt) if want_idx else pooled


def gated_attention_pool(h: torch.Tensor, Wa, ba, Wb, bb, Wc, bc):
    """SURVEY.md section 8 row f4 (models/model_clam.py:41-64, :178-183, :206): h [N, L] fp32 on the GPU,
    Wa / Wb [D, L], Wc [K, D] in nn.Linear layout.  -> (A_raw [K, N], M [K, L])."""
    ts = [h, Wa, ba, Wb, bb, Wc, bc]
    assert all(t.is_cuda and t.dtype == torch.float32 for t in ts), "gated_attention_pool: fp32 tensors on the GPU"
    h, Wa, ba, Wb, bb, Wc, bc = [t.detach().contiguous() for t in ts]
    N, L = h.shape
    D, K = Wa.shape[0], Wc.shape[0]
    assert Wa.shape == Wb.shape == (D, L) and Wc.shape == (K, D) and ba.numel() == bb.numel() == D and bc.numel() == K
    dev = h.device
    A_raw = torch.empty((K, N), dtype=torch.float32, device=dev)
    M = torch.empty((K, L), dtype=torch.float32, device=dev)
    nbytes = lib().moc_gated_attention_workspace(N, L, D, K)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    check(lib().moc_gated_attention_pool(ptr(h), N, L, ptr(Wa), ptr(ba), ptr(Wb), ptr(bb), D, ptr(Wc), ptr(bc), K,
                                         ptr(A_raw), ptr(M), ptr(ws), nbytes, _stream()), "moc_gated_attention_pool")
    return A_raw, M


def gated_attention_backward(h, Wa, ba, Wb, bb, Wc, A_raw, gA=None, gM=None):
    """The backward of gated_attention_pool (include/moc_hip.h moc_gated_attention_backward; autograd's work behind
    models/model_clam.py:58-63, :178-183, :206).  One recompute pass in HIP, then two plain GEMMs as library calls.
    -> (dh [N, L], dWa [D, L], dba [D], dWb [D, L], dbb [D], dWc [K, D], dbc [K])."""
    ts = [h, Wa, ba, Wb, bb, Wc, A_raw] + [g for g in (gA, gM) if g is not None]
    assert all(t.is_cuda and t.dtype == torch.float32 for t in ts), "gated_attention_backward: fp32 tensors on the GPU"
    h, Wa, ba, Wb, bb, Wc, A_raw = [t.detach().contiguous() for t in (h, Wa, ba, Wb, bb, Wc, A_raw)]
    gA = gA.detach().contiguous() if gA is not None else None
    gM = gM.detach().contiguous() if gM is not None else None
    N, L = h.shape
    D, K = Wa.shape[0], Wc.shape[0]
    assert Wa.shape == Wb.shape == (D, L) and Wc.shape == (K, D) and A_raw.shape == (K, N)
    assert (gA is None or gA.shape == (K, N)) and (gM is None or gM.shape == (K, L))
    dev = h.device
    S = lib().moc_gated_attention_dab_stride(D, K)
    dab = torch.empty((N, S), dtype=torch.float32, device=dev)     # da | db | p | 0
    ds = torch.empty((K, N), dtype=torch.float32, device=dev)
    dcol = torch.empty(((2 + K) * D,), dtype=torch.float32, device=dev)
    dbc = torch.empty((K,), dtype=torch.float32, device=dev)
    nbytes = lib().moc_gated_attention_backward_workspace(N, L, D, K)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    check(lib().moc_gated_attention_backward(ptr(h), N, L, ptr(Wa), ptr(ba), ptr(Wb), ptr(bb), D, ptr(Wc), K, ptr(A_raw),
                                             ptr(gA) if gA is not None else None, ptr(gM) if gM is not None else None,
                                             ptr(dab), ptr(ds), ptr(dcol), ptr(dbc), ptr(ws), nbytes, _stream()),
          "moc_gated_attention_backward")
    X = torch.zeros((S, L), dtype=torch.float32, device=dev)       # [Wa; Wb; gM; 0]: M = p h rides in the same GEMM
    X[:D], X[D:2 * D] = Wa, Wb
    if gM is not None:
        X[2 * D:2 * D + K] = gM
    dh = dab @ X
    dW = dab.t() @ h                                               # [S, L]: dWa over dWb (over M)
    return dh, dW[:D], dcol[:D], dW[D:2 * D], dcol[D:2 * D], dcol[2 * D:].view(K, D), dbc
