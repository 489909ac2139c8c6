"""Several independent training runs in ONE process, stepped in lockstep (include/moc_hip.h: moc_train_steps_runs).

The reference trains folds x shots of the same loop as separate processes -- `scripts/moc_train.sh:11-31` starts one
`python main_moc.py` per (fold, shot) and packs five of them onto a GPU.  One run is a chain of dependent meta-steps
(main_moc.py:380-410) that leaves most of an MI355X idle; here R runs share every launch: one forward and one step launch
per meta-step serve all of them (grid.y / grid.z = run), and phase A (mask -> scores -> selectors -> union, parameter free)
runs over all R x n slides in the same four launches, one pass ahead on a side stream.

Every run stays the exact recurrence of `main_moc.train`: its own slides in loader order, its own parameters and Adam
state, its own stream of row masks -- `torch.rand(N) > 0.5` per slide drawn from the run's OWN CPU generator (the
reference's runs are separate processes, each with its own default generator).  Per run the result is bit-identical to
training it alone with `main_moc.train` from the same generator state (tests/test_gpu_runs.py).

What the runs share: the classifier bank, the hyper-parameters of Adam, the number of visits per pass, topj / topk /
discard_classifiers.  The models' parameter tensors and the optimizers' moments are re-seated as views into one arena
per kind (state_dict() / load_state_dict() keep working; the tensors' values are preserved).
"""
from __future__ import annotations

import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import torch

from . import engine
from ._lib import MocRuns, check, lib, ptr
from .engine import HIDDEN, MetaState, SlideBatch

MAX_RUNS = 16


class TrainRuns:
    """R (model, optimizer, resident train split) triples trained in lockstep.  `generators`: one CPU torch.Generator
    per run -- the run's mask stream (default: fresh generators seeded from the default generator, in run order)."""

    def __init__(self, models, optimizers, splits, device, args, generators=None):
        from . import main_moc as M
        R = len(models)
        assert 1 <= R <= MAX_RUNS and len(optimizers) == R and len(splits) == R, f"1 .. {MAX_RUNS} runs"
        assert all(isinstance(sp, M.ResidentBags) for sp in splits), "train_runs: resident splits (main_moc.ResidentBags)"
        self.R, self.models, self.optimizers, self.splits, self.device = R, list(models), list(optimizers), list(splits), device
        n = len(splits[0])
        assert all(len(sp) == n for sp in splits), "train_runs: every run must make the same number of visits per pass"
        self.n = n
        dt, D = splits[0].X.dtype, splits[0].X.size(1)
        assert all(sp.X.dtype == dt and sp.X.size(1) == D for sp in splits), "train_runs: one storage type and width"
        assert not any(sp.loader_seed_draw for sp in splits), "train_runs: loader_seed_draw splits are not batched"
        if generators is None:
            generators = []
            for _ in range(R):
                g = torch.Generator()
                g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
                generators.append(g)
        assert len(generators) == R and all(g is not torch.default_generator for g in generators), \
            "train_runs: one private CPU generator per run (the masks of a pass are drawn a pass ahead)"
        self.generators = list(generators)
        # ---- all runs' bags in one packed array (a one-time copy), the visits of run r at slides r * n ...
        self.X = torch.cat([sp.X for sp in splits], 0) if R > 1 else splits[0].X
        sizes, starts, labels, row0 = [], [], [], 0
        self.run_rows = []                                   # (first flag, flags) of each run's pass
        for sp in splits:
            order = sp.visit_order()
            rows = 0
            for k in order:
                sizes.append(sp.sizes[k])
                starts.append(row0 + sp.starts[k])
                labels.append(sp.labels[k])
                rows += sp.sizes[k]
            self.run_rows.append((sum(s_ for s_ in sizes) - rows, rows))
            row0 += sp.X.size(0)
        bank = M._bank_for(self.X, device)
        assert bank.C == args.n_classes
        self.bank, self.args = bank, args
        T = sum(sizes)
        self.batches = [SlideBatch(self.X, sizes, bank.C, bank.Ce, args.topj, args.topk, args.discard_classifiers,
                                   mask=torch.ones(T, dtype=torch.uint8), x_starts=starts) for _ in range(2)]
        # (round 4 measurements, profiles/NOTES.md: the meta-steps crawl while a score pass streams beside them whether or not
        # it leaves them compute units, so phase A here runs at full width; MOC_RUNS_RESERVE_CUS brings the reservation back)
        self.reserve = int(os.environ.get("MOC_RUNS_RESERVE_CUS", "0"))
        self.lookahead = os.environ.get("MOC_RUNS_LOOKAHEAD", "0") != "0"
        self.upload_flags = os.environ.get("MOC_RUNS_UPLOAD_FLAGS", "0") != "0"    # (measured: no gain from the copy)
        if bank.Ce <= 16 and self.reserve > 0:
            for b in self.batches:
                b.reserve_cus(self.reserve)
        if os.environ.get("MOC_CACHE_SCORES", "0") == "1" or all(sp.cache_scores for sp in splits):
            # opt-in (main_moc.ResidentBags cache_scores): the statistics of every row once, then no score pass per epoch
            all_sizes, all_starts, r0_ = [], [], 0
            for sp in splits:
                all_sizes += sp.sizes
                all_starts += [r0_ + st for st in sp.starts[:-1]]
                r0_ += sp.X.size(0)
            cache = engine.build_stats_cache(self.X, all_sizes, all_starts, bank, args.topj, args.topk)
            for b in self.batches:
                b.stats_cache = cache
        self.labels = torch.tensor(labels, dtype=torch.int64).to(device)
        self.flags = [torch.empty(T, dtype=torch.uint8).pin_memory() for _ in range(3)]
        self.flag_busy = [None, None, None]
        self.side = torch.cuda.Stream(device=device)
        self.X.record_stream(self.side)
        for b in self.batches:
            for t in (b.kept, b.n_kept, b.stats, b.sel_flag, b.sel_idx, b.sel_row, b.n_sel, b.cand, b.row_off, b.x_off, b.ticket):
                if t is not None:
                    t.record_stream(self.side)
        self.pool = ThreadPoolExecutor(R)
        # ---- parameters, moments and operand images: one arena per kind, run r at r * stride
        H = HIDDEN
        n_par = H * D + H + 4 * H + 4
        self.par_stride = (n_par + 63) // 64 * 64
        f32 = dict(dtype=torch.float32, device=device)
        self.P, self.Mo, self.Vo = (torch.zeros((R, self.par_stride), **f32) for _ in range(3))
        img_b = max(lib().moc_w1_image_bytes(D, engine._lib.MOC_BF16), lib().moc_w1_image_bytes(D, engine._lib.MOC_F32))
        self.img_stride = (img_b + 255) // 256 * 256
        self.images = torch.empty((R, self.img_stride), dtype=torch.uint8, device=device)
        self.W2_alt = torch.empty((R, 4, H), **f32)
        offs = (0, H * D, H * D + H, H * D + H + 4 * H)
        shapes = ((H, D), (H,), (4, H), (4,))
        group0 = None
        for r, (model, opt) in enumerate(zip(self.models, self.optimizers)):
            meta = MetaState(model, opt)                      # (validates the pair; creates Adam's state if it is new)
            g = meta._group
            hp = (float(g["lr"]), tuple(float(b) for b in g["betas"]), float(g["eps"]), float(g["weight_decay"]), meta.c.step)
            group0 = group0 or hp
            assert hp == group0, "train_runs: the runs must share Adam's hyper-parameters and step count"
            for p, o, shp in zip(meta.params, offs, shapes):
                cnt = p.numel()
                for arena, src in ((self.P, p.data), (self.Mo, opt.state[p]["exp_avg"]), (self.Vo, opt.state[p]["exp_avg_sq"])):
                    arena[r, o:o + cnt].copy_(src.reshape(-1))
                p.data = self.P[r, o:o + cnt].view(shp)
                opt.state[p]["exp_avg"] = self.Mo[r, o:o + cnt].view(shp)
                opt.state[p]["exp_avg_sq"] = self.Vo[r, o:o + cnt].view(shp)
        self.meta = MetaState(self.models[0], self.optimizers[0])            # run 0's tensors: the base of every arena
        self.meta.c.W1_image = ptr(self.images)
        self.runs = MocRuns(n_runs=R, slide_stride=n, par_stride=self.par_stride, image_stride=self.img_stride)
        self.turn, self.ahead, self.steps_done = 0, None, [None, None]
        self.last = None
        # The runs step in lockstep inside a GROUP; several groups are independent chains on streams of their own, whose
        # latency-bound kernels interleave on the device (one group: every launch serves all R runs)
        # Shapes outside the tile-record step (wide banks): moc_train_steps_runs takes a group's runs one after the other,
        # so every run is a group of its own -- R chains of (forward, top-K, wide step) side by side, each of which keeps
        # a few dozen CUs busy (include/moc_hip.h moc_train_runs_mode)
        self.mode = int(lib().moc_train_runs_mode(C.byref(self.batches[0].c), C.byref(self.batches[0].meta_ws()[1])))
        assert self.mode != 0, "train_runs: this shape takes the three-launch step, whose scratch is one per batch -- train the runs one by one"
        G_default = (R + 7) // 8 if self.mode == 1 else R
        G = max(1, min(R, int(os.environ.get("MOC_RUNS_GROUPS", str(G_default)))))        # (measured, lockstep: chains of up to eight runs)
        per = (R + G - 1) // G
        self.groups = []
        for r0 in range(0, R, per):
            r1 = min(R, r0 + per)
            mc = type(self.meta.c).from_buffer_copy(self.meta.c)
            for name in ("W1", "b1", "W2", "b2", "m_W1", "m_b1", "m_W2", "m_b2", "v_W1", "v_b1", "v_W2", "v_b2"):
                setattr(mc, name, getattr(self.meta.c, name) + 4 * r0 * self.par_stride)
            mc.W1_image = ptr(self.images) + r0 * self.img_stride
            self.groups.append({"r0": r0, "meta": mc, "runs": MocRuns(n_runs=r1 - r0, slide_stride=n, par_stride=self.par_stride,
                                                                  image_stride=self.img_stride),
                                "w2alt": ptr(self.W2_alt) + 4 * r0 * 4 * H,
                                "stream": None if r0 == 0 else torch.cuda.Stream(device=device)})

    # ---- the masks of one pass: every run draws its own, side by side (moc_host_draw_masks releases the GIL)
    def _draw(self, buf):
        """-> (kept rows of all runs, the largest kept-row count of any slide)."""
        off_c = self.batches[0]._row_off_c

        def one(r):
            g = self.generators[r]
            st = g.get_state()
            first, rows = self.run_rows[r]
            kept = lib().moc_host_draw_masks(ptr(st), st.numel(), rows, buf.data_ptr() + first)
            if kept < 0:                                       # a generator whose state the replay does not know: torch draws
                m = torch.rand(rows, generator=g) > 0.5
                buf[first:first + rows].copy_(m)
                kept = int(m.sum())
            else:
                g.set_state(st)
            # the run's slides are slides r * n ... of the batch: its largest kept-row count, from the same thread
            mk = lib().moc_host_max_kept(buf.data_ptr(), C.c_void_p(C.addressof(off_c) + 8 * r * self.n), self.n)
            return int(kept), int(mk)
        res = list(self.pool.map(one, range(self.R)))
        return sum(k for k, _ in res), max(m for _, m in res)

    def _free_flags(self):
        for i, ev in enumerate(self.flag_busy):
            if ev is None or ev.query():
                self.flag_busy[i] = None
                return i
        self.flag_busy[0].synchronize()
        self.flag_busy[0] = None
        return 0

    def _phase_a(self, turn, head_only=False):
        """Masks + phase A of the next pass into work-array set `turn`, on the CURRENT stream (head_only: the flags and the
        kept-row lists only; `_phase_a_tail` does the rest)."""
        i = self._free_flags()
        kept, max_kept = self._draw(self.flags[i])
        batch = self.batches[turn]
        if self.upload_flags:
            # one asynchronous copy of the flags (the compaction kernel reading 4 MB of them in place over PCIe takes as
            # long as the link: 77 us at eight runs), the grids still tightened to the largest kept-row count
            batch.set_mask(self.flags[i], kept)
            batch.c.max_rows = max(1, max_kept)
        else:
            batch.use_host_mask(self.flags[i], kept, max_kept)
        if head_only:
            batch.phase_a_head(self.bank)
        else:
            batch.phase_a(self.bank)
        ev = torch.cuda.Event()
        ev.record(engine.stream_obj())
        self.flag_busy[i] = ev
        return ev

    def train_pass(self):
        """One pass (epoch) of every run: main_moc.train for each of them, in lockstep."""
        for m in self.models:
            if not m.training:
                m.train()
        use = engine.train_use_bits(self.args.discard_classifiers)
        ahead, self.ahead = self.ahead, None
        if ahead is not None:
            ahead["done"].wait(engine.stream_obj())
            self.turn = ahead["turn"]
            if ahead.get("head_only"):                        # its kept-row lists exist: score pass, selection, candidates now
                self.batches[self.turn].phase_a_tail(self.bank)
        else:
            self.turn = 1 - self.turn
            self._phase_a(self.turn)
        batch = self.batches[self.turn]
        t, ws0 = batch.meta_ws()
        batch.publish_n_sel()
        self.meta.refresh()
        main = engine.stream_obj()
        ready = None
        if len(self.groups) > 1:
            ready = torch.cuda.Event()
            ready.record(main)                               # phase A of this pass is in front of it on the main stream
        joins = []

        def group_call(grp, raw_stream):
            ws = type(ws0).from_buffer_copy(ws0)
            ws.W2_alt = grp["w2alt"]
            mc = grp["meta"]
            mc.lr, mc.beta1, mc.beta2, mc.eps, mc.weight_decay, mc.step = (self.meta.c.lr, self.meta.c.beta1, self.meta.c.beta2,
                                                                            self.meta.c.eps, self.meta.c.weight_decay, self.meta.c.step)
            return lib().moc_train_steps_runs(C.byref(batch.c), C.byref(mc), C.byref(grp["runs"]), C.byref(ws), ptr(self.labels),
                                              grp["r0"] * self.n, self.n, use, raw_stream)
        main_raw = engine._stream()
        raw_of = lambda grp: main_raw if grp["stream"] is None else C.c_void_p(grp["stream"].cuda_stream)
        for grp in self.groups:
            if grp["stream"] is not None:
                grp["stream"].wait_event(ready)
        if self.mode == 2 and len(self.groups) > 1:
            # one chain of three launches per meta-step and run: the host's launch calls are what would hold the chains apart
            # (120 steps x 3 launches x ~4 us per run and pass), so every group's pass is issued from a thread of its own (the
            # library call releases the GIL; the stream is handed over as its raw handle)
            def threaded(grp):
                torch.cuda.set_device(self.device)             # (a new thread's current device is device 0)
                return group_call(grp, raw_of(grp))
            rcs = list(self.pool.map(threaded, self.groups))
        else:
            rcs = [group_call(grp, raw_of(grp)) for grp in self.groups]
        for rc in rcs:
            check(rc, "moc_train_steps_runs")
        for grp in self.groups:
            if grp["stream"] is not None:
                ev = torch.cuda.Event()
                ev.record(grp["stream"])
                joins.append(ev)
        for ev in joins:
            main.wait_event(ev)
        for opt in self.optimizers:                           # n fused Adam steps in every optimizer's own counters
            for st in opt.state.values():
                if "step" in st:
                    st["step"] += self.n
        self.last = (batch, self.labels)
        mark = torch.cuda.Event()
        mark.record(engine.stream_obj())
        self.steps_done[self.turn] = mark
        # The NEXT pass, on the side stream, into the other set (free once the pass before this one has run): its flags and
        # kept-row lists -- the compaction kernel reads the flags over PCIe, 77 us at eight runs, which costs the steps
        # nothing -- or (MOC_RUNS_LOOKAHEAD=1) all of its phase A (measured slower: the steps crawl under a score pass)
        other = 1 - self.turn
        if self.steps_done[other] is not None:
            self.steps_done[other].synchronize()
        with torch.cuda.stream(self.side):
            done = self._phase_a(other, head_only=not self.lookahead)
        self.ahead = {"turn": other, "done": done, "head_only": not self.lookahead}

    def losses(self):
        """[R, n] losses of the last pass (device)."""
        batch, _ = self.last
        return batch.meta_ws()[0]["loss"].view(self.R, self.n)
