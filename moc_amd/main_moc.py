"""Drop-in for the hot-path callables of the reference's main_moc.py
(senet :299-312, slide_process :322-375, train :378-410, zs_evaluation :412-460,
evaluation :462-520, ablation_evaluation :523-582): same names, arguments, return
structures and quirks, computed by libmoc_hip on the MI355X.

Like the reference, train/evaluation read the classifier bank from module
globals `zeroshot_weights` / `zeroshot_weights_ext` (set them directly or through
set_classifier_bank).

How the loops map onto the GPU
  * slide_process has no trainable parameter, so for a whole loader pass its
    work (mask -> scores -> 4 selectors -> union -> candidates: "phase A") is done
    for ALL slides in a handful of batched launches before the meta-learner runs;
  * train then takes one fused Adam step per slide, in loader order, without
    ever synchronising the host (moc_train_steps); results are those of the
    reference's sequential loop because phase A does not depend on the parameters;
  * evaluation / zs_evaluation have no sequential dependence at all and are
    batched end to end; only the [n_slides, C] pooled logits come back to the
    host, where sklearn computes the AUC as in the reference.
  * the row mask is drawn on the CPU default generator, one `torch.rand(N) > 0.5`
    per slide in loader order (main_moc.py:330).  Given the same generator state at
    the first slide, the bits are the reference's.  What differs is what ELSE draws
    from that generator: the reference iterates a DataLoader(num_workers=1), and every
    DataLoader.__iter__ -- train and each evaluation pass -- first draws a 64-bit base
    seed from the same generator.  A torch DataLoader handed to train()/evaluation()
    here makes that draw too (same stream as the reference); a ResidentBags split does
    not unless built with `loader_seed_draw=True` (then every pass over it consumes the
    draw, and phase A is no longer issued a pass ahead: the generator state the next
    train() will find is not known in advance).
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from sklearn.metrics import roc_auc_score
from tqdm import tqdm

from . import engine
from .engine import Bank, MetaState, SlideBatch
from .patch_selection_classifier import (bottomk_irrel_classifier_pooling, delta_diff_classifier_pooling,
                                         delta_softmax_classifier_pooling, topj_pooling)

zeroshot_weights = None        # [D, C]      main_moc.py:161-202
zeroshot_weights_ext = None    # [D, C+4]

CONCH_TEMPERATURE = 56.3477    # main_moc.py:443, :505
# slides are processed in chunks of at most this many bag bytes per phase-A batch
MAX_BATCH_BYTES = 24 << 30


def set_classifier_bank(W: torch.Tensor, W_ext: torch.Tensor):
    global zeroshot_weights, zeroshot_weights_ext
    zeroshot_weights, zeroshot_weights_ext = W, W_ext


class _SenetFn(torch.autograd.Function):
    """`model(selected_feat)` for a caller who keeps the reference's loop body (main_moc.py:390-410) around
    slide_process: the forward is the training step's own forward kernel (moc_meta_forward over the given rows), the
    backward moc_senet_backward -- only the rows whose lambdas carry gradient (the <= K*C pooled ones) are touched."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, model):
        S, D = x.shape
        xc = x.detach().contiguous()
        batch = engine.CompactBatch(1, S, S, D, xc.dtype, 2, 3, S, 1, xc.device, X=xc)
        batch.n_sel.fill_(S)
        batch.set_layout([0, S])
        meta = MetaState(model)
        keep = any(ctx.needs_input_grad[1:5])            # (grad mode is off inside forward(): ask the context)
        engine.meta_forward(batch, meta, 0, 1, 0, keep_hidden=True)
        t, _ = batch.meta_ws()
        if keep:
            ctx.save_for_backward(xc, t["H1"], t["gates"], W2.detach())
            ctx.hold = (batch, meta)                       # (the work arrays live as long as the graph)
        return t["gates"].clone()

    @staticmethod
    def backward(ctx, g):
        xc, H1, gates, W2 = ctx.saved_tensors
        S, D = xc.shape
        dev = xc.device
        g = g.detach().to(torch.float32).contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        gW1, gb1 = torch.empty((engine.HIDDEN, D), **f32), torch.empty(engine.HIDDEN, **f32)
        gW2, gb2 = torch.empty((4, engine.HIDDEN), **f32), torch.empty(4, **f32)
        dz, dh = torch.empty((S, 4), **f32), torch.empty((S, engine.HIDDEN), **f32)
        rows, n = torch.empty(S, dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
        engine.check(engine.lib().moc_senet_backward(
            engine.ptr(xc), engine._dtype_code(xc.dtype), S, D, engine.ptr(H1), engine.ptr(gates), engine.ptr(g), engine.ptr(W2),
            engine.ptr(gW1), engine.ptr(gb1), engine.ptr(gW2), engine.ptr(gb2), engine.ptr(dz), engine.ptr(dh), engine.ptr(rows),
            engine.ptr(n), engine._stream()), "moc_senet_backward")
        return None, gW1, gb1, gW2, gb2, None


class senet(nn.Module):
    """The meta-learner; same modules / state_dict keys as main_moc.py:299-312.
    train()/evaluation() below do not go through forward(): they hand the parameter tensors to the fused kernels.
    forward() itself -- for a caller-written loop -- runs the same forward kernel and a HIP backward (_SenetFn)."""

    def __init__(self, in_dim, out_dim):
        super(senet, self).__init__()
        self.hidden_dim = 64
        self.model = nn.Sequential(
            nn.Linear(in_dim, self.hidden_dim),
            nn.ReLU(),
            nn.Linear(self.hidden_dim, out_dim),
            nn.Sigmoid()
        )

    def forward(self, x):
        lin1, lin2 = self.model[0], self.model[2]
        if not x.is_cuda:
            raise RuntimeError(f"moc_amd.senet: the meta-learner runs on the GPU only (got a {x.device} tensor); "
                               "there is no CPU fallback")
        assert x.dim() == 2 and x.size(1) == lin1.in_features, f"senet: expected [S, {lin1.in_features}] rows"
        assert lin2.out_features == 4 and lin1.in_features % 256 == 0, \
            "senet: the HIP forward takes D a multiple of 256 and four gates (main_moc.py:314)"
        if x.requires_grad:
            raise NotImplementedError("senet: no gradient w.r.t. the bag rows on the MOC path (selected_feat is data)")
        if x.dtype not in (torch.float32, torch.bfloat16, torch.float16):
            x = x.to(torch.float32)
        return _SenetFn.apply(x, lin1.weight, lin1.bias, lin2.weight, lin2.bias, self)


# --------------------------------------------------------------------------
def _bag_dtype(args, default):
    want = getattr(args, "bag_dtype", None)
    if want in (None, "keep"):
        return default
    return {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16,
            "bfloat16": torch.bfloat16, "fp16": torch.float16, "float16": torch.float16}[want]


def _pack(bags, device, dtype):
    """list of [N_i, D] tensors -> one contiguous [sum N_i, D] device array."""
    sizes = [int(b.size(0)) for b in bags]
    X = torch.empty((sum(sizes), bags[0].size(1)), dtype=dtype, device=device)
    o = 0
    for b, n in zip(bags, sizes):
        X[o:o + n].copy_(b.to(device, non_blocking=True))
        o += n
    return X, sizes


def _chunks(sizes, D, itemsize):
    out, cur, cur_bytes = [], [], 0
    for i, n in enumerate(sizes):
        nb = n * D * itemsize
        if cur and cur_bytes + nb > MAX_BATCH_BYTES:
            out.append(cur)
            cur, cur_bytes = [], 0
        cur.append(i)
        cur_bytes += nb
    if cur:
        out.append(cur)
    return out


class ResidentBags:
    """Slides kept packed in HBM across epochs with the
    loader/dataset protocol main_moc.py's loops use (iteration yields batch_size=1
    items; .dataset.real_len(), .dataset.repeat_num, len()).  train/evaluation
    recognise it and skip the per-epoch re-read + host->device copy."""

    def __init__(self, bags, labels, device, dtype=None, repeat_num=None, paths=None, loader_seed_draw=False, cache_scores=None):
        dtype = dtype or bags[0].dtype
        # opt-in: keep the per-row statistics of the split (28 B per row at two classes) from one unmasked score pass and let
        # every train pass copy its kept rows' statistics instead of reading the bags again: the same bits, phase A without
        # its HBM-bound kernel.  NOT the configuration bench.py's `value` is measured in (engine.build_stats_cache)
        self.cache_scores = (os.environ.get("MOC_CACHE_SCORES", "0") == "1") if cache_scores is None else bool(cache_scores)
        self._stats_caches = {}
        # True: every pass over this split first draws the 64-bit base seed a DataLoader.__iter__ would draw from the
        # CPU default generator (the reference's loaders do, train and evaluation alike), so a seeded run sees the
        # reference's mask stream exactly; phase A is then not issued a pass ahead (module docstring)
        self.loader_seed_draw = bool(loader_seed_draw)
        self.next_pass_len = None          # see resident_pass_done
        self.pass_after_next_len = None    # ... and the one after that, when it differs (mask draws run two passes ahead)
        self.X, self.sizes = _pack(bags, device, dtype)
        self.labels = [int(v) for v in labels]
        self.repeat_num = repeat_num
        self.paths = paths or [f"slide_{i}.h5" for i in range(len(bags))]
        self.starts = [0]
        for n in self.sizes:
            self.starts.append(self.starts[-1] + n)
        self.dataset = self
        self._plans = {}
        self._last_train_key = None
        self._order = None

    def train_plan(self, C_, Ce, topj, topk, discard):
        """Work arrays, labels and pinned mask staging for train() over the current visit order,
        built once and reused every epoch (only the mask bytes change)."""
        order = self.visit_order()                         # (a cached tuple)
        key = (order, C_, Ce, topj, topk, tuple(sorted(discard)) if discard else ())
        self._last_train_key = key
        plan = self._plans.get(key)
        if plan is None:
            if len(self._plans) > 8:
                self._plans.clear()
            sizes = [self.sizes[k] for k in order]
            T = sum(sizes)
            # two sets of work arrays: while the meta-learner steps through epoch e on one, phase A of
            # epoch e+1 (parameter-free) fills the other on a side stream (_resident_pass_setup)
            batches = [SlideBatch(self.X, sizes, C_, Ce, topj, topk, discard, mask=torch.ones(T, dtype=torch.uint8),
                                  x_starts=[self.starts[k] for k in order]) for _ in range(2)]
            if self.cache_scores:
                bank_ = _bank_for(self.X, self.X.device)
                ck = (id(bank_), C_, Ce)
                if ck not in self._stats_caches:
                    self._stats_caches.clear()
                    self._stats_caches[ck] = (engine.build_stats_cache(self.X, self.sizes, self.starts[:-1], bank_, topj, topk), bank_)
                for b in batches:
                    b.stats_cache = self._stats_caches[ck][0]
            if PREFETCH_PHASE_A and Ce <= 16 and not self.loader_seed_draw and not self.cache_scores:     # (loader_seed_draw: phase A always runs in line)
                # phase A runs beside the meta-steps of the pass before: leave them CUs.  (Banks of one n-tile only: wider
                # ones run ONE score workgroup per CU -- thirty classes with 64 CUs left free: score pass 242 -> 447 us,
                # 18.9 -> 18.8 k meta-steps/s.)
                for b in batches:
                    b.reserve_cus()
            lab = torch.tensor([self.labels[k] for k in order], dtype=torch.int64).to(self.X.device)
            stage = [torch.empty(T, dtype=torch.uint8).pin_memory() for _ in range(2)]     # (only when torch itself must draw)
            drawer = engine.MaskDrawer(T, batches[0]._row_off_c, len(sizes))
            side = torch.cuda.Stream(device=self.X.device)
            # the side stream reads X and writes the work arrays: tell the caching allocator, so that memory freed
            # while a pass-ahead phase A is still in flight is not handed to someone else under it
            self.X.record_stream(side)
            for b in batches:
                for t in (b.kept, b.n_kept, b.stats, b.sel_flag, b.sel_idx, b.sel_row, b.n_sel, b.cand, b.row_off, b.x_off, b.ticket):
                    if t is not None:
                        t.record_stream(side)
            plan = self._plans[key] = {"batch": batches[0], "batches": batches, "labels": lab, "stage": stage,
                                       "stage_free": [None, None], "turn": 0, "ahead": None, "side": side, "drawer": drawer}
        return plan

    def eval_plan(self, C_, Ce, topj, topk, discard):
        """Work arrays + labels for an un-masked pass over the current visit order, reused across calls."""
        order = tuple(self.visit_order())
        key = ("eval", order, C_, Ce, topj, topk, tuple(sorted(discard or ())))
        plan = self._plans.get(key)
        if plan is None:
            if len(self._plans) > 8:
                self._plans.clear()
            sizes = [self.sizes[k] for k in order]
            batch = SlideBatch(self.X, sizes, C_, Ce, topj, topk, discard, x_starts=[self.starts[k] for k in order])
            lab = torch.tensor([self.labels[k] for k in order], dtype=torch.int64).to(self.X.device)
            plan = self._plans[key] = {"batch": batch, "labels": lab, "label_list": [self.labels[k] for k in order]}
        return plan

    def real_len(self):
        return len(self.sizes)

    def __len__(self):
        return self.repeat_num if self.repeat_num else len(self.sizes)

    def visit_order(self):
        """Slide of each visit (repeat_num visits wrap around the real slides, dataset_generic.py:380-393); the
        tuple is cached per length: plans are keyed by it, once per pass."""
        n = len(self)
        if self._order is None or len(self._order) != n:
            self._order = tuple(i % len(self.sizes) for i in range(n))
        return self._order

    def __iter__(self):
        for k in self.visit_order():
            x = self.X[self.starts[k]:self.starts[k + 1]]
            yield (x.unsqueeze(0), torch.tensor([self.labels[k]]),
                   torch.zeros(1, x.size(0), 2, dtype=torch.int64), [self.paths[k]])


def _collect(loader, device, args):
    """One pass over a loader -> (X, sizes, x_starts|None, labels)."""
    if isinstance(loader, ResidentBags):
        order = loader.visit_order()
        sizes = [loader.sizes[k] for k in order]
        return loader.X, sizes, [loader.starts[k] for k in order], [loader.labels[k] for k in order]
    bags, labels = [], []
    for data in tqdm(loader, bar_format="{l_bar}{bar:10}{r_bar}", disable=args.disable_tqdm):
        feats, lbl, coords, full_path = data
        bags.append(feats.squeeze(0))
        labels.append(int(lbl.reshape(-1)[0]))
    dtype = _bag_dtype(args, bags[0].dtype if bags[0].dtype in (torch.float32, torch.bfloat16, torch.float16) else torch.float32)
    X, sizes = _pack(bags, device, dtype)
    return X, sizes, None, labels


def _sub_batch(X, sizes, x_starts, ids, C_, Ce, topj, topk, discard, masks=None):
    """SlideBatch over slides `ids` of a collected pass (views, no copies)."""
    if x_starts is None:
        starts, o = [], 0
        for n in sizes:
            starts.append(o)
            o += n
    else:
        starts = x_starts
    m = torch.cat([masks[i] for i in ids]) if masks is not None else None
    return SlideBatch(X, [sizes[i] for i in ids], C_, Ce, topj, topk, discard, mask=m,
                      x_starts=[starts[i] for i in ids])


def _bank_for(X, device, fg_from_ext=False):
    assert zeroshot_weights is not None and zeroshot_weights_ext is not None, \
        "set moc_amd.main_moc.zeroshot_weights / zeroshot_weights_ext first (set_classifier_bank)"
    return Bank.get(zeroshot_weights, zeroshot_weights_ext, X.dtype, device, fg_from_ext)


# --------------------------------------------------------------------------
def slide_process(feat, zeroshot_weights, zeroshot_weights_ext,
                  n_classes, topj=10, random_mask=False,
                  discard_classifiers=[]) -> dict:
    """main_moc.py:322-375.  `selected_index` indexes the MASKED rows, ascending."""
    device = zeroshot_weights.device
    if device.type != "cuda":
        raise RuntimeError("moc_amd.slide_process: the classifier bank must live on the GPU (no CPU fallback)")
    feat = feat.to(device)
    if feat.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        feat = feat.to(torch.float32)
    feat = feat.contiguous()
    N = feat.size(0)
    mask = None
    if random_mask:
        mask, _ = engine.draw_row_masks(N)               # == torch.rand(N) > 0.5 on the CPU default generator
    C_, Ce = zeroshot_weights.size(1), zeroshot_weights_ext.size(1)
    assert C_ == n_classes, "n_classes must equal zeroshot_weights.size(1)"
    batch = SlideBatch(feat, [N], C_, Ce, topj, 1, discard_classifiers, mask=mask)
    bank = Bank.get(zeroshot_weights, zeroshot_weights_ext, feat.dtype, device)
    batch.scores(bank)
    batch.select()
    sel_feat = batch.gather_candidates(with_feat=True)
    S = int(batch.n_sel.item())
    cand = batch.cand[:, :S]
    return {
        "selected_index": batch.sel_idx[:S].tolist(),
        "selected_feat": sel_feat[:S],
        "logits_top_classifier": cand[:C_].t().contiguous(),
        "logits_delta_softmax_classifier": cand[C_:2 * C_].t().contiguous(),
        "logits_delta_diff_classifier": cand[2 * C_].unsqueeze(1).expand(S, C_).contiguous(),
        "logits_bottomk_irrel_classifier": cand[2 * C_ + 1].unsqueeze(1).expand(S, C_).contiguous(),
    }


PREFETCH_PHASE_A = os.environ.get("MOC_PREFETCH_PHASE_A", "1") != "0"
MASK_AHEAD = os.environ.get("MOC_MASK_AHEAD", "1") != "0"   # keep flags drawn a pass ahead on a helper thread (engine.MaskDrawer)
_TRACE = os.environ.get("MOC_BENCH_TRACE") == "1"          # host time of train()'s four parts on stderr (diagnostic)


def _issue_phase_a(plan, turn, bank, rng_before, host_wait=None, chain_plan=None):
    """Draw the masks that follow generator state `rng_before` (main_moc.py:330: same stream of bits),
    upload them and run phase A into work-array set `turn` on the CURRENT stream.  `chain_plan`: the plan of the pass
    AFTER the one these masks are for, when it visits other rows (its drawer then starts on the flags that follow).
    -> generator state after the draws (None: the state's layout is unknown, torch drew and advanced itself)."""
    batch = plan["batches"][turn]
    same = chain_plan is None or chain_plan is plan
    drawn = plan["drawer"].take(rng_before, chain=same) if MASK_AHEAD else None    # usually ready: drawn a pass ahead on the helper thread
    if drawn is not None:
        stage, kept, max_kept, rng_after, buf = drawn
        if not same:
            chain_plan["drawer"].prefetch(rng_after)
    else:                                               # generator layout unknown to the replay: torch draws, in line
        if plan["stage_free"][turn] is not None:
            plan["stage_free"][turn].synchronize()      # the phase A that read that pinned buffer has run
        stage, kept, rng_after = engine.draw_row_masks_from(rng_before, batch.total, plan["stage"][turn])
        max_kept, buf = None, None
    if host_wait is not None:
        host_wait.synchronize()                         # (after the draw: the host works while it would wait)
    batch.use_host_mask(stage, kept, max_kept)          # read in place by the compaction kernel: no upload
    batch.phase_a(plan["bank"])
    ev = torch.cuda.Event()
    ev.record(engine.stream_obj())
    if buf is not None:
        plan["drawer"].attach(buf, ev)                  # ... so the buffer is free again when phase A has run
    else:
        plan["stage_free"][turn] = ev
    return rng_after


def _resident_pass_setup(res, device, args, defer_rng=False):
    """Train pass over a resident split: nothing is copied or allocated per epoch except the new mask
    bytes.  Returns (batch with this epoch's phase A issued, device labels, bank).

    Phase A has no trainable parameter, so the pass for epoch e+1 is issued one call AHEAD, on a side
    stream, while the sequential meta-steps of epoch e occupy a fraction of the GPU (resident_pass_done).
    It is speculative only in what the CPU generator will hold when train() is called again: the masks
    are drawn from a COPY of the state, torch's generator is left where epoch e put it, and the next call
    adopts the work only if it finds exactly that state (and the same bank); otherwise it is dropped and
    phase A runs here, as before."""
    bank = _bank_for(res.X, device)
    assert bank.C == args.n_classes
    plan = res.train_plan(bank.C, bank.Ce, args.topj, args.topk, args.discard_classifiers)
    plan["bank"] = bank
    lab = plan["labels"]
    now = torch.get_rng_state()
    ahead, plan["ahead"] = plan["ahead"], None
    plan["rng_after"] = None
    if ahead is not None and ahead["bank"] is bank and ahead["after"] is not None and torch.equal(ahead["before"], now):
        # the draws happened: the generator is put where they leave it -- by the caller, AFTER the pass's launches are
        # issued (train(): 3 us that the first launch of the pass need not wait for)
        if defer_rng:
            plan["rng_after"] = ahead["after"]
        else:
            torch.set_rng_state(ahead["after"])
        ahead["done"].wait(engine.stream_obj())         # (the current stream waits; its Stream object is a cached one)
        plan["turn"] = ahead["turn"]
    else:
        if ahead is not None:
            ahead["done"].synchronize()                 # dropped work still owns that set of arrays until it ends
        plan["turn"] = 1 - plan["turn"]
        after = _issue_phase_a(plan, plan["turn"], bank, now)
        if after is not None:
            torch.set_rng_state(after)
    plan["batch"] = plan["batches"][plan["turn"]]
    return plan["batch"], lab, bank


def _next_plan(res, plan, bank, args, which="next_pass_len"):
    """The plan of the pass that follows this one: the same visits (the reference's epoch loop), or what the caller
    announced in `res.next_pass_len` (0: no pass follows -> None).  which="pass_after_next_len": the pass after that."""
    hint = getattr(res, which, None)
    if hint == 0:
        return None
    if hint is None and which != "next_pass_len":
        hint = getattr(res, "next_pass_len", None)       # (not announced: as the next one)
        if hint == 0:
            return None
    if hint is None or hint == len(res):
        return plan
    keep, res.repeat_num = res.repeat_num, (hint if hint != res.real_len() else None)
    try:
        return res.train_plan(bank.C, bank.Ce, args.topj, args.topk, args.discard_classifiers)
    finally:
        res.repeat_num = keep


def resident_pass_done(res, device, args):
    """Call after the pass's meta-steps are issued: starts phase A of the NEXT pass on the side stream.

    The next pass is taken to visit what this one visited (the reference's epoch loop, main_moc.py:606-628).  A
    caller that knows better says so in `res.next_pass_len`: a number of visits (bench.py's partial passes), or 0
    for "no pass follows" (nothing is issued ahead)."""
    if not PREFETCH_PHASE_A or res.loader_seed_draw:
        return
    bank = _bank_for(res.X, device)
    plan = res.train_plan(bank.C, bank.Ce, args.topj, args.topk, args.discard_classifiers)
    # a set of arrays is free again when the meta-steps that read it have run: mark this pass's end on `main`;
    # the other set was last read by the PREVIOUS pass's meta-steps (their mark was left one call ago), so the
    # side stream waits for that one only and phase A of the next pass overlaps THIS pass's meta-steps
    mark = torch.cuda.Event()
    mark.record(engine.stream_obj())                    # (on the current stream -- `main` below)
    plan.setdefault("steps_done", [None, None])[plan["turn"]] = mark
    nplan = _next_plan(res, plan, bank, args)
    if nplan is None:
        return
    if nplan is not plan and nplan.get("ahead") is not None:     # (an unadopted speculation of that plan still owns a set)
        nplan["ahead"]["done"].synchronize()
        nplan["ahead"] = None
    nplan["bank"] = bank
    other, side = 1 - nplan["turn"], nplan["side"]
    before = torch.get_rng_state()
    # The HOST waits for that mark, not the side stream: a stream-side wait on an event of the main stream
    # cost ~45 us of GPU time per pass here (scripts/diag_overlap.py: 0.76 ms per epoch against 0.72), and
    # the host is a pass ahead with this pass's launches already queued, so its wait starves nothing.
    chain = _next_plan(res, plan, bank, args, which="pass_after_next_len")
    with torch.cuda.stream(side):
        after = _issue_phase_a(nplan, other, bank, before, host_wait=nplan.setdefault("steps_done", [None, None])[other],
                               chain_plan=chain)
        done = torch.cuda.Event()
        done.record(side)
    if after is None:                                   # torch drew for us and moved its generator: undo, no speculation
        torch.set_rng_state(before)
    nplan["ahead"] = {"turn": other, "before": before, "after": after, "done": done, "bank": bank}


def train(model, train_loader, optimizer, device, args):
    """main_moc.py:378-410: one Adam step per slide, in loader order."""
    if not model.training:
        model.train()
    use = engine.train_use_bits(args.discard_classifiers)
    if isinstance(train_loader, ResidentBags):
        _loader_seed_draw(train_loader)
        if _TRACE:
            import time
            t0 = time.perf_counter()
        batch, lab, bank = _resident_pass_setup(train_loader, device, args, defer_rng=True)      # phase A issued (or adopted)
        if _TRACE:
            t1 = time.perf_counter()
        meta = MetaState.cached(model, optimizer)
        if _TRACE:
            t2 = time.perf_counter()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        engine.train_steps(batch, meta, lab, 0, batch.n_slides, use)
        plan_ = train_loader._plans.get(train_loader._last_train_key)
        if plan_ is not None and plan_.get("rng_after") is not None:
            torch.set_rng_state(plan_["rng_after"])     # (see _resident_pass_setup)
            plan_["rng_after"] = None
        if _TRACE:
            ev1.record()
            train.trace_events = getattr(train, "trace_events", [])[-200:] + [(batch.n_slides, ev0, ev1)]
            t3 = time.perf_counter()
        resident_pass_done(train_loader, device, args)
        if _TRACE:
            t4 = time.perf_counter()
            # (kept, not printed: a print inside a timed region is 30 us of it; bench.py prints the list afterwards)
            train.trace_host = getattr(train, "trace_host", [])[-200:] + [
                f"setup {(t1 - t0) * 1e6:.0f}  meta {(t2 - t1) * 1e6:.0f}  launches {(t3 - t2) * 1e6:.0f}  pass_done {(t4 - t3) * 1e6:.0f} us"]
        train.last = (batch, lab)
        return
    X, sizes, x_starts, labels = _collect(train_loader, device, args)
    mask_all, _ = engine.draw_row_masks(sum(sizes))      # main_moc.py:330, one draw per slide, in order
    masks, o = [], 0
    for n in sizes:
        masks.append(mask_all[o:o + n])
        o += n
    meta = MetaState(model, optimizer)
    bank = _bank_for(X, device)
    C_, Ce = bank.C, bank.Ce
    assert C_ == args.n_classes
    for ids in _chunks(sizes, X.size(1), X.element_size()):
        batch = _sub_batch(X, sizes, x_starts, ids, C_, Ce, args.topj, args.topk, args.discard_classifiers, masks)
        lab = torch.tensor([labels[i] for i in ids], dtype=torch.int64).to(device, non_blocking=True)
        batch.phase_a(bank)
        engine.train_steps(batch, meta, lab, 0, len(ids), use)
        train.last = (batch, lab)    # keeps the buffers alive until the stream has drained; also for tests


_run_sets = {}


def train_runs(models, train_loaders, optimizers, device, args, generators=None):
    """main_moc.py:378-410 for several independent runs at once -- what scripts/moc_train.sh:11-31 starts as one process per
    (fold, shot): `train(models[r], train_loaders[r], optimizers[r], device, args)` for every r, stepped in lockstep by one
    launch pair per meta-step (moc_amd.runs.TrainRuns; include/moc_hip.h moc_train_steps_runs).  Per run bit-identical to
    `train` from the same mask stream; run r's masks come from `generators[r]` (a private CPU torch.Generator each; by
    default seeded from the default generator on the first call).  The loaders are resident splits (ResidentBags) of equal
    length.  -> the TrainRuns object (kept for the next pass: call again with the same lists)."""
    from .runs import TrainRuns
    key = (tuple(id(m) for m in models), tuple(id(o) for o in optimizers), tuple(id(l) for l in train_loaders),
           args.topj, args.topk, tuple(sorted(args.discard_classifiers or ())), tuple(len(l) for l in train_loaders))
    ent = _run_sets.get(key)
    if ent is None:
        if len(_run_sets) > 2:
            _run_sets.clear()
        ent = _run_sets[key] = (TrainRuns(models, optimizers, train_loaders, device, args, generators=generators),
                                list(models), list(optimizers), list(train_loaders))
    rs = ent[0]
    rs.train_pass()
    train_runs.last = rs
    return rs


def _binary_auc(pos: np.ndarray, score: np.ndarray) -> float:
    """Area under the ROC curve of `score` for the boolean labels `pos`: the Mann-Whitney statistic with
    mid-ranks for ties, which is what the trapezoid over sklearn's roc_curve thresholds adds up to."""
    order = np.argsort(score, kind="mergesort")
    s = score[order]
    # mid-rank of every tie group: first occurrence index + (group size - 1) / 2, 1-based
    # (plain array writes, not np.r_: its index-trick machinery is 25 us a call, a third of this function)
    start = np.empty(s.size, dtype=bool)
    start[0] = True
    np.not_equal(s[1:], s[:-1], out=start[1:])
    first = np.flatnonzero(start)
    size = np.empty(first.size, dtype=np.int64)
    size[:-1] = first[1:] - first[:-1]
    size[-1] = s.size - first[-1]
    mid = np.repeat(first + (size - 1) / 2.0 + 1.0, size)
    n_pos = int(pos.sum())
    n_neg = pos.size - n_pos
    return float((mid[pos[order]].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def _auc(y_true: np.ndarray, y_score: np.ndarray) -> float:
    """roc_auc_score(y, p[:, 1]) for two classes, roc_auc_score(y, p, multi_class='ovo', average='macro')
    otherwise (main_moc.py:507-514), computed directly: sklearn's input validation and per-pair Python
    machinery cost 0.4 ms per binary call and 0.1-0.6 s per call at 30-64 classes -- more than the GPU
    side of the whole evaluation.  Anything sklearn would reject (a class missing from y_true, ...) is
    handed to sklearn so that the same exception comes out."""
    classes = np.unique(y_true)
    if y_score.ndim == 1:
        if classes.size != 2:
            return roc_auc_score(y_true, y_score)
        return _binary_auc(y_true == classes[1], y_score)
    if classes.size != y_score.shape[1] or classes.size < 2 or not np.array_equal(classes, np.arange(classes.size)):
        return roc_auc_score(y_true, y_score, multi_class="ovo", average="macro")
    n, Cn = y_score.shape
    if n <= 4 * Cn * (Cn - 1):        # n^2 comparisons against C (C - 1) sorts of ~2n/C values: many classes, few slides
        # every pair at once.  G[i, j] = [p[i, y_i] > p[j, y_i]] + [==] / 2; U[a, b] = sum of G over i in a, j in b
        # = n_a n_b x the AUC of class a against class b on column a (Mann-Whitney, ties at one half).  Sums of
        # multiples of 1/2: exact.  (np.bincount, not a matrix product: BLAS starts a thread per visible core.)
        own = y_score[np.arange(n), y_true]
        cols = y_score[:, y_true].T
        G = (own[:, None] > cols) + 0.5 * (own[:, None] == cols)
        pair = (y_true[:, None] * Cn + y_true[None, :]).ravel()
        U = np.bincount(pair, weights=G.ravel(), minlength=Cn * Cn).reshape(Cn, Cn)
        cnt = np.bincount(y_true, minlength=Cn).astype(np.float64)
        A = U / np.outer(cnt, cnt)
        return float((A.sum() - np.trace(A)) / (Cn * (Cn - 1)))
    masks = [y_true == c for c in classes]
    total, n_pairs = 0.0, 0
    for a in range(classes.size):
        for b in range(a + 1, classes.size):
            ab = masks[a] | masks[b]
            a_true = masks[a][ab]
            total += (_binary_auc(a_true, y_score[ab, a]) + _binary_auc(~a_true, y_score[ab, b])) / 2.0
            n_pairs += 1
    return total / n_pairs


def _metrics(pooled_cpu, labels, losses, n_div, real_len, args):
    """The shared tail of the three evaluation loops (main_moc.py:439-460, :501-520)."""
    test_loss = 0
    for v in losses:
        test_loss += v
    test_loss /= n_div
    lbl_all = torch.tensor(labels, dtype=torch.int64)
    correct = int((pooled_cpu.argmax(dim=1) == lbl_all).sum().item())
    if args.pretrain == 'conch':
        temperature = CONCH_TEMPERATURE
    else:
        raise NotImplementedError
    probs = F.softmax(pooled_cpu * temperature, dim=1)
    class_probs = probs[:, 1] if probs.shape[1] == 2 else probs
    auc = _auc(lbl_all.numpy(), class_probs.numpy())
    return {"loss": test_loss, "acc": correct / real_len, "auc": auc}


def _eval_batches(loader, device, args, mode):
    """(batch, device labels, label list) per chunk of an evaluation pass."""
    discard = args.discard_classifiers if mode == "eval" else []
    if isinstance(loader, ResidentBags) and loader.X.numel() * loader.X.element_size() <= MAX_BATCH_BYTES:
        bank = _bank_for(loader.X, device, fg_from_ext=(mode == "zs_bottomk"))
        plan = loader.eval_plan(bank.C, bank.Ce, args.topj, args.topk, discard)
        return bank, [(plan["batch"], plan["labels"], plan["label_list"])]
    X, sizes, x_starts, labels = _collect(loader, device, args)
    bank = _bank_for(X, device, fg_from_ext=(mode == "zs_bottomk"))
    out = []
    for ids in _chunks(sizes, X.size(1), X.element_size()):
        batch = _sub_batch(X, sizes, x_starts, ids, bank.C, bank.Ce, args.topj, args.topk, discard)
        lab_list = [labels[i] for i in ids]
        out.append((batch, torch.tensor(lab_list, dtype=torch.int64).to(device, non_blocking=True), lab_list))
    return bank, out


def _loader_seed_draw(loader):
    """What `DataLoader.__iter__` takes from the CPU default generator before the first item (its base seed)."""
    if isinstance(loader, ResidentBags) and loader.loader_seed_draw:
        torch.empty((), dtype=torch.int64).random_()


_ext_as_bank = {}


def _ext_logits_bank(X, device):
    """A bank whose "foreground" columns are ALL of zeroshot_weights_ext (plus one zero background column, which the
    bank layout needs): after batch.scores(bank), stats[:Ce] is `feats @ zeroshot_weights_ext` (main_moc.py:428)."""
    We = zeroshot_weights_ext
    key = (We.data_ptr(), We._version, tuple(We.shape))
    pad = _ext_as_bank.get(key)
    if pad is None:
        _ext_as_bank.clear()
        pad = _ext_as_bank[key] = torch.cat([We, torch.zeros_like(We[:, :1])], 1).contiguous()
    return Bank.get(We, pad, X.dtype, device)


def _eval_pass_custom(loader, device, args, pooling_func):
    """zs_evaluation with a pooling function that is not one of this package's four: the reference calls anything
    else as `pooling_func(feats @ zeroshot_weights_ext, [args.topk], coords_list=args.n_classes)[1][args.topk]`
    (main_moc.py:431-432).  The logits are formed by the score kernel, slide-batched; the callable then runs per
    slide on the device tensor, as it would in the reference."""
    _loader_seed_draw(loader)
    X, sizes, x_starts, labels = _collect(loader, device, args)
    bank = _ext_logits_bank(X, device)
    Ce = bank.C
    outs = []
    for ids in _chunks(sizes, X.size(1), X.element_size()):
        batch = _sub_batch(X, sizes, x_starts, ids, bank.C, bank.Ce, args.topj, args.topk, [])
        batch.scores(bank)
        tensors, _ = batch.meta_ws()
        rows = []
        for b in range(len(ids)):
            o, n = batch.row_off_host[b], batch.sizes[b]
            logits_ext = batch.stats[:Ce, o:o + n].t().contiguous()
            rows.append(pooling_func(logits_ext, [args.topk], coords_list=args.n_classes)[1][args.topk].reshape(1, -1))
        pooled = torch.cat(rows, 0).to(torch.float32).contiguous()
        lab = torch.tensor([labels[i] for i in ids], dtype=torch.int64).to(device)
        loss = torch.empty(len(ids), dtype=torch.float32, device=device)
        pred = torch.empty(len(ids), dtype=torch.int32, device=device)
        engine.check(engine.lib().moc_ce_loss(engine.ptr(pooled), engine.ptr(lab), len(ids), pooled.size(1),
                                              engine.ptr(loss), engine.ptr(pred), engine._stream()), "moc_ce_loss")
        outs.append(torch.cat([pooled, loss.unsqueeze(1)], 1).cpu())
    allv = torch.cat(outs, 0)
    return allv[:, :-1].contiguous(), labels, allv[:, -1].tolist()


def _eval_pass(loader, device, args, mode, model=None, pooling_func=None):
    _loader_seed_draw(loader)
    bank, batches = _eval_batches(loader, device, args, mode)
    C_ = bank.C
    meta = MetaState(model) if model is not None else None
    outs, labels = [], []
    for batch, lab, lab_list in batches:
        n = batch.n_slides
        tensors, _ = batch.meta_ws()
        if mode.startswith("zs"):
            batch.scores(bank)
            st = batch.stats
            kind = mode[3:]
            if kind == "topj":
                keys, vals, small, shared = st[:C_], st[:C_], False, False
            elif kind == "delta_softmax":
                keys, vals, small, shared = st[C_:2 * C_], st[:C_], False, False
            elif kind == "delta_diff":
                keys, vals, small, shared = st[2 * C_:2 * C_ + 1], st[:C_], False, True
            else:   # bottomk: K rows of smallest background mass, mean of their foreground logits
                keys, vals, small, shared = st[2 * C_ + 1:2 * C_ + 2], st[:C_], True, True
            p = engine.topk_mean(keys, vals, args.topk, smallest=small, key_shared=shared, seg_off=batch.row_off)
            tensors["pooled"].copy_(p)
            engine.loss_only(batch, lab, 0, n)
        else:
            batch.phase_a(bank, for_eval=(mode == "eval"))
            if mode == "eval":
                engine.meta_forward(batch, meta, 0, n, engine.eval_use_bits(args.discard_classifiers), keep_hidden=False)
            else:
                engine.mix_fixed(batch, 0, n, args.ablation_study)
            engine.pool_loss(batch, lab, 0, n)
        # one device->host transfer per chunk: [n, C] pooled logits | loss
        outs.append(torch.cat([tensors["pooled"], tensors["loss"].unsqueeze(1)], 1).cpu())
        labels.extend(lab_list)
    allv = torch.cat(outs, 0)
    return allv[:, :-1].contiguous(), labels, allv[:, -1].tolist()


def zs_evaluation(loader, device, args, pooling_func=topj_pooling):
    """main_moc.py:412-460."""
    with torch.no_grad():
        real_len = loader.dataset.real_len()
        set_len = loader.dataset.repeat_num
        loader.dataset.repeat_num = real_len
        kinds = {topj_pooling: "topj", delta_softmax_classifier_pooling: "delta_softmax",
                 delta_diff_classifier_pooling: "delta_diff", bottomk_irrel_classifier_pooling: "bottomk"}
        try:
            if pooling_func in kinds:
                pooled, labels, losses = _eval_pass(loader, device, args, "zs_" + kinds[pooling_func])
            else:           # any other callable, as the reference accepts (main_moc.py:429-432)
                pooled, labels, losses = _eval_pass_custom(loader, device, args, pooling_func)
        finally:
            loader.dataset.repeat_num = set_len
    return _metrics(pooled, labels, losses, len(loader.dataset), real_len, args)


def evaluation(model, loader, device, args):
    """main_moc.py:462-520 (incl. the eval-side mix quirk, see engine.eval_use_bits)."""
    if model.training:                   # (nn.Module.eval() walks the module tree: 20 us of an 1 ms pass)
        model.eval()
    with torch.no_grad():
        real_len = loader.dataset.real_len()
        set_len = len(loader.dataset)
        loader.dataset.repeat_num = real_len
        try:
            pooled, labels, losses = _eval_pass(loader, device, args, "eval", model=model)
        finally:
            loader.dataset.repeat_num = set_len
    return _metrics(pooled, labels, losses, len(loader.dataset), real_len, args)


def ablation_evaluation(loader, device, args):
    """main_moc.py:523-582, args.ablation_study in {avg, sum, max}."""
    real_len = loader.dataset.real_len()
    set_len = len(loader.dataset)
    loader.dataset.repeat_num = real_len
    try:
        with torch.no_grad():
            pooled, labels, losses = _eval_pass(loader, device, args, "ablation")
    finally:
        loader.dataset.repeat_num = set_len
    return _metrics(pooled, labels, losses, len(loader.dataset), real_len, args)
