#!/usr/bin/env python3
"""Headline benchmark: meta-steps/sec of the MOC train loop (+ slides/sec of
evaluation) on synthetic NSCLC 16-shot bags (32 slides x 15k x 512), BASELINE.json
configs[1].

    python bench.py --gpus 1 --steps 1600 --warmup 160
    python bench.py --gpus N --steps K --warmup W            # starts its own N ranks (one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one meta-step = one slide through mask -> scores -> 4 selectors ->
union -> meta-learner -> top-K pooling -> CE -> backward -> Adam (main_moc.py:380-410),
everything the reference's train() does per slide, bags already resident in HBM.

At N > 1 the default (`--train-mode runs`, round 4) is what shards naturally in TRAINING: whole runs.  Every GPU trains
an independent 16-shot run of its own (the reference's scripts/moc_train.sh starts one process per fold x shot), nothing
is exchanged, every run is the reference's one-Adam-step-per-slide trajectory bit for bit; a step is one slide, `value`
counts the slides all ranks consume per second, scaling "weak".  `--train-mode seq` keeps ONE run over the GPUs, exact-
sequential (moc_amd.dist.train_seq, SURVEY.md section 8e mode 1: bags and phase A sharded, compact results all-gathered,
the recurrence on every rank -- bit-identical to one GPU, and a latency chain that no second GPU shortens); `--seq-extra`
adds it to a default run under `exact_sequential`.  Minibatch data parallelism (`--train-mode dp`: one slide per rank per
synchronous step, the meta-gradient summed over the ranks, one Adam step per N slides) is fast but changes the trajectory
-- measured AUC deviations of 0.01-0.2 from the sequential run (profiles/round2_dp_auc_study.jsonl), outside the +-0.002
bar -- so it is an opt-in extension (`--dp-extra`: the `minibatch_dp` block).

Prints ONE JSON line on rank 0.  Extra keys: steady_state (>= 50 whole epochs of the same model in the
same run: what a training run of many epochs sees, whatever --steps was), roofline (dominant kernel =
the score pass, timed live with events on the launch stream), cpu_baseline (the oracle's train loop on
this box's host cores, bounded sample), eval slides/sec.
"""
import argparse
import gc
import hashlib
import json
import os
import socket
import subprocess
import sys
import time
import types

# dmabuf IPC for the peer-mapped exchange buffers of `--dp-exchange auto` (hipIpcGetMemHandle): must be in the environment
# before the HIP runtime starts, so it is decided from argv here.  The default exchange is RCCL through torch.distributed,
# which needs nothing from this script: the variable is then left exactly as the launcher's environment has it.
if "auto" in [a_ for i_, a_ in enumerate(sys.argv) if i_ and sys.argv[i_ - 1] == "--dp-exchange"] or "--dp-exchange=auto" in sys.argv:
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRACE = os.environ.get("MOC_BENCH_TRACE") == "1"      # host time per pass on stderr (diagnostic)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def granted_cpus():
    """CPUs this process may actually use: affinity capped by the cgroup quota.  torch sizes its
    OpenMP pool from the machine (256 on the GPU box) while the container is granted 16; the
    spinning surplus threads burn the CFS quota and the whole process is throttled for ~90 ms
    at a time -- measured, see DESIGN.md."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(q) // int(p)))
    except (OSError, ValueError):
        pass
    return n


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1600)
    ap.add_argument("--warmup", type=int, default=160)
    ap.add_argument("--dtype", default="fp32", choices=["bf16", "fp16", "fp32"],
                    help="bag STORAGE type; arithmetic is always fp32.  fp32 (default) is the storage of the reference's h5 "
                         "embeddings and the one pinned end to end to the reference's main(); 16-bit storage moves the trained "
                         "AUC by 0.001-0.03 (profiles/round3_storage_fidelity.jsonl), so it is reported as an extra block")
    ap.add_argument("--no-16bit-extra", action="store_true", help="skip the bf16_storage extra block of a default fp32 run")
    ap.add_argument("--slides", type=int, default=32, help="train slides per epoch (NSCLC 16-shot: 32)")
    ap.add_argument("--patches", type=int, default=15000)
    ap.add_argument("--classes", type=int, default=2)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--topj", type=int, default=400)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--eval-slides", type=int, default=202, help="NSCLC-16 fold-0 test split size")
    ap.add_argument("--lognormal", action="store_true", help="log-normal bag sizes around --patches")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-eval", action="store_true")
    ap.add_argument("--no-steady", action="store_true", help="skip the steady_state block")
    ap.add_argument("--steady-epochs", type=int, default=3200,
                    help="whole epochs of the same model timed right after the --steps region (default: 3,200 epochs = 102,400 "
                         "meta-steps, about 2 s of GPU time: long enough for an outside sampler of GPU activity to see it)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--train-mode", default="runs", choices=["runs", "seq", "dp"],
                    help="N > 1: runs = one independent run per GPU, nothing exchanged (default: what shards naturally in "
                         "training is whole runs -- the reference's scripts/moc_train.sh; every run the reference's trajectory); "
                         "seq = ONE run over the GPUs, exact-sequential (one Adam step per slide, bit-identical to one GPU: a "
                         "latency chain that no second GPU shortens); dp = minibatch data parallelism (one step per N slides: "
                         "changes the trajectory)")
    ap.add_argument("--seq-extra", action="store_true",
                    help="N > 1, runs: also measure the exact-sequential mode of ONE run over the GPUs (`exact_sequential` block; "
                         "its RCCL all-gathers have never run on two distinct devices on this pool, hence opt-in)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="--train-mode dp: strong = the same --slides slides sharded over the ranks (the BASELINE metric); "
                         "weak = every rank its own --slides slides")
    ap.add_argument("--dp-extra", action="store_true",
                    help="N > 1, seq: also measure the opt-in minibatch data-parallel mode (`minibatch_dp` block).  Off by default: its "
                         "gradient exchange (a private RCCL communicator, or peer-mapped buffers) has never run on two distinct "
                         "devices -- this pool has one GPU per box -- and a hang there would cost the line its `value`")
    ap.add_argument("--no-replicas", action="store_true", help="N > 1: skip the `replicas` block (N independent runs, no collective)")
    ap.add_argument("--dp-exchange", default="rccl", choices=["rccl", "auto"],
                    help="minibatch data parallelism: rccl = ONE RCCL all-reduce of the flat meta-gradient per step (default: "
                         "the collective is the path that needs no peer mapping); auto = the sum inside the step kernel over "
                         "peer-mapped xGMI buffers when its two set-up self-checks pass, else the collective")
    ap.add_argument("--packed-runs", type=int, default=4,
                    help="N = 1: also measure this many independent training runs side by side on the ONE GPU (separate processes, "
                         "as scripts/moc_train.sh of the reference packs five folds onto a GPU); 0 = skip")
    ap.add_argument("--no-cached-extra", action="store_true", help="skip the cached_scores extra block")
    ap.add_argument("--batched-runs", default="8,16",
                    help="N = 1: also time R independent runs batched in THIS process (moc_amd.runs), for every R of this "
                         "comma-separated list (the `batched_runs` block; '' or 0: skip)")
    ap.add_argument("--replicas-only", action="store_true", help=argparse.SUPPRESS)      # the child of --packed-runs
    ap.add_argument("--force-dp", action="store_true", help="run the data-parallel trainer even at 1 GPU (rehearsal)")
    ap.add_argument("--force-seq", action="store_true", help="run the exact-sequential multi-GPU trainer even at 1 GPU "
                                                              "(rehearsal: what its pack / gather / compact-batch machinery costs)")
    return ap.parse_args()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks of this script (one per GPU) BEFORE this
    process has made any GPU call, relay rank 0's JSON line and the first non-zero exit code.  (Never re-exec a
    process that has touched the GPU; this parent never does.)"""
    port = _free_port()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MOC_BENCH_CHILD="1")
        out = subprocess.PIPE if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    line, _ = procs[0].communicate()
    rc = procs[0].returncode
    deadline = time.time() + 120
    for p in procs[1:]:
        try:
            p.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()                                   # exactly the child this parent started
            p.wait()
        rc = rc or p.returncode
    sys.stdout.write(line.decode())
    sys.stdout.flush()
    sys.exit(rc)


def workload_name(a):
    C = a.classes
    if C == 2 and a.slides == 32 and a.dim == 512:
        name = "NSCLC 2-way 16-shot"
    elif C == 2 and a.slides == 2 and a.dim == 512:
        name = "NSCLC 2-way 1-shot"
    elif C == 3 and a.slides == 48 and a.dim == 512:
        name = "RCC 3-way 16-shot"
    elif C == 30 and a.slides == 120 and a.dim == 512:
        name = "EBRAINS-30 30-way 4-shot"
    else:
        name = f"synthetic {C}-way"
    return (f"{name} train loop: {a.slides} slides/epoch x {'~' if a.lognormal else ''}{a.patches} patches x {a.dim}, "
            f"topj {a.topj}, topk {a.topk}, row mask on")


def measured_traffic(kname, avg_bytes):
    """HBM bytes per launch from the PMC passes committed under profiles/ (scripts/pmc_traffic.py) -- only when that
    capture is of THIS kernel, THIS source and a launch of THIS size; otherwise None (never a number from another run)."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        src = hashlib.sha256(open(os.path.join(ROOT, "moc_amd", "csrc", "moc_scores.hip"), "rb").read()).hexdigest()[:16]
        for e in tj.get("entries", [tj]):             # one entry per captured launch size (the persistent grid is the same for all)
            if e.get("kernel") != kname or e.get("scores_src_sha16") != src:
                continue
            if abs(e.get("algorithmic_bytes_per_launch", 0) - avg_bytes) > 0.01 * avg_bytes:
                continue
            return {"hbm_bytes_per_launch": e["hbm_bytes_per_launch"], "source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / "
                    "WRITE_SIZE, separate passes, same kernel source and launch size)", "captured_at": e.get("git_head")}
        return None
    except (OSError, ValueError, KeyError):
        return None


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)                      # does not return
    import torch
    # stdout carries the ONE JSON line and nothing else: whatever libraries print there (librccl announces its
    # path on stdout when a communicator is created) goes to stderr instead
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    n_dev = torch.cuda.device_count()
    one_device = os.environ.get("MOC_BENCH_ONE_DEVICE") == "1"     # rehearsal: all ranks on one card (not a measurement)
    assert n_dev >= world or one_device, f"--gpus {world} but this node shows {n_dev} GPU(s)"
    # (rehearsal / packed runs: every rank on the SAME card -- device MOC_BENCH_DEVICE, default 0 -- whatever the node has)
    dev = torch.device("cuda", int(os.environ.get("MOC_BENCH_DEVICE", "0")) % n_dev if one_device else local_rank)
    torch.cuda.set_device(dev)
    cpus = granted_cpus()
    torch.set_num_threads(max(1, min(8, cpus // max(1, world))))   # the GPU leg's host side is serial
    import torch.distributed as dist
    os.environ["MOC_DP_EXCHANGE"] = a.dp_exchange
    dp = world > 1 or a.force_dp
    if dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from moc_amd import engine, main_moc as M, synth
    from moc_amd import dist as mdist

    C, D, j, K = a.classes, a.dim, a.topj, a.topk
    store = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(a.dtype, torch.float32)
    esz = 4 if a.dtype == "fp32" else 2
    W, We = synth.make_bank(1234, D, C)
    M.set_classifier_bank(W.to(dev), We.to(dev))
    args = types.SimpleNamespace(disable_tqdm=True, n_classes=C, topj=j, topk=K, discard_classifiers=[],
                                 pretrain="conch", ablation_study="none")
    all_sizes = synth.bag_sizes(99, a.slides, a.patches, fixed=not a.lognormal)

    def make_split(mode, store=store):
        """This rank's resident train split.  seq: the contiguous block [rank*per, (rank+1)*per) of the ONE task's
        slides (phase A shards, the recurrence does not); dp_strong: slide i of the ONE task lives on rank i mod world
        (step t consumes slides [t*world, (t+1)*world)); dp_weak / single: a whole task of its own per rank."""
        if mode == "seq":
            ids = mdist.block_lists(a.slides, world)[rank]          # contiguous, balanced
            bags = [synth.make_bag_device(1234 + i, all_sizes[i], D, We, C, i % C, dev, store) for i in ids]
            return mdist.SeqShardedBags(bags, all_sizes, [i % C for i in range(a.slides)], dev, rank, world)
        if mode == "dp_strong" and world > 1:
            assert a.slides % world == 0, "--scaling strong: --slides must be a multiple of --gpus"
            ids = list(range(rank, a.slides, world))
            bags = [synth.make_bag_device(1234 + i, all_sizes[i], D, We, C, i % C, dev, store) for i in ids]
            return M.ResidentBags(bags, [i % C for i in ids], dev)
        sizes = all_sizes if rank == 0 else synth.bag_sizes(99 + rank, a.slides, a.patches, fixed=not a.lognormal)
        bags = [synth.make_bag_device(1234 + 1000 * rank + i, n, D, We, C, i % C, dev, store) for i, n in enumerate(sizes)]
        return M.ResidentBags(bags, [i % C for i in range(a.slides)], dev)

    def new_model():
        torch.manual_seed(0)
        m = M.senet(D, 4).to(dev)
        return m, torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-4)

    def fence():
        # (the host spins on an event of the main stream first: hipDeviceSynchronize's blocking wait wakes up 20-30 us
        # after the GPU is done, 5 % of a 20-step region; the synchronize that follows still takes 21-25 us with every
        # stream idle -- MOC_BENCH_TRACE=1 prints the four host times -- and stays inside the region, as the contract asks)
        e = fence.event                  # (one event object, re-recorded: creating one costs 3 us of a 0.5-ms region)
        st = fence.stamps
        if st is not None:
            st.append(time.perf_counter())
        e.record()
        if st is not None:
            st.append(time.perf_counter())
        while not e.query():
            pass
        if st is not None:
            st.append(time.perf_counter())
        torch.cuda.synchronize()
        if st is not None:
            st.append(time.perf_counter())
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    fence.event = torch.cuda.Event()
    fence.stamps = None                  # MOC_BENCH_TRACE=1: host times inside the fence that ends the timed region

    def max_over_ranks(dt):
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            if one_device:
                t = t.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    class Loop:
        """The train loop over one resident split: passes of `per_pass` steps (an epoch), the last one of a run
        possibly partial.  Every pass is told the length of the pass that follows it, so that its phase A -- issued a
        pass ahead on the side stream, as in a run of many epochs -- is for the right visits even when --warmup or
        --steps is not a whole number of epochs."""

        trace = []                       # MOC_BENCH_TRACE=1: host time per pass, printed after the regions

        def __init__(self, res, model, opt, mode):
            self.res, self.model, self.opt, self.mode = res, model, opt, mode
            self.per_pass = res.real_len()

        def schedule(self, n_steps):
            out, done = [], 0
            while done < n_steps:
                out.append(min(self.per_pass, n_steps - done))
                done += out[-1]
            return out

        def run(self, lengths, then=None, then2=None):
            """Passes of the given lengths; `then` / `then2` = lengths of the two passes the caller runs next (None:
            nothing follows, no phase A is issued ahead / no masks are drawn ahead)."""
            res = self.res
            upcoming = list(lengths) + [then, then2 if then is not None else None]
            for i, m in enumerate(lengths):
                nxt, nxt2 = upcoming[i + 1], upcoming[i + 2]
                res.repeat_num = m if m < self.per_pass else None
                res.next_pass_len = nxt if nxt is not None else 0      # 0: no pass follows
                res.pass_after_next_len = nxt2 if nxt2 is not None else 0
                t_in = time.perf_counter()
                if self.mode == "seq":
                    mdist.train_seq(self.model, res, self.opt, dev, args)
                elif self.mode in ("dp_strong", "dp_weak"):
                    mdist.train_dp(self.model, res, self.opt, dev, args)
                else:
                    M.train(self.model, res, self.opt, dev, args)
                if TRACE:
                    Loop.trace.append(f"pass of {m} (next {nxt}, then {nxt2}) issued in {(time.perf_counter() - t_in) * 1e6:.0f} us")
            res.repeat_num = None
            res.next_pass_len = None
            res.pass_after_next_len = None

        def allocate(self, lengths):
            """Work arrays for every pass length that will occur: allocated here, not inside a timed region."""
            if self.mode == "seq":
                bank0 = M._bank_for(self.res.local.X, dev)
                for m in set(lengths):
                    mdist._seq_plan(self.res, m, bank0, args)
                return
            bank0 = M._bank_for(self.res.X, dev)
            for m in set(lengths):
                self.res.repeat_num = m if m < self.per_pass else None
                self.res.train_plan(bank0.C, bank0.Ce, j, K, [])
            self.res.repeat_num = None

    def ranks_agree(model):
        """Data-parallel sanity: no exchange time-out anywhere and bit-identical parameters on every rank."""
        if world == 1:
            return True
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        h = flat.view(torch.int32).to(torch.int64)
        sig = torch.stack([h.sum(), (h * torch.arange(1, h.numel() + 1, device=dev)).sum(),
                           torch.tensor(mdist.exchange_error(), device=dev, dtype=torch.int64)])
        if one_device:
            sig = sig.cpu()
        lo, hi = sig.clone(), sig.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return bool(torch.equal(lo, hi)) and int(hi[2].item()) == 0

    def measure(mode, n_steps, n_warm, steady_epochs, store=store):
        """-> dict(value, dt, steady, loop, exchange, fallback) for one scaling mode.  The timed region is EXACTLY
        n_steps steps between two fences, after n_warm untimed steps of the same model."""
        res = make_split(mode, store)
        model, opt = new_model()
        loop = Loop(res, model, opt, mode)
        units = world if mode in ("dp_strong", "dp_weak", "replicas") else 1     # slides one step consumes, over the whole job
        warm, timed = loop.schedule(n_warm), loop.schedule(n_steps)
        steady = [loop.per_pass] * steady_epochs
        # The loop runs one pass ahead of itself: while the meta-steps of a pass run, phase A of the NEXT pass streams on
        # the side stream (and the masks of the one after are drawn on the host).  A timed region of K steps therefore
        # contains the phase A of whatever pass follows it -- so a pass as long as the region's last one follows it (the
        # `spacer`, untimed): the region then issues phase A for exactly as many slides as it consumes, as every pass of
        # a long run does.  (Before: the 32-slide phase A of the steady block trailed a 20-step region.)
        spacer = [timed[-1]] if steady else []
        loop.allocate(warm + timed + steady)
        fallback = None

        def prime():
            """Runtime warm-up that is not training: the meta-learner goes through passes of every length that will
            occur -- twice each, once per set of work arrays -- so that code objects, allocator pools, the side stream, the
            pass-ahead pipeline and the captured pass graphs (moc_train_steps_graph: keyed on the tensors of THIS model and
            optimizer) exist; then its parameters and Adam state are put back, in place, to what they were.  The model
            that is measured has afterwards taken exactly the W untimed and K timed steps, from its initial values."""
            params = list(model.parameters())
            saved = [p.detach().clone() for p in params]
            tmp = Loop(res, model, opt, mode)
            seq = []
            for m in sorted(set(warm + timed + steady), reverse=True):
                seq += [m] * (4 if m == loop.per_pass else 2)
            tmp.run(seq + [loop.per_pass], then=(warm + timed)[0])
            torch.cuda.synchronize()
            with torch.no_grad():
                for p, s0 in zip(params, saved):
                    p.copy_(s0)
                    st = opt.state[p]
                    st["exp_avg"].zero_()
                    st["exp_avg_sq"].zero_()
                    st["step"].zero_()
            torch.cuda.synchronize()

        # (the cyclic collector may not choose the 0.5-40 ms of a timed region for a full collection: one run in a few
        # dozen measured 11 ms for 100 steps that take 2.4.  Collected HERE, before the runtime warm-up -- a pause of
        # milliseconds any later lets the GPU's clocks fall: five warm-up steps do not bring them back, --steps 20 then
        # measured 23 k instead of 29 k -- and switched off until the regions are over.)
        gc.collect()
        gc.disable()
        after_timed = (spacer + steady + [None, None])[:2]
        prime()
        loop.run(warm, then=timed[0], then2=(timed + after_timed)[1])
        fence()
        exchange = getattr(mdist.train_dp, "exchange", None) if mode.startswith("dp") else None
        if world > 1 and mode != "replicas" and not ranks_agree(model):
            # the in-kernel exchange misbehaved on this node: fall back to the RCCL collective, from scratch
            err = mdist.exchange_error()
            if rank == 0:
                print(f"bench: ranks disagree after warm-up with exchange={exchange} (error word {err}); "
                      "re-running with the collective", file=sys.stderr)
            assert exchange == "p2p", "ranks disagree on the collective path"
            fallback = {"from": "p2p", "exchange_error": err}
            mdist.drop_p2p()                      # collective: the errored exchange is closed and forgotten everywhere
            os.environ["MOC_DP_EXCHANGE"] = "rccl"
            model, opt = new_model()
            loop = Loop(res, model, opt, mode)
            loop.run(warm, then=timed[0], then2=(timed + after_timed)[1])
            fence()
            exchange = mdist.train_dp.exchange
            assert ranks_agree(model), "ranks disagree on the collective path"
        if engine.SCORE_EVENTS is not None:
            engine.SCORE_EVENTS.clear()
        if TRACE:
            Loop.trace.clear()
            fence.stamps = []
        t0 = time.perf_counter()
        loop.run(timed, then=after_timed[0], then2=after_timed[1])
        fence()
        dt = max_over_ranks(time.perf_counter() - t0)
        if TRACE:
            st = fence.stamps
            fence.stamps = None
            print(f"trace: timed region {dt * 1e6:.0f} us: run() returned at {(st[0] - t0) * 1e6:.0f}, event recorded {(st[1] - t0) * 1e6:.0f}, "
                  f"event done {(st[2] - t0) * 1e6:.0f}, synchronize returned {(st[3] - t0) * 1e6:.0f}", file=sys.stderr)
            for ln in Loop.trace + getattr(M.train, "trace_host", [])[-len(timed):]:
                print("trace:   " + ln, file=sys.stderr)
        if TRACE and getattr(M.train, "trace_events", None):
            for n_, e0_, e1_ in M.train.trace_events[-len(timed):]:
                print(f"trace: timed pass of {n_}: GPU time between the pass's first launch and its last kernel {e0_.elapsed_time(e1_) * 1e3:.0f} us; "
                      f"timed region {dt * 1e6:.0f} us", file=sys.stderr)
        ev = list(engine.SCORE_EVENTS) if engine.SCORE_EVENTS is not None else []
        assert mode == "replicas" or ranks_agree(model), "data-parallel ranks ended the timed region with different parameters"
        out = {"value": n_steps * units / dt, "dt": dt, "loop": loop, "exchange": exchange, "fallback": fallback,
               "events": ev, "steady": None, "res": res, "model": model}
        if steady:
            # the same model keeps training: whole epochs only, the pass-ahead pipeline in its periodic state
            engine_events, engine.SCORE_EVENTS = engine.SCORE_EVENTS, None
            loop.run(spacer, then=steady[0], then2=steady[0])
            fence()
            t0 = time.perf_counter()
            loop.run(steady, then=None)
            fence()
            sdt = max_over_ranks(time.perf_counter() - t0)
            engine.SCORE_EVENTS = engine_events
            n = steady_epochs * loop.per_pass
            out["steady"] = {"value": round(n * units / sdt, 1), "unit": "meta-steps/s", "epochs": steady_epochs,
                             "steps": n, "ms_per_step": round(sdt / n * 1e3, 5),
                             "note": "whole epochs of the same model right after the timed region (same process, same "
                                     "clocks): the rate a run of many epochs sees"}
            assert mode == "replicas" or ranks_agree(model), "data-parallel ranks ended the steady-state block with different parameters"
        gc.enable()
        return out

    if a.replicas_only:
        # child of `--packed-runs`: `world` independent runs, timed together; one small JSON line
        r3 = measure("replicas", a.steps, a.warmup, 0)
        if rank == 0:
            os.write(real_stdout, (json.dumps({"runs": world, "value": round(r3["value"], 1), "steps_per_run": a.steps,
                                               "ms_per_step_per_run": round(r3["dt"] / a.steps * 1e3, 5)}) + "\n").encode())
        if dp:
            mdist.shutdown()
            dist.destroy_process_group()
        return
    engine.SCORE_EVENTS = []          # (start, stop, algorithmic bytes) per score-pass launch
    if world == 1:
        main_mode = "dp_weak" if a.force_dp else "seq" if a.force_seq else "single"
    else:
        main_mode = "replicas" if a.train_mode == "runs" else "seq" if a.train_mode == "seq" else "dp_" + a.scaling
    r = measure(main_mode, a.steps, a.warmup, 0 if a.no_steady else a.steady_epochs)
    value, dt, loop, exchange = r["value"], r["dt"], r["loop"], r["exchange"]
    res, model = r["res"], r["model"]

    # ---- roofline of the dominant kernel (score pass), from live HIP events on the launch stream that carry the
    # kernel's own start / end stamps (moc_scores_timed: hipExtLaunchKernel), so that the duration is the one rocprofv3
    # reports for the same launch -- an event pair recorded AROUND the launch adds the queue's 10-15 us on a busy GPU
    ev = r["events"]
    ms = [s.elapsed_time(e) for s, e, _ in ev]
    by = [b for _, _, b in ev]
    engine.SCORE_EVENTS = None
    roof = None
    if ms:
        avg_ms = sum(ms) / len(ms)
        avg_bytes = sum(by) / len(by)
        achieved = avg_bytes / (avg_ms * 1e-3) / 1e9
        # the kernel moc_scores launches for this shape (moc_scores.hip), as rocprofv3 names it
        nt = (C + 4 + 15) // 16
        half, f16 = ("true", "true" if a.dtype == "fp16" else "false") if esz == 2 else ("false", "false")
        ticketed = False
        if nt <= (3 if esz == 2 else 4):
            # (the look-ahead launches of a train pass take the ticketed form and stay off MOC_RESERVE_CUS compute units,
            # engine.RESERVE_CUS; four n-tiles keep the static walk)
            ticketed = engine.RESERVE_CUS > 0 and M.PREFETCH_PHASE_A and nt == 1
            kname = (f"scores_stream_kernel<{16 if (D * esz) % 1024 == 0 else 8}, {half}, {nt}, {f16}, "
                     f"{'true' if ticketed else 'false'}>")
        elif esz == 2 and nt <= 8 and D % 64 == 0:
            kname = f"scores_wide_ring_kernel<{nt}, {f16}>" if nt <= 5 else f"scores_wide_kernel<{nt}, {f16}>"
        else:
            kname = f"scores_kernel<{512 if D % 512 == 0 else 256}, {half}, {f16}>"
        roof = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "avg_launch_us": round(avg_ms * 1e3, 2), "algorithmic_bytes_per_launch": int(avg_bytes),
                "launches": len(ms)}
        if ticketed:
            roof["placement"] = {"compute_units_left_to_the_meta_steps": engine.RESERVE_CUS, "tiles": "by ticket",
                                 "note": "the timed launches run beside the previous pass's meta-steps and stay off that many "
                                         "of the 256 compute units (include/moc_hip.h); `whole_chip` is the same kernel as an "
                                         "evaluation pass launches it: every compute unit, static walk, nothing beside it"}
        # The statistics the pass must write -- (2C + 3) floats and a flag byte per kept row -- are not in SURVEY.md's
        # algorithmic figure (reads of the bag: 8d); they are 3 % of the bytes at two classes and 25 % at thirty, and at
        # that ratio HBM serves reads at 4.1-4.2 TB/s whatever the shape of the stores (profiles/round2_store_shape_bench.txt)
        stat_b = avg_bytes / (D * esz) * ((2 * C + 3) * 4 + 1)
        roof["with_statistics_writes"] = {"bytes_per_launch": int(avg_bytes + stat_b),
                                          "achieved": round((avg_bytes + stat_b) / (avg_ms * 1e-3) / 1e9, 1),
                                          "frac": round((avg_bytes + stat_b) / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        tr = measured_traffic(kname, avg_bytes)
        if tr:
            roof["traffic"] = tr["hbm_bytes_per_launch"]
            roof["traffic_source"] = tr["source"]
        # the same launch with nothing else on the GPU (in the loop it shares the memory system with the previous
        # pass's meta-steps on the main stream): the last train batch, as it stands, ten launches
        last = (mdist.train_seq.last_local if main_mode == "seq" else
                mdist.train_dp.last[0] if main_mode.startswith("dp") else M.train.last[0])
        Xl = res.local.X if main_mode == "seq" else res.X
        torch.cuda.synchronize()

        def alone_ms(batch):
            """Median duration of the batch's score launch with nothing else on the GPU: launches back to back on the one
            stream (each is alone: the stream serialises them), three of them untimed first and no host synchronisation
            in between -- a launch after an idle gap runs on clocks that have dropped (round 3 timed every launch behind a
            fence and reported `alone` SLOWER than the live launches on the wide configurations)."""
            bank_ = M._bank_for(Xl, dev)
            for _ in range(3):
                engine.timed_scores(batch, bank_)
            evs = [engine.timed_scores(batch, bank_) for _ in range(10)]
            torch.cuda.synchronize()
            t = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
            # the FASTEST of the ten: what the kernel does with nothing beside it.  (Ten launches of a 0.25-1 ms streaming
            # kernel back to back run a few per cent slower than the same launch between a pass's meta-steps -- sustained
            # HBM load against a duty cycle of one launch per pass -- so their median can sit below the live figure.)
            return t[0]

        iso_ms = alone_ms(last)
        iso_bytes = last.kept_rows_host * D * esz
        roof["alone"] = {"achieved": round(iso_bytes / (iso_ms * 1e-3) / 1e9, 1),
                         "frac": round(iso_bytes / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "launch_us": round(iso_ms * 1e3, 2),
                         "algorithmic_bytes": int(iso_bytes)}
        if ticketed and last.c.tile_ticket is not None:
            keep = (last.c.tile_ticket, last.c.cu_reserved)
            last.c.tile_ticket, last.c.cu_reserved = None, None
            try:
                w_ms = alone_ms(last)
            finally:
                last.c.tile_ticket, last.c.cu_reserved = keep
            roof["whole_chip"] = {"kernel": kname.replace(", true>", ", false>"), "achieved": round(iso_bytes / (w_ms * 1e-3) / 1e9, 1),
                                  "frac": round(iso_bytes / (w_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "launch_us": round(w_ms * 1e3, 2)}

    # ---- the same loop on 16-bit copies of the bags (N = 1, default fp32 run): BASELINE configs[1] names bf16 storage,
    # but the trained AUC does not stay within +-0.002 of the fp32 reference there, so it is an extra block, not `value`
    half_block = None
    if world == 1 and a.dtype == "fp32" and main_mode == "single" and not a.no_16bit_extra:
        try:
            ev_keep, engine.SCORE_EVENTS = engine.SCORE_EVENTS, None
            rh = measure("single", a.steps, a.warmup, 0 if a.no_steady else min(20, a.steady_epochs), store=torch.bfloat16)
            engine.SCORE_EVENTS = ev_keep
            half_block = {"bag_storage": "bf16", "value": round(rh["value"], 1), "unit": "meta-steps/s", "steps": a.steps, "warmup": a.warmup,
                          "steady_state": rh["steady"] and rh["steady"]["value"],
                          "fidelity": "NOT the reference's numbers: from the same fp32 bags, bf16 storage moves a fixed model's pooled "
                                      "logits by 1e-4..1e-2, per-epoch validation AUC by up to 0.06, test AUC at best val by "
                                      "0.0006-0.03 over 9 tasks (fp16: 0.0005-0.04); evaluation of an untrained model stays within "
                                      "5e-6 (loss) / equal AUC.  The sequential loop's own seed-to-seed spread on those tasks is "
                                      "0.002-0.06 (std), so this is reseeding-sized noise -- but outside north_star's +-0.002 "
                                      "(profiles/round3_storage_fidelity.jsonl, round3_dp_auc_control.jsonl)"}
            rh["loop"] = rh["res"] = rh["model"] = None
            del rh
        except Exception as e:  # noqa: BLE001 -- an extra block must not cost the line its `value`
            half_block = {"error": f"{type(e).__name__}: {e}"[:300]}

    # ---- the modes that are not `value`, from shorter runs of this process, as extra keys (N > 1 only)
    extras = {}
    seq_block = None
    if world > 1 and main_mode == "replicas" and a.seq_extra:
        try:
            k2 = max(a.slides, min(a.steps, 10 * a.slides))
            r2 = measure("seq", k2, min(a.warmup, a.slides), 0)
            seq_block = {"value": round(r2["value"], 1), "unit": "meta-steps/s (ONE run over all ranks)", "steps": k2, "scaling": "strong",
                         "note": "exact-sequential: bags and phase A sharded over the GPUs, compact results all-gathered (RCCL), every "
                                 "rank runs the one-Adam-step-per-slide recurrence -- bit-identical to one GPU; the recurrence is a "
                                 "latency chain that a second GPU does not shorten"}
            r2["loop"] = r2["res"] = None
        except Exception as e:  # noqa: BLE001 -- an extra block must not cost the line its `value`
            seq_block = {"error": f"{type(e).__name__}: {e}"[:300]}
    if world > 1 and (main_mode.startswith("dp") or a.dp_extra):
        del res, loop
        r["res"] = r["loop"] = None
        todo = [m for m in ("dp_strong", "dp_weak") if m != main_mode and not (m == "dp_strong" and a.slides % world)]
        for omode in todo:
            per = a.slides if omode == "dp_weak" else a.slides // world
            k2 = max(per, min(a.steps, 10 * per))
            try:
                r2 = measure(omode, k2, min(a.warmup, per), 0)
            except Exception as e:  # noqa: BLE001 -- an extra block must not cost the line its `value`
                extras[omode] = {"error": f"{type(e).__name__}: {e}"[:300]}
                continue
            extras[omode] = {"value": round(r2["value"], 1), "unit": "meta-steps/s (slides consumed by all ranks)",
                             "sync_steps": k2, "ms_per_sync_step": round(r2["dt"] / k2 * 1e3, 5), "exchange": r2["exchange"]}
            if r2["fallback"]:
                extras[omode]["exchange_fallback"] = r2["fallback"]
            model = r2["model"]
            r2["loop"] = r2["res"] = None

    # ---- N independent runs, one per GPU, no communication at all: how the reference itself uses several GPUs
    # (scripts/moc_train.sh gives every fold x shot its own process and GPU) -- every run the reference's trajectory
    replicas = None
    if world > 1 and not a.no_replicas and main_mode != "replicas":
        try:
            k3 = max(a.slides, min(a.steps, 10 * a.slides))
            r3 = measure("replicas", k3, min(a.warmup, a.slides), 0)
            replicas = {"value": round(r3["value"], 1), "unit": "meta-steps/s (all runs together)", "steps_per_run": k3,
                        "ms_per_step_per_run": round(r3["dt"] / k3 * 1e3, 5), "scaling": "weak",
                        "note": f"{world} independent training runs ({a.slides} slides of its own each), one per GPU, no collective: "
                                "the reference's own way of filling a node (scripts/moc_train.sh: one process per fold x shot)"}
            r3["loop"] = r3["res"] = None
        except Exception as e:  # noqa: BLE001
            replicas = {"error": f"{type(e).__name__}: {e}"[:300]}

    # ---- evaluation throughput (slides/sec), batched end to end; slides sharded over the ranks, no data-path collective
    eval_rate = None
    if not a.no_eval:
        n_eval = a.eval_slides
        ids = list(range(rank, n_eval, world))
        esizes = synth.bag_sizes(7, n_eval, a.patches, fixed=not a.lognormal)
        ebags = [synth.make_bag_device(777 + i, esizes[i], D, We, C, i % C, dev, store) for i in ids]
        eres = M.ResidentBags(ebags, [i % C for i in ids], dev)
        del ebags

        def eval_once():
            if world == 1:
                return M.evaluation(model, eres, dev, args)
            return mdist.evaluation_dp(model, eres, dev, args, [i % C for i in range(n_eval)], ids,
                                       [list(range(q, n_eval, world)) for q in range(world)])
        for _ in range(3):                       # first pass allocates the plan; two more settle the clocks
            eval_once()
        fence()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            eval_once()
        fence()
        edt = max_over_ranks(time.perf_counter() - t0)
        eval_rate = reps * n_eval / edt
        del eres

    # ---- several independent runs side by side on this ONE GPU (separate processes; nothing of this process runs meanwhile)
    packed = None
    if world == 1 and a.packed_runs > 1 and not (a.force_dp or a.force_seq):
        torch.cuda.synchronize()
        k3 = 20 * a.slides                      # its own length, whatever --steps is: the block states it (steps_per_run)
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(a.packed_runs), "--replicas-only", "--steps", str(k3),
               "--warmup", str(a.slides), "--dtype", a.dtype, "--slides", str(a.slides), "--patches", str(a.patches),
               "--classes", str(C), "--dim", str(D), "--topj", str(j), "--topk", str(K)] + (["--lognormal"] if a.lognormal else [])
        env = {k_: v for k_, v in os.environ.items() if k_ not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
        env["MOC_BENCH_ONE_DEVICE"] = "1"
        env["MOC_BENCH_DEVICE"] = str(dev.index or 0)
        try:
            cp = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
            line = [ln for ln in cp.stdout.decode().splitlines() if ln.startswith("{")]
            packed = json.loads(line[-1]) if cp.returncode == 0 and line else {"error": f"child exited {cp.returncode}"}
        except Exception as e:  # noqa: BLE001 -- an extra block must not cost the line its `value`
            packed = {"error": f"{type(e).__name__}: {e}"[:300]}
        if "value" in packed:
            packed.update(unit="meta-steps/s (all runs together, ONE GPU)",
                          vs_one_run=round(packed["value"] / value, 2),
                          note=f"{a.packed_runs} independent training runs as separate processes on the same GPU -- the reference's "
                               "scripts/moc_train.sh packs five folds onto one GPU; moc_amd.run_many is its job queue.  One run is a "
                               "latency chain that leaves most of the GPU idle; independent runs interleave")

    # ---- the same, inside ONE process: R independent runs stepped in lockstep by one launch pair per meta-step
    # (moc_amd.runs / moc_train_steps_runs).  `value` above stays the rate of ONE run.
    batched = None

    def run_batched(R_):
            try:
                torch.cuda.synchronize()
                models_, opts_, splits_ = [], [], []
                for r_ in range(R_):
                    bags_ = [synth.make_bag_device(50000 + 1000 * r_ + i, a.patches, D, We, C, i % C, dev, store) for i in range(a.slides)]
                    splits_.append(M.ResidentBags(bags_, [i % C for i in range(a.slides)], dev))
                    del bags_
                    torch.manual_seed(r_)
                    m_ = M.senet(D, 4).to(dev)
                    models_.append(m_)
                    opts_.append(torch.optim.Adam(m_.parameters(), lr=1e-3, weight_decay=1e-4))
                for _ in range(6):
                    M.train_runs(models_, splits_, opts_, dev, args)
                fence()
                engine.SCORE_EVENTS = []
                E_ = 40
                t0 = time.perf_counter()
                for _ in range(E_):
                    M.train_runs(models_, splits_, opts_, dev, args)
                fence()
                bdt = time.perf_counter() - t0
                ev_, engine.SCORE_EVENTS = engine.SCORE_EVENTS, None
                sms = sum(s_.elapsed_time(e_) for s_, e_, _ in ev_)
                sby = sum(b_ for _, _, b_ in ev_)
                blk = {
                    "runs": R_, "value": round(R_ * a.slides * E_ / bdt, 1), "unit": "meta-steps/s (all runs together, ONE GPU, ONE process)",
                    "vs_one_run": round(R_ * a.slides * E_ / bdt / value, 2), "passes": E_, "us_per_pass": round(bdt / E_ * 1e6, 1)}
                if sms > 0:
                    blk["score_pass"] = {"achieved": round(sby / (sms * 1e-3) / 1e9, 1), "frac": round(sby / (sms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                         "unit": "GB/s", "avg_launch_us": round(sms / max(1, len(ev_)) * 1e3, 1),
                                         "note": "one launch over the R x %d slides of a pass, whole chip, nothing beside it" % a.slides}
                del models_, opts_, splits_
                M._run_sets.clear()
                torch.cuda.empty_cache()
                return blk
            except Exception as e:  # noqa: BLE001 -- an extra block must not cost the line its `value`
                engine.SCORE_EVENTS = None
                M._run_sets.clear()
                return {"error": f"{type(e).__name__}: {e}"[:300]}

    if world == 1 and a.batched_runs not in ("", "0") and not a.replicas_only and not (a.force_dp or a.force_seq):
        batched = {}
        for R_ in [int(v) for v in a.batched_runs.split(",") if int(v) > 1]:
            batched[f"runs_{R_}"] = run_batched(R_)
        batched["note"] = ("R independent training runs (own slides, parameters, Adam state, mask stream) in ONE process: one forward "
                           "and one step launch per meta-step serve all of them (grid.y / grid.z = run), phase A over all R x n "
                           "slides in one pass.  Per run bit-identical to main_moc.train (tests/test_gpu_runs.py).  The reference's "
                           "scripts/moc_train.sh:11-31 starts these runs as separate processes")

    # ---- opt-in: the per-row statistics kept from ONE score pass (ResidentBags(cache_scores=True)): no score pass per epoch.
    # NOT `value` -- its score pass reads the bags every pass, as the reference recomputes them -- an extra block
    cached_block = None
    if world == 1 and main_mode == "single" and not a.no_cached_extra and not a.replicas_only:
        try:
            ev_keep, engine.SCORE_EVENTS = engine.SCORE_EVENTS, None
            keep_env = os.environ.get("MOC_CACHE_SCORES")
            os.environ["MOC_CACHE_SCORES"] = "1"
            try:
                rc_ = measure("single", a.steps, a.warmup, 0 if a.no_steady else min(200, a.steady_epochs))
                cached_runs8 = run_batched(8) if a.batched_runs not in ("", "0") else None
            finally:
                if keep_env is None:
                    os.environ.pop("MOC_CACHE_SCORES", None)
                else:
                    os.environ["MOC_CACHE_SCORES"] = keep_env
            engine.SCORE_EVENTS = ev_keep
            cached_block = {"value": round(rc_["value"], 1), "unit": "meta-steps/s", "steps": a.steps, "warmup": a.warmup,
                            "steady_state": rc_["steady"] and rc_["steady"]["value"], "batched_runs_8": cached_runs8,
                            "note": "opt-in (ResidentBags(cache_scores=True), run_moc --cache_scores 1): the classifier bank is frozen, so a "
                                    "row's statistics are the same on every visit; they are kept from ONE unmasked score pass (28 B per "
                                    "2-KiB row) and a train pass copies its kept rows' statistics instead of reading the bags again -- "
                                    "bit-identical training (tests: test_cached_statistics_give_the_score_pass_bits).  Not `value`: "
                                    "there the score pass reads the bags every pass, as the reference recomputes feat @ W "
                                    "(main_moc.py:336-337)"}
            rc_["loop"] = rc_["res"] = rc_["model"] = None
            del rc_
        except Exception as e:  # noqa: BLE001 -- an extra block must not cost the line its `value`
            cached_block = {"error": f"{type(e).__name__}: {e}"[:300]}

    # ---- CPU baseline: the oracle's train loop AND evaluation loop on the host cores (rank 0, N=1 only), a bounded
    # sample of the same workload (slide-granular: wide configurations do not get through an epoch in the budget)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu and isinstance(res, M.ResidentBags):
        import torch.nn.functional as F
        from oracle import moc_oracle as O
        labels = res.labels
        cpu_bags = [res.X[res.starts[i]:res.starts[i + 1]].to(torch.float32).cpu() for i in range(a.slides)]
        torch.manual_seed(0)
        cm = O.Senet(D, 4)
        co = O.make_optimizer(cm)

        def cpu_train(ids):
            O.train_epoch(cm, co, [cpu_bags[i] for i in ids], [labels[i] for i in ids], W, We, C, j, K)

        def cpu_eval(ids):
            """evaluation()'s per-slide work (main_moc.py:472-498); the metrics tail over [n, C] logits is left out"""
            cm.eval()
            with torch.no_grad():
                for i in ids:
                    sr = O.slide_process(cpu_bags[i], W, We, C, j)
                    pooled = O.pool_top(O.mix_eval(cm(sr["selected_feat"]), sr, ()), [K])[1][K]
                    F.cross_entropy(pooled, torch.as_tensor(labels[i]).view(1)).item()
            cm.train()

        # pick the fastest intra-op thread count the quota allows (more is not faster: section 6 probes)
        t0 = time.perf_counter()
        cpu_train([0])
        one = time.perf_counter() - t0                       # (first touch included: only sizes the calibration)
        n_cal = 8 if one < 0.05 else 2
        best = None
        for th in sorted({1, 4, 8, 16, cpus} & set(range(1, cpus + 1))):
            torch.set_num_threads(th)
            cpu_train([0])                                   # warm-up at this width
            t0 = time.perf_counter()
            cpu_train([i % a.slides for i in range(n_cal)])
            rate = n_cal / (time.perf_counter() - t0)
            if best is None or rate > best[1]:
                best = (th, rate)
        threads = best[0]
        torch.set_num_threads(threads)
        n_done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < a.cpu_seconds:
            cpu_train([n_done % a.slides])
            n_done += 1
        cdt = time.perf_counter() - t0
        n_ev, t0 = 0, time.perf_counter()
        cpu_eval([0])
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < a.cpu_seconds / 2:
            cpu_eval([n_ev % a.slides])
            n_ev += 1
        edt_cpu = time.perf_counter() - t0
        cpu = {"value": round(n_done / cdt, 2), "unit": "meta-steps/s", "cores": threads, "kind": "port",
               "eval_slides_per_sec": round(n_ev / edt_cpu, 2),
               "sample": f"train: {n_done} meta-steps in {cdt:.1f} s over the run's own {a.slides} train slides in loader order "
                         f"({n_done / a.slides:.2f} epochs; row mask on); eval: {n_ev} full-bag slides in {edt_cpu:.1f} s (the same "
                         f"slides, no mask; evaluation()'s per-slide work, metrics tail left out); fp32 copies of the bag "
                         f"values, torch-CPU oracle, {threads} threads"}
        del cpu_bags

    if rank == 0:
        if world == 1 and main_mode != "seq":
            par = "single GPU, one Adam step per slide"
        elif main_mode == "replicas":
            par = (f"runs x GPUs: {world} independent training runs ({a.slides} slides of its own each), one per GPU, NOTHING exchanged "
                   "-- what shards naturally in training is whole runs (the reference's scripts/moc_train.sh: one process per fold x "
                   "shot); every run is the reference's one-Adam-step-per-slide trajectory, bit for bit")
        elif main_mode == "seq":
            par = (f"seq{world}: exact-sequential -- bags and phase A (mask, scores, selectors, union) sharded over the {world} GPUs "
                   f"in contiguous balanced blocks of {a.slides // world}-{(a.slides + world - 1) // world} slides, unpadded compact results "
                   "all-gathered (RCCL) a pass ahead, every rank runs the one-Adam-step-per-slide recurrence: bit-identical to one GPU")
        else:
            how = ("inside the step kernel (peer-mapped xGMI buffers)" if exchange == "p2p" else "by one RCCL all-reduce")
            par = (f"dp{world}: one slide per rank per synchronous step, 33,092-float meta-gradient summed over the ranks {how}; "
                   + (f"the same {a.slides} slides sharded over the ranks ({a.slides // world} steps per epoch)"
                      if main_mode == "dp_strong" else f"{a.slides} slides of its own per rank")
                   + "; CHANGES the optimisation trajectory (one Adam step per N slides)")
        out = {
            "metric": "meta-steps/sec (train), slides/sec (eval) on 16-shot NSCLC synthetic bags",
            "value": round(value, 1), "unit": "meta-steps/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 5), "higher_is_better": True,
            "scaling": "weak" if main_mode in ("single", "dp_weak", "replicas") else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_name(a), "bag_storage": a.dtype, "arithmetic": "fp32 accumulate (MFMA)",
                       "parallelism": par},
            "steady_state": r["steady"],
            "eval_slides_per_sec": None if eval_rate is None else round(eval_rate, 1),
            "roofline": roof, "cpu_baseline": cpu,
        }
        if half_block:
            out["bf16_storage"] = half_block
        if world > 1:
            out["ranks_seen"] = dist.get_world_size()
            try:
                out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:  # noqa: BLE001
                out["rccl_version"] = f"unavailable ({type(e).__name__})"
        if replicas:
            out["replicas"] = replicas
        if seq_block:
            out["exact_sequential"] = seq_block
        if packed:
            out["packed_runs"] = packed
        if batched:
            out["batched_runs"] = batched
        if cached_block:
            out["cached_scores"] = cached_block
        if extras:
            out["minibatch_dp" if main_mode in ("seq", "replicas") else "other_modes"] = dict(
                extras, note="synchronous minibatch data parallelism: one Adam step per N slides -- an opt-in extension "
                             "(--train-mode dp), its AUC is NOT within +-0.002 of the sequential reference "
                             "(profiles/round2_dp_auc_study.jsonl)")
        if r["fallback"]:
            out["exchange_fallback"] = r["fallback"]
        if one_device:
            out["rehearsal"] = "all ranks on ONE device (MOC_BENCH_ONE_DEVICE=1): not a scaling measurement"
        if cpu:
            out["speedup_vs_cpu_baseline"] = round(value / cpu["value"], 1)
            if eval_rate is not None and cpu.get("eval_slides_per_sec"):
                out["eval_speedup_vs_cpu_baseline"] = round(eval_rate / cpu["eval_slides_per_sec"], 1)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dp:
        mdist.shutdown()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
