"""Many runs over the GPUs of a node: the reference's scripts/moc_train.sh (folds x shots, each its own
`python main_moc.py` process pinned to a GPU with CUDA_VISIBLE_DEVICES, five folds side by side on one GPU per shot)
as a job queue over moc_amd.run_moc.

    python -m moc_amd.run_many --folds 0 1 2 3 4 --shots 1 2 4 8 16 --gpus 0 1 2 3 --runs-per-gpu 4 \
        --result_dir results/moc_train/nsclc -- \
        --dataset nsclc --topj 400 --topk 10 --disable_tqdm

Why several runs per GPU: one run is a chain of dependent 20-us meta-steps (one Adam step per slide, main_moc.py:380-410)
that no second GPU shortens and that leaves most of an MI355X idle; independent runs interleave on the device.  Measured
(bench.py `packed_runs`, NSCLC 16-shot shape, one MI355X): 1 run 43 k meta-steps/s, 2 runs 67 k together, 3 runs 93 k,
4 runs 112 k; 5 or 6 runs fall back to 53-62 k (more processes than the GPU has hardware queues for their streams:
DESIGN.md section 12) -- hence the default of 4.  288 GB of HBM hold the resident splits of dozens of runs (0.5 GB each at bf16).  Every run is
the reference's exact trajectory -- nothing is exchanged between them.

Layout of the results, as the reference's script: `<result_dir>/<shot>_shot/` holds the json / checkpoint files of every
fold of that shot plus `fold_<f>_shot_<s>_output.txt` (stdout + stderr of the run).  Exit code: 0 when every run ended 0.
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
import time


def parse(argv=None):
    p = argparse.ArgumentParser(description="folds x shots of moc_amd.run_moc over the GPUs of a node")
    p.add_argument("--folds", type=int, nargs="+", default=[0, 1, 2, 3, 4])
    p.add_argument("--shots", type=int, nargs="+", default=[1, 2, 4, 8])
    p.add_argument("--gpus", type=int, nargs="+", default=None, help="device ids (default: every visible GPU)")
    p.add_argument("--runs-per-gpu", type=int, default=4, help="runs side by side on one GPU (measured sweet spot: 4)")
    p.add_argument("--result_dir", type=str, default="results/moc_train", help="base directory; one <shot>_shot/ below it per shot")
    p.add_argument("--seed", type=int, default=None, help="base seed: run (shot, fold) gets seed + 100 * shot + fold")
    p.add_argument("--dry-run", action="store_true", help="print the commands and the slot each would take; run nothing")
    p.add_argument("--runner", nargs="+", default=None, help=argparse.SUPPRESS)      # tests: the program to start instead of run_moc
    p.add_argument("rest", nargs=argparse.REMAINDER, help="-- followed by arguments handed to every moc_amd.run_moc")
    a = p.parse_args(argv)
    a.rest = [x for x in a.rest if x != "--"]
    assert "--fold" not in a.rest and "--shot" not in a.rest and "--result_dir" not in a.rest, \
        "--fold / --shot / --result_dir are set per run: give --folds / --shots / --result_dir to run_many itself"
    return a


def jobs_of(a):
    """(shot, fold) in the order the reference's script starts them: shot by shot, folds ascending."""
    return [(s, f) for s in a.shots for f in a.folds]


def command(a, shot, fold):
    out_dir = os.path.join(a.result_dir, f"{shot}_shot")
    base = a.runner if a.runner else [sys.executable, "-m", "moc_amd.run_moc"]
    cmd = base + ["--fold", str(fold), "--shot", str(shot), "--result_dir", out_dir] + a.rest
    if a.seed is not None:
        cmd += ["--seed", str(a.seed + 100 * shot + fold)]
    return cmd, out_dir, os.path.join(out_dir, f"fold_{fold}_shot_{shot}_output.txt")


def visible_gpus():
    """What a child's HIP_VISIBLE_DEVICES must say for each device THIS process sees, in order.  With a HIP / CUDA mask
    in the parent's environment (a scheduler's allocation) those are the mask's own entries -- bare indices 0..n-1 would
    point at the node's first GPUs, outside the allocation; ROCR_VISIBLE_DEVICES is left in place, HIP indices being
    relative to it."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip():
            return [e.strip() for e in v.split(",") if e.strip()]
    import torch
    return [str(i) for i in range(torch.cuda.device_count())]          # (counting devices does not initialise the GPU)


def main(argv=None):
    a = parse(argv)
    seen = visible_gpus()
    if a.gpus is not None:                                   # --gpus: positions among the devices this process sees
        assert all(0 <= g < len(seen) for g in a.gpus) or not seen, f"--gpus {a.gpus}: this process sees {len(seen)} device(s)"
        gpus = [seen[g] if seen else str(g) for g in a.gpus]
    else:
        gpus = seen
    assert gpus and a.runs_per_gpu >= 1, "no GPU to run on"
    queue = jobs_of(a)
    slots = {(g, k): None for g in gpus for k in range(a.runs_per_gpu)}       # slot -> (Popen, shot, fold, log, t0)
    if a.dry_run:
        order = sorted(slots, key=lambda k: (k[1], k[0]))      # as below: every GPU's first slot before any second one
        for i, (shot, fold) in enumerate(queue):
            cmd, _, log = command(a, shot, fold)
            print(f"gpu {order[i % len(order)][0]}: {' '.join(cmd)} >> {log}")
        return 0
    failed, done = [], []
    t_start = time.time()
    while queue or any(v is not None for v in slots.values()):
        for key in sorted(slots, key=lambda k: (k[1], k[0])):            # fill every GPU's first slot before any second one
            if slots[key] is None and queue:
                shot, fold = queue.pop(0)
                cmd, out_dir, log = command(a, shot, fold)
                os.makedirs(out_dir, exist_ok=True)
                env = dict(os.environ, HIP_VISIBLE_DEVICES=str(key[0]), CUDA_VISIBLE_DEVICES=str(key[0]))
                pkg_parent = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))     # `-m moc_amd.run_moc` from any cwd
                env["PYTHONPATH"] = pkg_parent + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
                fh = open(log, "a")
                proc = subprocess.Popen(cmd, stdout=fh, stderr=subprocess.STDOUT, env=env)
                slots[key] = (proc, shot, fold, fh, time.time())
                print(f"[{time.time() - t_start:7.1f}s] start shot {shot} fold {fold} on gpu {key[0]} (slot {key[1]})", flush=True)
        time.sleep(0.2)
        for key, v in slots.items():
            if v is not None and v[0].poll() is not None:
                proc, shot, fold, fh, t0 = v
                fh.close()
                slots[key] = None
                (done if proc.returncode == 0 else failed).append((shot, fold, proc.returncode, time.time() - t0))
                print(f"[{time.time() - t_start:7.1f}s] shot {shot} fold {fold} ended rc={proc.returncode} after {time.time() - t0:.1f}s", flush=True)
    print(f"{len(done)} runs ended 0, {len(failed)} failed, {time.time() - t_start:.1f}s in all")
    for shot, fold, rc, _ in failed:
        print(f"  FAILED shot {shot} fold {fold} (rc {rc}): see {command(a, shot, fold)[2]}")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
