#!/usr/bin/env python3
"""Run ON the GPU box (through gpurun): the rocprofv3 passes of one bench command, each into its OWN directory under
gpurun_out/<tag>/, plus a meta.json that says what was profiled (git head handed in by the caller -- the box has no
.git --, the bench arguments, sha256 of every kernel source, the kernel the capture must contain).

    python scripts/profile_capture.py --tag round2_e30 --head $(git rev-parse --short HEAD) \
        --expect "scores_stream_kernel<16, true, 3, false>" [--no-pmc] -- --classes 30 --slides 120 --steps 360 --warmup 120 --no-cpu

Three passes (MI355X_MICROARCH.md: counters in their own runs, FETCH_SIZE and WRITE_SIZE apart; never combined with
other trace domains):  stats = --kernel-trace --stats;  pmc_fetch = --kernel-trace --pmc FETCH_SIZE;
pmc_write = --kernel-trace --pmc WRITE_SIZE.  The program after `--` is python3 itself (no shell, no env wrapper).
scripts/profile_summaries.py <tag> turns the result into the files committed under profiles/ and refuses anything whose
meta.json does not match."""
import argparse, hashlib, json, os, subprocess, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--tag", required=True)
ap.add_argument("--head", required=True)
ap.add_argument("--expect", action="append", default=[], help="kernel name (as rocprofv3 prints it, without namespace) that must appear")
ap.add_argument("--no-pmc", action="store_true")
ap.add_argument("--program", default="bench.py")
ap.add_argument("rest", nargs=argparse.REMAINDER)
a = ap.parse_args()
bench_args = [x for x in a.rest if x != "--"]
out = os.path.join(root, "gpurun_out", a.tag)
os.makedirs(out, exist_ok=True)
src = {}
for f in sorted(os.listdir(os.path.join(root, "moc_amd", "csrc"))):
    if f.endswith((".hip", ".h", ".cpp")):
        src[f] = hashlib.sha256(open(os.path.join(root, "moc_amd", "csrc", f), "rb").read()).hexdigest()[:16]
meta = {"tag": a.tag, "git_head": a.head, "program": a.program, "bench_args": bench_args, "expect_kernels": a.expect,
        "source_sha16": src, "started": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "passes": {}}
env = dict(os.environ, TMPDIR="/tmp")
passes = [("stats", ["--kernel-trace", "--stats"])]
if not a.no_pmc:
    passes += [("pmc_fetch", ["--kernel-trace", "--pmc", "FETCH_SIZE"]), ("pmc_write", ["--kernel-trace", "--pmc", "WRITE_SIZE"])]
rc_all = 0
for name, flags in passes:
    d = os.path.join(out, name)
    cmd = ["rocprofv3", *flags, "--output-format", "csv", "-d", d, "-o", "run", "--", sys.executable, os.path.join(root, a.program), *bench_args]
    t0 = time.time()
    with open(os.path.join(out, name + ".log"), "w") as log:
        p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=log)
    line = p.stdout.decode().strip().splitlines()
    meta["passes"][name] = {"rc": p.returncode, "seconds": round(time.time() - t0, 1), "cmd": " ".join(cmd[:cmd.index("--") + 1]) + " python3 " + a.program + " " + " ".join(bench_args),
                            "bench_line": (json.loads(line[-1]) if line and line[-1].startswith("{") else None)}
    print(f"{name}: rc={p.returncode} {time.time() - t0:.0f}s", flush=True)
    rc_all = rc_all or p.returncode
    json.dump(meta, open(os.path.join(out, "meta.json"), "w"), indent=1)
sys.exit(rc_all)
