#!/usr/bin/env python3
"""Turn ONE capture made by scripts/profile_capture.py (gpurun_out/<tag>/{stats,pmc_fetch,pmc_write}/ + meta.json) into
the summaries committed under profiles/:

    python scripts/profile_summaries.py <tag> [--traffic "<kernel name>"]

  gpurun_out/<tag>/stats/**/*kernel_stats.csv   ->  profiles/<tag>_kernel_stats.csv       (as is)
  gpurun_out/<tag>/stats/**/*kernel_trace.csv   ->  profiles/<tag>_per_grid_medians.csv   (median/min/max per kernel and grid)
  gpurun_out/<tag>/pmc_{fetch,write}/**/*counter_collection.csv -> profiles/<tag>_pmc_{fetch,write}.csv
                                                   (library kernels, first 4 dispatches per kernel/grid, counter values in KiB)
  gpurun_out/<tag>/meta.json                    ->  profiles/<tag>_meta.json              (git head, bench args, source hashes, bench lines)
  --traffic K: also profiles/traffic.json for kernel K (what bench.py may quote as roofline.traffic -- only for the same
               kernel, the same moc_scores.hip and a launch of the same algorithmic size)

Every input is looked up under the TAG's own directory, and the capture's meta.json must (a) carry this tag, (b) name
source hashes equal to the working tree's, (c) list kernels that are really present in each PMC file -- otherwise nothing
is written (round 1 re-labelled a stale capture four times this way)."""
import csv, glob, hashlib, json, os, shutil, statistics, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
traffic_kernel = sys.argv[sys.argv.index("--traffic") + 1] if "--traffic" in sys.argv else None
force_src = "--allow-source-drift" in sys.argv
cap = os.path.join(root, "gpurun_out", tag)
meta = json.load(open(os.path.join(cap, "meta.json")))
assert meta["tag"] == tag, f"meta.json is of capture {meta['tag']!r}, not {tag!r}"
for f, h in meta["source_sha16"].items():
    now = hashlib.sha256(open(os.path.join(root, "moc_amd", "csrc", f), "rb").read()).hexdigest()[:16]
    if now != h and not force_src:
        sys.exit(f"refusing: {f} changed since the capture ({h} -> {now}); re-capture, or pass --allow-source-drift and say so")
for name, p in meta["passes"].items():
    assert p["rc"] == 0, f"pass {name} exited {p['rc']}"


def find(sub, pat):
    hits = glob.glob(os.path.join(cap, sub, "**", pat), recursive=True)
    assert hits, f"no {pat} under gpurun_out/{tag}/{sub}"
    return hits[0]


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def ours(n):
    return "anonymous namespace" in n and "at::native" not in n


shutil.copy(find("stats", "*kernel_stats.csv"), os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"))
groups = {}
for r in csv.DictReader(open(find("stats", "*kernel_trace.csv"))):
    if not ours(r["Kernel_Name"]):
        continue
    grid = "x".join(r[k] for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
    wg = "x".join(r[k] for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"))
    groups.setdefault((short(r["Kernel_Name"]), grid, wg), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
seen_kernels = {k[0] for k in groups}
for k in meta["expect_kernels"]:
    assert k in seen_kernels, f"expected kernel {k!r} is not in the trace (has: {sorted(seen_kernels)})"
with open(os.path.join(root, "profiles", f"{tag}_per_grid_medians.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid_threads", "workgroup", "launches", "median_us", "min_us", "max_us"])
    for (name, grid, wg), v in sorted(groups.items()):
        w.writerow([name, grid, wg, len(v), round(statistics.median(v), 2), round(min(v), 2), round(max(v), 2)])
per_kernel = {}
for kind, scale in (("fetch", 2.0), ("write", 1.0)):       # gfx950: FETCH_SIZE counts half of a 16-B-per-lane stream (MI355X_MICROARCH.md)
    if "pmc_" + kind not in meta["passes"]:
        continue
    rows = list(csv.DictReader(open(find("pmc_" + kind, "*counter_collection.csv"))))
    names = {short(r["Kernel_Name"]) for r in rows if ours(r["Kernel_Name"])}
    for k in meta["expect_kernels"]:
        assert k in names, f"expected kernel {k!r} is not in the {kind} counters (has: {sorted(names)})"
    seen, vals = {}, {}
    with open(os.path.join(root, "profiles", f"{tag}_pmc_{kind}.csv"), "w", newline="") as f:
        w = None
        for r in rows:
            if not ours(r["Kernel_Name"]):
                continue
            name = short(r["Kernel_Name"])
            vals.setdefault((name, r["Grid_Size"]), []).append(float(r["Counter_Value"]))
            key = (name, r["Grid_Size"])
            seen[key] = seen.get(key, 0) + 1
            if seen[key] > 4:
                continue
            row = {k: r[k] for k in ("Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Counter_Name", "Counter_Value") if k in r}
            row["Kernel_Name"] = name
            if w is None:
                w = csv.DictWriter(f, fieldnames=list(row))
                w.writeheader()
            w.writerow(row)
    for (name, grid), v in vals.items():
        per_kernel.setdefault(f"{name} grid={grid}", {})[kind + "_bytes"] = int(statistics.median(v) * 1024 * scale)
meta["hbm_bytes_per_launch_median"] = {n: dict(v, total=v.get("fetch_bytes", 0) + v.get("write_bytes", 0)) for n, v in sorted(per_kernel.items())}
json.dump(meta, open(os.path.join(root, "profiles", f"{tag}_meta.json"), "w"), indent=1)
if traffic_kernel:
    hit = [(n, v) for n, v in per_kernel.items() if n.startswith(traffic_kernel + " grid=")]
    assert hit and "fetch_bytes" in hit[0][1] and "write_bytes" in hit[0][1], f"no fetch+write counters for {traffic_kernel!r}"
    # the launch the bench line of the stats pass reports (same command in every pass)
    line = meta["passes"]["stats"]["bench_line"] or {}
    roof = line.get("roofline") or {}
    n, v = max(hit, key=lambda kv: kv[1]["fetch_bytes"]) if len(hit) > 1 and "--largest" in sys.argv else hit[0]
    out = {"kernel": traffic_kernel, "git_head": meta["git_head"], "bench_args": meta["bench_args"],
           "scores_src_sha16": meta["source_sha16"]["moc_scores.hip"],
           "algorithmic_bytes_per_launch": roof.get("algorithmic_bytes_per_launch"),
           "hbm_bytes_per_launch": v["fetch_bytes"] + v["write_bytes"], "fetch_bytes_corrected_x2": v["fetch_bytes"], "write_bytes": v["write_bytes"],
           "ratio_to_algorithmic": (round((v["fetch_bytes"] + v["write_bytes"]) / roof["algorithmic_bytes_per_launch"], 4) if roof.get("algorithmic_bytes_per_launch") else None),
           "grid": n.split("grid=")[1], "note": "medians over the capture's launches of that grid; FETCH_SIZE KiB x1024 x2 (gfx950 correction), WRITE_SIZE KiB x1024"}
    if "--as-default" in sys.argv:
        # profiles/traffic.json holds one entry per (kernel, launch size): this capture replaces its own, keeps the others
        path = os.path.join(root, "profiles", "traffic.json")
        try:
            old = json.load(open(path))
            entries = old.get("entries", [old])
        except (OSError, ValueError):
            entries = []
        same = lambda e: (e.get("kernel") == out["kernel"] and out["algorithmic_bytes_per_launch"] and e.get("algorithmic_bytes_per_launch") and
                          abs(e["algorithmic_bytes_per_launch"] - out["algorithmic_bytes_per_launch"]) <= 0.01 * out["algorithmic_bytes_per_launch"])
        entries = [e for e in entries if not same(e)] + [out]
        json.dump({"entries": entries}, open(path, "w"), indent=1)
    else:
        json.dump(out, open(os.path.join(root, "profiles", f"{tag}_traffic.json"), "w"), indent=1)
    print(json.dumps(out))
print("ok", tag)
