#!/usr/bin/env python3
"""Derive profiles/traffic.json (HBM bytes per score-pass launch) from two rocprofv3 PMC passes:

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 320 --warmup 32 --no-cpu --no-eval
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...   (separate pass: TCC slots)

MI355X_MICROARCH.md (HBM): counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of the
bytes of a 16-B-per-lane streaming read, WRITE_SIZE is exact for streaming stores."""
import csv, glob, json, os, statistics, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
per_kernel = {}
for tag, scale in (("fetch", 2.0), ("write", 1.0)):
    f = glob.glob(os.path.join(root, f"gpurun_out/pmc_{tag}/runc/*_counter_collection.csv"))[0]
    groups = {}
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "anonymous namespace" in n and "at::native" not in n:
            name = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            groups.setdefault((name, r["Grid_Size"]), []).append(float(r["Counter_Value"]))
    for (name, grid), v in groups.items():
        per_kernel.setdefault(f"{name} grid={grid}", {})[tag + "_bytes"] = statistics.median(v) * 1024 * scale
k = [v for n, v in per_kernel.items() if n.startswith("scores_stream_kernel")][0]
out = {"dtype": "bf16", "slides": 32, "patches": 15000, "kernel": "scores_stream_kernel<16, true, 1, false>",
       "hbm_bytes_per_launch": int(k["fetch_bytes"] + k["write_bytes"]),
       "fetch_bytes_corrected_x2": int(k["fetch_bytes"]), "write_bytes": int(k["write_bytes"]),
       "note": "medians over 11 launches; FETCH_SIZE KiB x1024 x2 (gfx950 correction), WRITE_SIZE KiB x1024",
       "per_kernel": {n: {a: int(b) for a, b in v.items()} for n, v in sorted(per_kernel.items())}}
json.dump(out, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps({k2: out[k2] for k2 in ("hbm_bytes_per_launch", "fetch_bytes_corrected_x2", "write_bytes")}))
