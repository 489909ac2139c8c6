#!/usr/bin/env python3
"""Turn one round's rocprofv3 outputs under gpurun_out/ into the summaries committed under profiles/:

    python scripts/profile_summaries.py <tag>        # e.g. round1_h

  gpurun_out/prof_<tag>/*_kernel_stats.csv  ->  profiles/<tag>_kernel_stats.csv      (as is)
  gpurun_out/prof_<tag>/*_kernel_trace.csv  ->  profiles/<tag>_per_grid_medians.csv  (median/min/max per kernel and grid)
  gpurun_out/pmc_{fetch,write}/*/*_counter_collection.csv -> profiles/<tag>_pmc_{fetch,write}.csv (library kernels, first 4 dispatches per kernel/grid)
"""
import csv, glob, os, shutil, statistics, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
short = tag.split("_", 1)[1] if "_" in tag else tag
prof = os.path.join(root, "gpurun_out", f"prof_{'r1_' + short if tag.startswith('round1_') else tag}")
stats = glob.glob(os.path.join(prof, "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"))
trace = glob.glob(os.path.join(prof, "*_kernel_trace.csv"))[0]
groups = {}
for r in csv.DictReader(open(trace)):
    n = r["Kernel_Name"]
    if "anonymous namespace" not in n or "at::native" in n:
        continue
    name = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    grid = "x".join(r[k] for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
    wg = "x".join(r[k] for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"))
    groups.setdefault((name, grid, wg), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(os.path.join(root, "profiles", f"{tag}_per_grid_medians.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid_threads", "workgroup", "launches", "median_us", "min_us", "max_us"])
    for (name, grid, wg), v in sorted(groups.items()):
        w.writerow([name, grid, wg, len(v), round(statistics.median(v), 2), round(min(v), 2), round(max(v), 2)])
for kind in ("fetch", "write"):
    src = glob.glob(os.path.join(root, f"gpurun_out/pmc_{kind}/*/*_counter_collection.csv"))
    if not src:
        continue
    seen = {}
    with open(os.path.join(root, "profiles", f"{tag}_pmc_{kind}.csv"), "w", newline="") as f:
        w = None
        for r in csv.DictReader(open(src[0])):
            n = r["Kernel_Name"]
            if "anonymous namespace" not in n or "at::native" in n:
                continue
            key = (n, r["Grid_Size"])
            seen[key] = seen.get(key, 0) + 1
            if seen[key] > 4:
                continue
            row = {k: r[k] for k in ("Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Counter_Name", "Counter_Value") if k in r}
            row["Kernel_Name"] = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if w is None:
                w = csv.DictWriter(f, fieldnames=list(row))
                w.writeheader()
            w.writerow(row)
print("ok")
