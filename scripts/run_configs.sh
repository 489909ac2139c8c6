#!/bin/bash
# The lines of profiles/<round>_configs.jsonl: python bench.py --no-cpu plus the flags of each configuration.
#   bash scripts/run_configs.sh gpurun_out/configs.jsonl
out=${1:-gpurun_out/configs.jsonl}
: > "$out"
run() { timeout -k 10 400 python bench.py --no-cpu --packed-runs 0 "$@" | tail -1 >> "$out" || exit 1; }
run
run --dtype fp16
run --dtype fp32
run --classes 3 --slides 48
run --classes 30 --slides 120 --steps 360 --warmup 120 --steady-epochs 5
run --classes 64 --dim 1024 --patches 50000 --slides 64 --dtype fp16 --steps 192 --warmup 64 --steady-epochs 3
