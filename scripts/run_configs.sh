#!/bin/bash
# The lines of profiles/<round>_configs.jsonl: python bench.py with the flags of each BASELINE.json
# configuration, CPU baseline leg included (train meta-steps/s and eval slides/s of the oracle on the box's host cores).
#   bash scripts/run_configs.sh gpurun_out/configs.jsonl
out=${1:-gpurun_out/configs.jsonl}
: > "$out"
run() { timeout -k 10 500 python bench.py --packed-runs 0 --cpu-seconds 8 ${BATCHED:---batched-runs 0 --no-cached-extra} "$@" | tail -1 >> "$out" || exit 1; echo "done: $*" >&2; }
# cfg 1: NSCLC 2-way 1-shot, 256 patches per bag (2 train slides = repeat_num shot x C)
run --slides 2 --patches 256 --eval-slides 49 --steps 200 --warmup 20 --steady-epochs 200
# cfg 2: NSCLC 2-way 16-shot, full bags: the headline (fp32 storage), then its 16-bit storage variants
BATCHED=" " run            # (the headline line keeps its batched_runs block: --batched-runs 8,16)
run --dtype bf16
run --dtype fp16
# cfg 3: RCC 3-way 16-shot (48 visits per epoch)
run --classes 3 --slides 48
# cfg 4: EBRAINS-30 30-way 4-shot (fp32 and bf16 storage)
run --classes 30 --slides 120 --steps 360 --warmup 120 --steady-epochs 5
run --classes 30 --slides 120 --steps 360 --warmup 120 --steady-epochs 5 --dtype bf16
# cfg 5: synthetic 64-way, 50k x 1024, fp16 storage
run --classes 64 --dim 1024 --patches 50000 --slides 64 --dtype fp16 --steps 192 --warmup 64 --steady-epochs 3 --eval-slides 64
