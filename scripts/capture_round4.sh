#!/bin/bash
# Every rocprofv3 capture of round 4 in one go (run ON the GPU box through gpurun); then scripts/profile_summaries.py <tag> here.
#   HEAD=$(git rev-parse --short HEAD) gpurun -- 'HEAD=... bash scripts/capture_round4.sh [tags...]'
set -e
H=${HEAD:?set HEAD to git rev-parse --short HEAD (the GPU box has no .git)}
TAGS=("$@")
sel() { [ ${#TAGS[@]} -eq 0 ] && return 0; for t in "${TAGS[@]}"; do [ "$t" = "$1" ] && return 0; done; return 1; }
if sel default; then python scripts/profile_capture.py --tag round4_default --head $H --expect "scores_stream_kernel<16, false, 1, false, true>" --expect "pool_w1_step_tiles_kernel<12, TileStepArgs>" -- --steps 320 --warmup 32 --no-cpu --no-eval --steady-epochs 5 --packed-runs 0 --batched-runs 0 --no-cached-extra --no-16bit-extra; fi
if sel s20; then python scripts/profile_capture.py --tag round4_s20 --head $H --expect "scores_stream_kernel<16, false, 1, false, true>" -- --slides 20 --steps 200 --warmup 20 --no-cpu --no-eval --steady-epochs 5 --packed-runs 0 --batched-runs 0 --no-cached-extra --no-16bit-extra; fi
if sel fp32eval; then python scripts/profile_capture.py --tag round4_fp32eval --head $H --no-pmc -- --steps 320 --warmup 32 --no-cpu --steady-epochs 5 --packed-runs 0 --batched-runs 0 --no-16bit-extra; fi
if sel e30; then python scripts/profile_capture.py --tag round4_e30 --head $H --expect "scores_stream_kernel<16, true, 3, false, false>" -- --classes 30 --slides 120 --steps 360 --warmup 120 --steady-epochs 5 --dtype bf16 --no-cpu --no-eval --packed-runs 0 --batched-runs 0; fi
if sel w64; then python scripts/profile_capture.py --tag round4_w64 --head $H --expect "scores_wide_ring_kernel<5, true>" -- --classes 64 --dim 1024 --patches 50000 --slides 64 --dtype fp16 --steps 192 --warmup 64 --steady-epochs 3 --no-cpu --no-eval --packed-runs 0 --batched-runs 0; fi
if sel e30eval; then python scripts/profile_capture.py --tag round4_e30eval --head $H --no-pmc -- --classes 30 --slides 120 --steps 240 --warmup 120 --steady-epochs 2 --dtype bf16 --no-cpu --packed-runs 0 --batched-runs 0; fi
if sel runs8; then python scripts/profile_capture.py --tag round4_runs8 --head $H --no-pmc --expect "pool_w1_step_tiles_kernel<12, TileStepArgsRuns>" -- --steps 32 --warmup 32 --no-cpu --no-eval --packed-runs 0 --no-16bit-extra --batched-runs 8 --no-cached-extra --steady-epochs 2; fi
