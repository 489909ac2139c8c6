#!/bin/bash
# Every rocprofv3 capture of round 3 in one go (run ON the GPU box through gpurun); then scripts/profile_summaries.py <tag> here.
#   HEAD=$(git rev-parse --short HEAD) gpurun -- 'HEAD=... bash scripts/capture_round3.sh'
set -e
python scripts/profile_capture.py --tag round3_default --head ${HEAD:?set HEAD to git rev-parse --short HEAD (the GPU box has no .git)} --expect "scores_stream_kernel<16, false, 1, false, true>" -- --steps 320 --warmup 32 --no-cpu --no-eval --steady-epochs 5 --packed-runs 0 --no-16bit-extra
python scripts/profile_capture.py --tag round3_s20 --head ${HEAD:?set HEAD to git rev-parse --short HEAD (the GPU box has no .git)} --expect "scores_stream_kernel<16, false, 1, false, true>" -- --slides 20 --steps 200 --warmup 20 --no-cpu --no-eval --steady-epochs 5 --packed-runs 0 --no-16bit-extra
python scripts/profile_capture.py --tag round3_fp32eval --head ${HEAD:?set HEAD to git rev-parse --short HEAD (the GPU box has no .git)} --no-pmc --expect "scores_stream_kernel<16, false, 1, false, false>" -- --no-cpu --packed-runs 0 --no-16bit-extra
python scripts/profile_capture.py --tag round3_e30 --head ${HEAD:?set HEAD to git rev-parse --short HEAD (the GPU box has no .git)} --expect "scores_stream_kernel<16, true, 3, false, false>" -- --classes 30 --slides 120 --steps 360 --warmup 120 --steady-epochs 5 --dtype bf16 --no-cpu --no-eval --packed-runs 0
python scripts/profile_capture.py --tag round3_w64 --head ${HEAD:?set HEAD to git rev-parse --short HEAD (the GPU box has no .git)} --expect "scores_wide_ring_kernel<5, true>" -- --classes 64 --dim 1024 --patches 50000 --slides 64 --dtype fp16 --steps 192 --warmup 64 --steady-epochs 3 --no-cpu --no-eval --packed-runs 0
python scripts/profile_capture.py --tag round3_e30eval --head ${HEAD:?set HEAD to git rev-parse --short HEAD (the GPU box has no .git)} --no-pmc -- --classes 30 --slides 120 --steps 240 --warmup 120 --steady-epochs 2 --dtype bf16 --no-cpu --packed-runs 0
