#!/bin/bash
# ON the GPU box: scripts/capture_round4.sh one tag at a time, each turned into its committed summaries right there
# (scripts/profile_summaries.py), the summaries copied to gpurun_out/p4/ and the raw traces deleted (gpurun carries 64 MiB back).
#   HEAD=$(git rev-parse --short HEAD) gpurun -- 'HEAD=... bash scripts/capture_and_summarise.sh [tags...]'
set -e
mkdir -p gpurun_out/p4
tags=("$@"); [ ${#tags[@]} -eq 0 ] && tags=(default s20 fp32eval e30 w64 e30eval runs8)
for t in "${tags[@]}"; do
  bash scripts/capture_round4.sh $t
  case $t in
    default|s20) extra=(--traffic "scores_stream_kernel<16, false, 1, false, true>" --as-default);;
    e30) extra=(--traffic "scores_stream_kernel<16, true, 3, false, false>");;
    w64) extra=(--traffic "scores_wide_ring_kernel<5, true>");;
    *) extra=();;
  esac
  python scripts/profile_summaries.py round4_$t "${extra[@]}"
  cp profiles/round4_${t}_* gpurun_out/p4/
  { [ "$t" = default ] || [ "$t" = s20 ]; } && cp profiles/traffic.json gpurun_out/p4/
  rm -rf gpurun_out/round4_$t/stats gpurun_out/round4_$t/pmc_fetch gpurun_out/round4_$t/pmc_write
done
ls -la gpurun_out/p4 | tail -40
