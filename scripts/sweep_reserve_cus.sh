#!/bin/bash
# Steady-state and timed-region rates of the default workload with the look-ahead score pass kept off N compute units
# (MOC_RESERVE_CUS; 0 = the static walk over the whole chip).
# usage: [BENCH_EXTRA='--dtype bf16'] scripts/sweep_reserve_cus.sh OUT.jsonl N [N ...]
out=$1; shift
extra="${BENCH_EXTRA:-}"
: > "$out"
for n in "$@"; do
  for rep in 1 2; do
    export MOC_RESERVE_CUS=$n
    python bench.py --no-cpu --no-eval --no-16bit-extra --packed-runs 0 --steps 1600 --warmup 160 $extra 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(json.dumps({'reserve': '$n', 'value': d['value'], 'steady': d['steady_state']['value'], 'score_live_GBs': r['achieved'], 'score_us': r['avg_launch_us'], 'alone_GBs': r['alone']['achieved']}))" >> "$out" || exit 1
  done
done
cat "$out"
