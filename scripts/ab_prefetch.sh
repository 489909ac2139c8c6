# diagnostic: next-slide prefetch of the tile-record step on (default) / off (MOC_STEP_PREFETCH=0)
mkdir -p gpurun_out/s5v
for v in 0 1 0 1; do
  echo "prefetch $v"; MOC_STEP_PREFETCH=$v python scripts/diag_mall.py 2>&1 | grep "between" | head -2
  MOC_STEP_PREFETCH=$v python bench.py --steps 20 --warmup 5 --no-cpu --no-eval --packed-runs 0 --no-16bit-extra --batched-runs 8 --no-cached-extra --steady-epochs 800 > gpurun_out/s5v/p$v.json 2> gpurun_out/s5v/p$v.err
  python3 -c "
import json
j=json.loads(open('gpurun_out/s5v/p$v.json').read().strip().splitlines()[-1]); print('  bench', j['value'], j['steady_state']['value'], j['batched_runs']['runs_8']['value'])"
done
