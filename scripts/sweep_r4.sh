out=gpurun_out/r4g/sweep.jsonl
mkdir -p gpurun_out/r4g; : > $out
for cfg in "64 0" "64 384" "64 256" "64 192" "64 128" "64 96" "64 64" "32 128" "16 128" "8 128" "32 192" "8 192"; do
    set -- $cfg
    export MOC_RESERVE_CUS=$1 MOC_LOOKAHEAD_WGS=$2
    python bench.py --no-cpu --no-eval --no-16bit-extra --packed-runs 0 --batched-runs 0 --no-cached-extra --steps 640 --warmup 64 --steady-epochs 40 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(json.dumps({'reserve': '$1', 'wgs': '$2', 'value': d['value'], 'steady': d['steady_state']['value'], 'score_live_GBs': r['achieved'], 'score_us': r['avg_launch_us']}))" >> $out || exit 1
done
cat $out
