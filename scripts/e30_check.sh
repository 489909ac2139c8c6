set -e
d=gpurun_out/$1; mkdir -p $d
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide or gradient_only or cached_statistics" > $d/tests.log 2>&1
python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_graph.py -x -q -m gpu >> $d/tests.log 2>&1
python scripts/diag_stamps_wide.py bf16 > $d/stamps_bf16.txt 2>&1
python bench.py --classes 30 --slides 120 --steps 360 --warmup 120 --steady-epochs 20 --dtype bf16 --no-cpu --no-eval --packed-runs 0 --batched-runs= --no-cached-extra > $d/e30_bf16.json 2>$d/e30_bf16.err
python bench.py --classes 30 --slides 120 --steps 360 --warmup 120 --steady-epochs 20 --no-cpu --no-eval --packed-runs 0 --no-16bit-extra --batched-runs= --no-cached-extra > $d/e30_fp32.json 2>$d/e30_fp32.err
tail -3 $d/tests.log; tail -12 $d/stamps_bf16.txt
