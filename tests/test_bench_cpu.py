"""bench.py's launcher, without a GPU: `python bench.py --gpus N` must start its own N ranks before any GPU
call (the driver invokes exactly that form), every rank must get as far as the GPU assert, and the parent must
hand back a non-zero exit code."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_2_self_launches_two_ranks_up_to_the_gpu_assert():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert r.stderr.count("AssertionError: bench.py needs a GPU") == 2, r.stderr[-2000:]
    assert r.stdout.strip() == ""                       # no JSON line without a measurement


def test_workload_label_follows_the_shape():
    sys.path.insert(0, ROOT)
    import types
    import bench
    a = types.SimpleNamespace(classes=2, slides=32, dim=512, patches=15000, topj=400, topk=10, lognormal=False)
    assert bench.workload_name(a).startswith("NSCLC 2-way 16-shot")
    a.classes, a.slides = 30, 120
    assert bench.workload_name(a).startswith("EBRAINS-30")
    a.classes, a.slides, a.dim = 64, 64, 1024
    assert bench.workload_name(a).startswith("synthetic 64-way") and "NSCLC" not in bench.workload_name(a)


def test_traffic_is_null_unless_captured_for_this_kernel_source_and_launch():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.measured_traffic("scores_stream_kernel<16, true, 1, false>", 1.0) is None      # wrong launch size
    assert bench.measured_traffic("no_such_kernel", 245760000.0) is None
