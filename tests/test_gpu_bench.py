"""bench.py as the driver runs it, on the GPU box: the one-GPU line carries what the contract asks for (value over
exactly --steps steps, a steady_state block, roofline measured live, traffic null or matched), and `--gpus 2` starts
its own ranks -- rehearsed with both ranks on the one card (MOC_BENCH_ONE_DEVICE=1, gloo): runs x GPUs is `value`, the
exact-sequential mode and minibatch data parallelism extra blocks (their ranks end with bit-identical parameters:
bench.py asserts that itself)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, env=None, timeout=600):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout, env=e)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must carry ONE JSON line, got {len(lines)}"
    return json.loads(lines[0])


def test_one_gpu_line(gpu_device):
    d = _bench("--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "2", "--steady-epochs", "10", "--packed-runs", "2",
               "--batched-runs", "2")
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["unit"] == "meta-steps/s"
    assert d["value"] > 5000 and abs(d["ms_per_step"] * d["value"] - 1000.0) < 1.0
    assert d["config"]["workload"].startswith("NSCLC 2-way 16-shot") and d["dtype"] == "f32" and d["vs_baseline"] is None
    ss = d["steady_state"]
    assert ss["epochs"] == 10 and ss["steps"] == 320 and ss["value"] > d["value"] * 0.6
    roof = d["roofline"]
    assert d["config"]["bag_storage"] == "fp32"                      # the storage pinned to the reference's main(); bf16 is an extra block
    # (the timed launches are look-ahead launches: the ticketed form, off the compute units left to the meta-steps)
    assert roof["bound"] == "hbm" and roof["kernel"] == "scores_stream_kernel<16, false, 1, false, true>" and roof["peak"] == 8000.0
    assert roof["placement"]["compute_units_left_to_the_meta_steps"] == 64
    assert roof["whole_chip"]["kernel"] == "scores_stream_kernel<16, false, 1, false, false>" and roof["whole_chip"]["frac"] > roof["frac"] * 0.9
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and 0.2 < roof["frac"] < 1.0
    assert roof["traffic"] is None or "traffic_source" in roof       # a 20-slide launch matches no committed capture
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["value"] > 10 and cpu["cores"] >= 1 and "sample" in cpu and cpu["eval_slides_per_sec"] > 10
    hb = d["bf16_storage"]
    assert hb["bag_storage"] == "bf16" and hb["value"] > 5000 and "+-0.002" in hb["fidelity"]
    # the driver's 20-step region must see most of the rate of a long run (a spacer pass behind the region, phase A and
    # the mask draws a pass ahead); the margin covers clock and launch jitter on a 0.5-ms region
    assert d["value"] >= 0.6 * ss["value"], (d["value"], ss["value"])
    assert d["eval_slides_per_sec"] > 10000
    br = d["batched_runs"]["runs_2"]                                  # two runs batched in one process (moc_amd.runs)
    assert br["runs"] == 2 and br["value"] > d["value"] and 0.2 < br["score_pass"]["frac"] < 1.0
    cs = d["cached_scores"]                                           # opt-in extra: statistics kept from one score pass
    assert cs["value"] > 5000 and cs["steady_state"] > 0.9 * ss["value"] and "Not `value`" in cs["note"] and cs["batched_runs_8"]["runs"] == 8
    pk = d["packed_runs"]
    assert pk["runs"] == 2 and pk["value"] > 5000 and "vs_one_run" in pk      # two independent runs on the one GPU, timed together


def test_gpus_2_starts_its_own_ranks_and_reports_runs_x_gpus(gpu_device):
    """N > 1: `value` is runs x GPUs (one independent run per rank, nothing exchanged, scaling weak); the exact-sequential
    mode of ONE run over the ranks and minibatch data parallelism are opt-in extra blocks."""
    d = _bench("--gpus", "2", "--steps", "32", "--warmup", "32", "--no-eval", "--steady-epochs", "2", "--dp-exchange", "auto", "--dp-extra",
               "--seq-extra", "--batched-runs", "0",
               env={"MOC_BENCH_ONE_DEVICE": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 100
    assert d["config"]["parallelism"].startswith("runs x GPUs: 2 independent training runs")
    assert "rehearsal" in d and d["ranks_seen"] == 2 and "rccl_version" in d
    sq = d["exact_sequential"]
    assert sq["value"] > 100 and sq["scaling"] == "strong" and "bit-identical" in sq["note"]
    mb = d["minibatch_dp"]
    assert mb["dp_strong"]["value"] > 100 and mb["dp_weak"]["value"] > 100 and mb["dp_strong"]["exchange"] == "p2p"
    assert "NOT within" in mb["note"]
    assert "replicas" not in d                                        # (it IS `value` now)


def test_gpus_2_train_mode_seq_is_still_there(gpu_device):
    d = _bench("--gpus", "2", "--steps", "32", "--warmup", "32", "--no-eval", "--steady-epochs", "2", "--train-mode", "seq", "--batched-runs", "0",
               env={"MOC_BENCH_ONE_DEVICE": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["parallelism"].startswith("seq2: exact-sequential")
    rep = d["replicas"]
    assert rep["value"] > 100 and rep["scaling"] == "weak" and "no collective" in rep["note"]
