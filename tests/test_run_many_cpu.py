"""moc_amd.run_many (the reference's scripts/moc_train.sh as a job queue): every (shot, fold) runs once, pinned to a GPU
through HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES, never more than --runs-per-gpu side by side on one GPU, results and
logs laid out as the reference's script lays them out, failures reported in the exit code.  The runs themselves are
stand-in programs here (no GPU)."""
import json
import os
import sys

from moc_amd import run_many

FAKE = r'''
import json, os, sys, time
a = sys.argv[1:]
fold, shot, out = a[a.index("--fold") + 1], a[a.index("--shot") + 1], a[a.index("--result_dir") + 1]
t0 = time.time()
time.sleep(0.5)
print("hello from", fold, shot)
rec = dict(fold=int(fold), shot=int(shot), gpu=os.environ["HIP_VISIBLE_DEVICES"], cuda=os.environ["CUDA_VISIBLE_DEVICES"],
           rocr=os.environ.get("ROCR_VISIBLE_DEVICES"),
           t0=t0, t1=time.time(), rest=a, out=out)
open(os.path.join(os.environ["FAKE_LOG_DIR"], f"{shot}_{fold}.json"), "w").write(json.dumps(rec))
sys.exit(3 if (fold, shot) == ("1", "2") and os.environ.get("FAKE_FAIL") else 0)
'''


def _run(tmp_path, monkeypatch, fail=False):
    prog = tmp_path / "fake_run.py"
    prog.write_text(FAKE)
    logs = tmp_path / "records"
    logs.mkdir()
    monkeypatch.setenv("FAKE_LOG_DIR", str(logs))
    if fail:
        monkeypatch.setenv("FAKE_FAIL", "1")
    rc = run_many.main(["--folds", "0", "1", "2", "--shots", "2", "4", "--gpus", "0", "1", "--runs-per-gpu", "2", "--seed", "7",
                        "--result_dir", str(tmp_path / "res"), "--runner", sys.executable, str(prog), "--", "--topj", "400", "--disable_tqdm"])
    recs = [json.loads((logs / f).read_text()) for f in sorted(os.listdir(logs))]
    return rc, recs


def test_every_job_runs_once_within_the_per_gpu_cap(tmp_path, monkeypatch):
    rc, recs = _run(tmp_path, monkeypatch)
    assert rc == 0
    assert sorted((r["shot"], r["fold"]) for r in recs) == [(2, 0), (2, 1), (2, 2), (4, 0), (4, 1), (4, 2)]
    for r in recs:
        assert r["gpu"] in ("0", "1") and r["cuda"] == r["gpu"]
        assert r["out"].endswith(f"{r['shot']}_shot") and "--topj" in r["rest"] and "--disable_tqdm" in r["rest"]
        assert r["rest"][r["rest"].index("--seed") + 1] == str(7 + 100 * r["shot"] + r["fold"])
        log = os.path.join(str(tmp_path / "res"), f"{r['shot']}_shot", f"fold_{r['fold']}_shot_{r['shot']}_output.txt")
        assert "hello from" in open(log).read()
    for g in ("0", "1"):                                     # never more than two runs alive on one GPU at any instant
        mine = [r for r in recs if r["gpu"] == g]
        for r in mine:
            alive = sum(1 for o in mine if o["t0"] < r["t1"] and r["t0"] < o["t1"])
            assert alive <= 2, (g, alive)
    assert {r["gpu"] for r in recs} == {"0", "1"}            # and both GPUs are used


def test_a_failed_run_is_reported(tmp_path, monkeypatch, capsys):
    rc, recs = _run(tmp_path, monkeypatch, fail=True)
    assert rc == 1 and len(recs) == 6
    assert "FAILED shot 2 fold 1 (rc 3)" in capsys.readouterr().out


def test_dry_run_prints_the_commands(tmp_path, capsys):
    rc = run_many.main(["--folds", "0", "1", "--shots", "16", "--gpus", "3", "--dry-run", "--result_dir", str(tmp_path), "--", "--dataset", "rcc"])
    out = capsys.readouterr().out
    assert rc == 0 and out.count("moc_amd.run_moc") == 2 and "--shot 16" in out and "--dataset rcc" in out and "gpu 3:" in out


def test_children_stay_inside_the_parents_device_mask(tmp_path, monkeypatch, capsys):
    """A scheduler hands this process GPUs 4 and 5 (HIP_VISIBLE_DEVICES=4,5, possibly under a ROCR mask): the children
    must be pinned to 4 and 5 -- not to indices 0 and 1 of the node -- and the ROCR mask must survive."""
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4,5")
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0,1,2,3,4,5")
    rc = run_many.main(["--folds", "0", "1", "--shots", "16", "--dry-run", "--result_dir", str(tmp_path), "--", "--dataset", "rcc"])
    out = capsys.readouterr().out
    assert rc == 0 and "gpu 4:" in out and "gpu 5:" in out and "gpu 0:" not in out
    rc, recs = _run(tmp_path, monkeypatch)                    # --gpus 0 1: positions in the mask
    assert rc == 0 and {r["gpu"] for r in recs} == {"4", "5"} and all(r["rocr"] == "0,1,2,3,4,5" for r in recs)
