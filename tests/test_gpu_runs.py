"""Batched runs: R independent meta-learners stepped in lockstep by one forward launch and one step
launch per meta-step (moc_train_steps_runs; moc_amd.runs).  Each run is the exact recurrence of main_moc.train
(/root/reference main_moc.py:378-410 per process of scripts/moc_train.sh:11-31): parameters, both Adam moments and the
losses of every pass are BIT-identical to training the run alone from the same mask stream."""
import numpy as np
import pytest
import torch

import helpers as H
from moc_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(gpu_device):
    return gpu_device


def _task(run, n, C, D, lo, hi, dtype):
    W, We = synth.make_bank(31 + C, D, C)                       # the runs share the classifier bank
    g = np.random.default_rng(500 + run)
    sizes = [int(v) for v in g.integers(lo, hi, size=n)]
    bags, labels = synth.make_slide_set(9000 + 37 * run, sizes, D, We, C)
    return W, We, [b.to(dtype) for b in bags], labels


def _alone(dev, run, n, C, D, lo, hi, dtype, j, K, epochs, discard):
    from moc_amd import main_moc as M
    W, We, bags, labels = _task(run, n, C, D, lo, hi, dtype)
    M.set_classifier_bank(W.to(dev), We.to(dev))
    torch.manual_seed(100 + run)
    model = M.senet(D, 4).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    res = M.ResidentBags(bags, labels, dev)
    args = H.make_args(C, j, K, discard)
    torch.manual_seed(7000 + run)                               # the run's mask stream
    losses = []
    for _ in range(epochs):
        M.train(model, res, opt, dev, args)
        torch.cuda.synchronize()
        losses.append(M.train.last[0].meta_ws()[0]["loss"].cpu().numpy().copy())
    return H.flat_params(model), H.flat_state(opt, "exp_avg"), H.flat_state(opt, "exp_avg_sq"), losses


@pytest.mark.parametrize("R,n,C,D,dtype,j,K,discard", [
    (2, 5, 2, 512, torch.float32, 400, 10, ()),
    (5, 4, 2, 512, torch.float32, 300, 10, ()),
    (8, 3, 2, 512, torch.float32, 400, 10, ("delta_diff",)),
    (3, 4, 3, 512, torch.bfloat16, 200, 10, ()),
    (8, 16, 2, 256, torch.float32, 100, 5, ()),                 # 128 slides in one phase A: the grouped selector
    (2, 3, 2, 1024, torch.float16, 150, 1, ()),
    # wide banks (outside the tile-record step: moc_train_runs_mode == 2): every run a chain of its own on its own stream
    (3, 3, 30, 512, torch.bfloat16, 40, 10, ()),                # wide step, pooling inside the kernel
    (2, 2, 30, 512, torch.float32, 400, 10, ()),                # ... behind topk_mean_kernel (more than 48 k / C scores)
    (4, 2, 20, 1024, torch.float16, 60, 5, ("bottomk",)),
])
def test_every_run_of_a_batch_is_the_run_alone_bit_for_bit(dev, R, n, C, D, dtype, j, K, discard):
    from moc_amd import main_moc as M
    lo, hi = (600, 1500) if n >= 16 else (2200, 4200)
    epochs = 3
    alone = [_alone(dev, r, n, C, D, lo, hi, dtype, j, K, epochs, discard) for r in range(R)]
    models, opts, splits, gens = [], [], [], []
    for r in range(R):
        W, We, bags, labels = _task(r, n, C, D, lo, hi, dtype)
        M.set_classifier_bank(W.to(dev), We.to(dev))
        torch.manual_seed(100 + r)
        model = M.senet(D, 4).to(dev)
        models.append(model)
        opts.append(torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4))
        splits.append(M.ResidentBags(bags, labels, dev))
        g = torch.Generator()
        g.manual_seed(7000 + r)
        gens.append(g)
    args = H.make_args(C, j, K, discard)
    losses = []
    for _ in range(epochs):
        rs = M.train_runs(models, splits, opts, dev, args, generators=gens)
        torch.cuda.synchronize()
        losses.append(rs.losses().cpu().numpy().copy())
    assert rs.mode == (1 if C <= 16 else 2) and len(rs.groups) == (1 if C <= 16 else R)
    for r in range(R):
        p, m, v, ls = alone[r]
        np.testing.assert_array_equal(H.flat_params(models[r]), p, err_msg=f"run {r}: parameters")
        np.testing.assert_array_equal(H.flat_state(opts[r], "exp_avg"), m, err_msg=f"run {r}: exp_avg")
        np.testing.assert_array_equal(H.flat_state(opts[r], "exp_avg_sq"), v, err_msg=f"run {r}: exp_avg_sq")
        for e in range(epochs):
            np.testing.assert_array_equal(losses[e][r], ls[e], err_msg=f"run {r} pass {e}: losses")
        assert all(int(float(opts[r].state[q]["step"])) == epochs * n for q in models[r].parameters())
    # the models are ordinary modules afterwards: evaluation, state_dict
    if C <= n:                                                # (a multi-class AUC wants every class among the slides: sklearn, as in the reference)
        ev = M.evaluation(models[0], splits[0], dev, args)
        assert np.isfinite(ev["loss"]) and 0.0 <= ev["acc"] <= 1.0
    sd = opts[1].state_dict()
    assert len(sd["state"]) == 4 and tuple(models[1].state_dict()["model.0.weight"].shape) == (64, D)


def test_runs_must_agree_on_what_they_share(dev):
    from moc_amd import main_moc as M
    W, We, bags, labels = _task(0, 3, 2, 512, 2200, 4200, torch.float32)
    M.set_classifier_bank(W.to(dev), We.to(dev))
    models = [M.senet(512, 4).to(dev) for _ in range(2)]
    opts = [torch.optim.Adam(models[0].parameters(), lr=1e-3), torch.optim.Adam(models[1].parameters(), lr=2e-3)]
    splits = [M.ResidentBags(bags, labels, dev) for _ in range(2)]
    with pytest.raises(AssertionError, match="share Adam"):
        M.train_runs(models, splits, opts, dev, H.make_args(2, 100, 10))
    opts[1] = torch.optim.Adam(models[1].parameters(), lr=1e-3)
    with pytest.raises(AssertionError, match="private CPU generator"):
        M.train_runs(models, splits, opts, dev, H.make_args(2, 100, 10), generators=[torch.default_generator, torch.Generator()])


def test_folds_in_one_process_reproduce_each_fold_alone(dev, tmp_path):
    """`run_moc --folds 0,1,2`: the reference's launcher (scripts/moc_train.sh:11-31) in one process.  Every fold's result
    file holds the numbers of `--fold F` alone, its best checkpoint the same bits."""
    import json, os
    from moc_amd import run_moc
    common = ["--synthetic", "6", "--shot", "2", "--seed", "3", "--epochs", "4", "--topj", "100", "--topk", "10", "--disable_tqdm"]
    together = run_moc.cli(common + ["--folds", "0,1,2", "--result_dir", str(tmp_path / "together")])
    assert len(together) == 3
    for fold in range(3):
        alone = run_moc.cli(common + ["--fold", str(fold), "--result_dir", str(tmp_path / "alone")])
        a = json.load(open(tmp_path / "alone" / f"best_results_shot_2_fold_{fold}.json"))
        b = json.load(open(tmp_path / "together" / f"best_results_shot_2_fold_{fold}.json"))
        for k in ("zero_shot_train", "zero_shot_val", "zero_shot_test", "best_val", "test_at_best_val", "test_acc_at_best_val", "best_epoch"):
            assert a[k] == b[k] == alone[k], (fold, k, a[k], b[k])
        sa = torch.load(tmp_path / "alone" / f"best_model_shot_2_fold_{fold}.pt", map_location="cpu")
        sb = torch.load(tmp_path / "together" / f"best_model_shot_2_fold_{fold}.pt", map_location="cpu")
        assert sa.keys() == sb.keys() and all(torch.equal(sa[k], sb[k]) for k in sa)
        assert os.path.exists(tmp_path / "together" / f"zs_results_shot_2_fold_{fold}.json")
