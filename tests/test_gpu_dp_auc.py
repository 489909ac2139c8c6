"""Which multi-GPU training mode keeps the reference's AUC (north_star: within +-0.002)?  Decided here, on the task the
reference's own main() was run on (tests/golden/driver.npz, main_moc.py:611-628), not on speed:

  * exact-sequential (dist.train_seq) is bit-identical to the one-GPU loop (tests/test_gpu_seq.py) and reproduces the
    fixture with every split spread over two ranks (same file) -- it is the default for multi-GPU training;
  * synchronous minibatch data parallelism takes one Adam step per G slides.  Its trajectory is reproduced in ONE
    process by dist.train_minibatch (gradient accumulation) and compared with the same fixture: it leaves the bar at
    G = 2, 4 and 8 (best-val AUC off by 0.05 / 0.10 / 0.16; more tasks, seeds and learning-rate rules in
    profiles/round2_dp_auc_study.jsonl).  The expectation is written as a STRICT xfail: should minibatch-DP ever land
    inside the bar, the suite goes red and the default has to be revisited."""
import numpy as np
import pytest
import torch

import helpers as H
from moc_amd import synth

pytestmark = pytest.mark.gpu


def _run_fixture_task(dev, cid, G, tmp_path):
    from moc_amd import dist as mdist, main_moc as M, run_moc
    g = H.golden("driver")
    _, ntr, nva, nte, C, j, K, rep, seed = [int(v) for v in g["cases"][cid]]
    W, We = synth.make_bank(seed, 512, C)
    M.set_classifier_bank(W.to(dev), We.to(dev))
    loaders = []
    for s_i in range(3):
        sizes = [int(v) for v in g[f"c{cid}_sizes{s_i}"]]
        bags, labels = synth.make_slide_set(seed + 1000 * (s_i + 1), sizes, 512, We, C, confusion=0.47, gain=0.12)
        loaders.append(M.ResidentBags(bags, labels, dev, repeat_num=rep if s_i == 0 else None))
    args = run_moc.get_args(["--topj", str(j), "--topk", str(K), "--shot", "4", "--fold", "0", "--disable_tqdm",
                             "--result_dir", str(tmp_path / f"res{G}")])
    args.n_classes, args.check_zeroshot = C, False
    torch.manual_seed(seed)
    model = M.senet(512, 4).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    orig = run_moc._train
    if G > 1:
        run_moc._train = lambda m, l, o, d, a: mdist.train_minibatch(m, l, o, d, a, G)
    try:
        torch.manual_seed(seed + 1)
        res = run_moc.main(args, model, opt, *loaders, dev)
    finally:
        run_moc._train = orig
    return res, g[f"c{cid}_result"]


def test_one_step_per_slide_reproduces_the_reference_run(gpu_device, tmp_path):
    res, exp = _run_fixture_task(gpu_device, 1, 1, tmp_path)
    assert abs(res["best_val"] - exp[0]) < 2e-3 and abs(res["test_at_best_val"] - exp[1]) < 2e-3 and res["best_epoch"] == int(exp[3])


@pytest.mark.xfail(strict=True, reason="minibatch data parallelism changes the optimisation trajectory: AUC outside +-0.002 "
                                       "of the sequential reference (exact-sequential train_seq is the multi-GPU default)")
@pytest.mark.parametrize("G", [2, 4, 8])
def test_minibatch_dp_keeps_the_auc_of_the_sequential_reference(gpu_device, tmp_path, G):
    res, exp = _run_fixture_task(gpu_device, 1, G, tmp_path)
    print(f"G={G}: best_val {res['best_val']:.4f} vs {exp[0]:.4f}, test at best val {res['test_at_best_val']:.4f} vs {exp[1]:.4f}")
    assert abs(res["best_val"] - exp[0]) < 2e-3 and abs(res["test_at_best_val"] - exp[1]) < 2e-3
