"""Rows f1 + f2 on the GPU: moc_amd.run_moc.main() -- zero-shot evals, 25 epochs, best-val bookkeeping,
result files -- on slides read from disk through moc_amd.datasets, against what the reference's own
main() produced for the same task on the CPU (tests/golden/driver.npz)."""
import json
import os

import numpy as np
import pandas as pd
import pytest
import torch

import helpers as H
from moc_amd import synth

pytestmark = pytest.mark.gpu


def _task_on_disk(root, cid, g):
    """Write the fixture's synthetic task as dataset csv + split csv + bag files; returns (C, W, We, labels)."""
    from moc_amd import datasets as DS
    _, ntr, nva, nte, C, j, K, rep, seed = [int(v) for v in g["cases"][cid]]
    W, We = synth.make_bank(seed, 512, C)
    names = [f"CLS{c}" for c in range(C)]
    rows, split, data = [], {}, os.path.join(root, "data", "t", "merge_features_conch")
    for s_i, (key, n) in enumerate((("train", ntr), ("val", nva), ("test", nte))):
        sizes = [int(v) for v in g[f"c{cid}_sizes{s_i}"]]
        bags, labels = synth.make_slide_set(seed + 1000 * (s_i + 1), sizes, 512, We, C, confusion=0.47, gain=0.12)
        ids = []
        for i, (b, y) in enumerate(zip(bags, labels)):
            sid = f"{key}_{i:03d}"
            DS.write_bag(data, sid, b, fmt="pt" if i % 2 else "npy")
            rows.append((f"p_{sid}", sid, names[y]))
            ids.append(sid)
        split[key] = pd.Series(ids)
    os.makedirs(os.path.join(root, "dataset_csv"), exist_ok=True)
    pd.DataFrame(rows, columns=["case_id", "slide_id", "label"]).to_csv(os.path.join(root, "dataset_csv", "t.csv"), index=False)
    os.makedirs(os.path.join(root, "splits"), exist_ok=True)
    pd.DataFrame(split).to_csv(os.path.join(root, "splits", "splits_0.csv"))
    return (C, j, K, rep, seed, W, We, {n: i for i, n in enumerate(names)}, data)


@pytest.mark.parametrize("cid", [0, 1])
def test_main_matches_reference_run(gpu_device, tmp_path, cid):
    from moc_amd import datasets as DS, main_moc as M, run_moc
    g = H.golden("driver")
    C, j, K, rep, seed, W, We, label_map, data = _task_on_disk(str(tmp_path), cid, g)
    dev = gpu_device
    M.set_classifier_bank(W.to(dev), We.to(dev))
    ds = DS.Generic_MIL_Dataset(csv_path=str(tmp_path / "dataset_csv" / "t.csv"), data_dir=data, print_info=False,
                                label_dict=label_map)
    tr, va, te = ds.return_splits(from_id=False, csv_path=str(tmp_path / "splits" / "splits_0.csv"), repeat_num=rep)
    loaders = [DS.to_resident(sp, dev) for sp in (tr, va, te)]
    assert len(loaders[0]) == rep
    args = run_moc.get_args(["--topj", str(j), "--topk", str(K), "--shot", "4", "--fold", "0", "--disable_tqdm",
                             "--result_dir", str(tmp_path / "res")])
    args.n_classes = C
    torch.manual_seed(seed)
    model = M.senet(512, 4).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    val_aucs, orig = [], M.evaluation

    def logged(model_, loader, device, a):
        r = orig(model_, loader, device, a)
        if loader is loaders[1]:
            val_aucs.append(r["auc"])
        return r
    M.evaluation = logged
    try:
        torch.manual_seed(seed + 1)                    # the reference run seeded here, before main()
        res = run_moc.main(args, model, opt, *loaders, dev)
    finally:
        M.evaluation = orig
    exp = g[f"c{cid}_result"]
    np.testing.assert_allclose(val_aucs, g[f"c{cid}_val_auc"], atol=2e-3)          # AUC within +-0.002 every epoch
    assert abs(res["best_val"] - exp[0]) < 2e-3 and abs(res["test_at_best_val"] - exp[1]) < 2e-3
    assert abs(res["test_acc_at_best_val"] - exp[2]) < 1e-9 and res["best_epoch"] == int(exp[3])
    on_disk = json.load(open(tmp_path / "res" / "best_results_shot_4_fold_0.json"))
    assert set(on_disk) == {"zero_shot_train", "zero_shot_val", "zero_shot_test", "best_val", "test_at_best_val",
                            "test_acc_at_best_val", "best_epoch", "best_model_path"}
    zs = json.load(open(tmp_path / "res" / "zs_results_shot_4_fold_0.json"))
    got_zs = np.array([[zs[k]["loss"], zs[k]["acc"], zs[k]["auc"]] for k in ("zs_train", "zs_val", "zs_test")])
    np.testing.assert_allclose(got_zs, g[f"c{cid}_zs"], atol=1e-4)
    state = torch.load(res["best_model_path"], map_location="cpu")
    assert list(state) == ["model.0.weight", "model.0.bias", "model.2.weight", "model.2.bias"]
    best = torch.cat([v.reshape(-1) for v in state.values()]).numpy()
    np.testing.assert_allclose(best, g[f"c{cid}_best_params"], atol=2e-4)           # Adam-amplified fp noise, 25 epochs
    np.testing.assert_allclose(H.flat_params(model), g[f"c{cid}_final_params"], atol=5e-4)


def _run_main_at(tmp_path, cid, bag_dtype, dev):
    """run_moc.main() on fixture task `cid` with the resident store rounding the fp32 bag files to `bag_dtype`."""
    from moc_amd import datasets as DS, main_moc as M, run_moc
    g = H.golden("driver")
    C, j, K, rep, seed, W, We, label_map, data = _task_on_disk(str(tmp_path), cid, g)
    M.set_classifier_bank(W.to(dev), We.to(dev))
    ds = DS.Generic_MIL_Dataset(csv_path=str(tmp_path / "dataset_csv" / "t.csv"), data_dir=data, print_info=False,
                                label_dict=label_map)
    tr, va, te = ds.return_splits(from_id=False, csv_path=str(tmp_path / "splits" / "splits_0.csv"), repeat_num=rep)
    loaders = [DS.to_resident(sp, dev, dtype=bag_dtype) for sp in (tr, va, te)]
    assert loaders[0].X.dtype == bag_dtype
    args = run_moc.get_args(["--topj", str(j), "--topk", str(K), "--shot", "4", "--fold", "0", "--disable_tqdm",
                             "--result_dir", str(tmp_path / "res")])
    args.n_classes = C
    torch.manual_seed(seed)
    model = M.senet(512, 4).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    val_aucs, orig = [], M.evaluation

    def logged(model_, loader, device, a):
        r = orig(model_, loader, device, a)
        if loader is loaders[1]:
            val_aucs.append(r["auc"])
        return r
    M.evaluation = logged
    try:
        torch.manual_seed(seed + 1)
        res = run_moc.main(args, model, opt, *loaders, dev)
    finally:
        M.evaluation = orig
    zs = json.load(open(tmp_path / "res" / "zs_results_shot_4_fold_0.json"))
    got_zs = np.array([[zs[k]["loss"], zs[k]["acc"], zs[k]["auc"]] for k in ("zs_train", "zs_val", "zs_test")])
    return g, res, val_aucs, got_zs


@pytest.mark.parametrize("bag_dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cid", [0, 1])
def test_16bit_storage_keeps_the_untrained_numbers_of_the_reference(gpu_device, tmp_path, cid, bag_dtype):
    """From the SAME fp32 bag files, with the resident store rounding to bf16 / fp16: everything that involves no
    training -- the three zero-shot evaluations of main() (main_moc.py:590-600) -- stays within the reference's bars
    (loss 1e-4, accuracy equal, AUC +-0.002).  Measured deviation of the loss: <= 5e-6."""
    g, res, val_aucs, got_zs = _run_main_at(tmp_path, cid, bag_dtype, gpu_device)
    exp = g[f"c{cid}_zs"]
    np.testing.assert_allclose(got_zs[:, 0], exp[:, 0], atol=1e-4)
    np.testing.assert_array_equal(got_zs[:, 1], exp[:, 1])
    np.testing.assert_allclose(got_zs[:, 2], exp[:, 2], atol=2e-3)
    assert len(val_aucs) == 25 and all(0.0 <= v <= 1.0 for v in val_aucs)


@pytest.mark.parametrize("bag_dtype", [torch.bfloat16, torch.float16])
@pytest.mark.xfail(strict=True, reason="16-bit bag STORAGE is outside north_star's +-0.002 once the meta-learner trains: rounding the "
                   "embeddings moves the logits by ~1e-4 relative, rows swap at the top-j / top-K boundaries, and 25 epochs of Adam "
                   "amplify it -- measured per-epoch val AUC off by 0.056 / 0.073 (bf16) and 0.056 / 0.094 (fp16) on the two fixture tasks "
                   "(profiles/round3_storage_fidelity.jsonl).  That is why bench.py's default storage is fp32.  Strict: should 16-bit "
                   "storage ever land inside the bar, this goes red and the headline can move back.")
def test_16bit_storage_trained_auc_within_reference_bar(gpu_device, tmp_path, bag_dtype):
    worst = 0.0
    for cid in (0, 1):
        g, res, val_aucs, _ = _run_main_at(tmp_path / f"c{cid}", cid, bag_dtype, gpu_device)
        worst = max(worst, float(np.max(np.abs(np.asarray(val_aucs) - g[f"c{cid}_val_auc"]))),
                    abs(res["best_val"] - float(g[f"c{cid}_result"][0])), abs(res["test_at_best_val"] - float(g[f"c{cid}_result"][1])))
    assert worst < 2e-3, f"largest AUC deviation from the reference main() at {bag_dtype}: {worst:.4f}"


@pytest.mark.parametrize("bag_dtype", [torch.bfloat16, torch.float16])
def test_evaluation_fixtures_at_16bit_storage_from_fp32_bags(gpu_device, bag_dtype):
    """evaluation() (main_moc.py:462-520) of the fixtures' seeded, untrained meta-learner on 16-bit copies of the fp32
    bags against the REFERENCE's numbers: the loss stays within 1e-4 (measured 5e-6), accuracy and AUC are equal."""
    from moc_amd import main_moc as M
    g = H.golden("evaluation")
    dev = gpu_device
    for cid, ns, N, C, j, K, dmask, repeat_num, seed in g["cases"]:
        ns, N, C, j, K = int(ns), int(N), int(C), int(j), int(K)
        W, We = synth.make_bank(seed, 512, C)
        bags, labels = synth.make_slide_set(seed + 100, [N] * ns, 512, We, C)
        torch.manual_seed(int(seed))
        model = M.senet(512, 4).to(dev)
        M.set_classifier_bank(W.to(dev), We.to(dev))
        res = M.ResidentBags(bags, labels, dev, dtype=bag_dtype, repeat_num=int(repeat_num) or None)
        got = M.evaluation(model, res, dev, H.make_args(C, j, K, H.discard_from_mask(dmask)))
        exp = g[f"c{cid}_eval"]
        assert abs(got["loss"] - exp[0]) < 1e-4 and abs(got["acc"] - exp[1]) < 1e-12 and abs(got["auc"] - exp[2]) < 2e-3, (cid, got, exp)


def test_nsclc16_sized_task_fp32_vs_bf16_storage(gpu_device):
    """An NSCLC-16-shot-sized synthetic task (32 / 64 / 202 slides), 25 epochs at fp32 and at bf16 storage from the same
    fp32 bags.  What rounding alone does to a FIXED model is bounded here (pooled logits of the fp32-trained model on
    bf16 copies of the test bags: measured 2e-4..4e-4 at these sizes, 1e-2 at 15 k rows); what it does to the TRAINED
    AUC is reported by scripts/storage_fidelity.py and is reseeding-sized (profiles/round3_dp_auc_control.jsonl)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import storage_fidelity as SF
    task = SF.nsclc_task(31020, confusion=0.30, gain=0.20)
    base, model32, loaders32, args = SF.run_loop(task, "fp32")
    r16, _, _, _ = SF.run_loop(task, "bf16")
    p32 = SF.pooled_logits(model32, loaders32[2], args)
    from moc_amd import main_moc as M
    te16 = M.ResidentBags(*task["splits"][2], gpu_device, dtype=torch.bfloat16)
    p16 = SF.pooled_logits(model32, te16, args)
    d_pool = float((p16 - p32).abs().max())
    print(f"fp32 best-val {base['best_val']:.4f} test {base['test_at_best_val']:.4f} | bf16 best-val {r16['best_val']:.4f} "
          f"test {r16['test_at_best_val']:.4f} | max |d pooled logit| of the same model {d_pool:.2e}")
    assert 1e-6 < d_pool < 5e-3                       # rounding is visible (not within 1e-4 in general) and small
    assert base["best_val"] > 0.9 and abs(r16["best_val"] - base["best_val"]) < 0.03 and abs(r16["test_at_best_val"] - base["test_at_best_val"]) < 0.03


def test_file_backed_dataloader_equals_resident(gpu_device, tmp_path):
    """--resident 0 (a torch DataLoader over bag files, as the reference runs) and the resident path give
    the same numbers; the DataLoader's base-seed draw from the default generator is part of the stream."""
    from moc_amd import datasets as DS, main_moc as M
    g = H.golden("driver")
    C, j, K, rep, seed, W, We, label_map, data = _task_on_disk(str(tmp_path), 0, g)
    dev = gpu_device
    M.set_classifier_bank(W.to(dev), We.to(dev))
    ds = DS.Generic_MIL_Dataset(csv_path=str(tmp_path / "dataset_csv" / "t.csv"), data_dir=data, print_info=False, label_dict=label_map)
    tr, va, te = ds.return_splits(from_id=False, csv_path=str(tmp_path / "splits" / "splits_0.csv"), repeat_num=rep)
    for sp in (tr, va, te):
        sp.load_full_path(True)
    args = H.make_args(C, j, K)
    out = []
    for mode in ("loader", "resident"):
        torch.manual_seed(seed)
        model = M.senet(512, 4).to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
        tr.repeat_num = rep
        # loader_seed_draw: the resident split makes the draw DataLoader.__iter__ makes for its base seed, every pass
        loader = (torch.utils.data.DataLoader(tr, batch_size=1, shuffle=False, num_workers=0) if mode == "loader"
                  else DS.to_resident(tr, dev, loader_seed_draw=True))
        vloader = (torch.utils.data.DataLoader(va, batch_size=1, shuffle=False, num_workers=0) if mode == "loader"
                   else DS.to_resident(va, dev, loader_seed_draw=True))
        torch.manual_seed(77)
        for _ in range(2):
            M.train(model, loader, opt, dev, args)
            mid = M.evaluation(model, vloader, dev, args)          # (an evaluation between the passes draws too)
        out.append((H.flat_params(model), mid, M.evaluation(model, vloader, dev, args), torch.get_rng_state()))
    np.testing.assert_array_equal(out[0][0], out[1][0])
    assert out[0][1] == out[1][1] and out[0][2] == out[1][2]
    assert torch.equal(out[0][3], out[1][3]), "the two paths leave the CPU generator in different states"


def test_run_many_packs_runs_onto_the_gpu(gpu_device, tmp_path):
    """moc_amd.run_many (the reference's scripts/moc_train.sh as a job queue) with real moc_amd.run_moc processes: two
    folds side by side on the one GPU, each a whole (short) synthetic run; result files per fold where the reference's
    script puts them."""
    from moc_amd import run_many
    rc = run_many.main(["--folds", "0", "1", "--shots", "2", "--gpus", "0", "--runs-per-gpu", "2", "--seed", "3",
                        "--result_dir", str(tmp_path / "res"), "--", "--synthetic", "6", "--epochs", "2", "--disable_tqdm",
                        "--topj", "100", "--topk", "5"])
    assert rc == 0
    d = tmp_path / "res" / "2_shot"
    for fold in (0, 1):
        best = json.load(open(d / f"best_results_shot_2_fold_{fold}.json"))
        assert 0.0 <= best["best_val"] <= 1.0 and os.path.exists(best["best_model_path"])
        assert "End training." in open(d / f"fold_{fold}_shot_2_output.txt").read()
