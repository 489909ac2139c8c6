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
