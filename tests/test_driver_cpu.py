"""CPU tests of the widened rows f2 (ingestion) and f1 (--summary): CSV / split semantics of the
reference's dataset_generic.py, bag file formats, and the summary CSVs against the fixture the
reference's own summary block produced."""
import importlib.util
import os
import sys

import numpy as np
import pandas as pd
import pytest
import torch

import helpers as H
from moc_amd import datasets as DS


def _make_task(root, n_per_class=(7, 5), fmt="pt"):
    os.makedirs(os.path.join(root, "dataset_csv"), exist_ok=True)
    rows, sid = [], 0
    for cls, n in zip(("LUAD", "LUSC"), n_per_class):
        for _ in range(n):
            rows.append((f"patient_{sid:03d}", f"0{sid:04d}", cls))       # zero-padded numeric ids must survive
            sid += 1
    pd.DataFrame(rows, columns=["case_id", "slide_id", "label"]).to_csv(os.path.join(root, "dataset_csv", "t.csv"), index=False)
    ids = [r[1] for r in rows]
    split = pd.DataFrame({"train": pd.Series([ids[9], ids[0], ids[7], ids[1]]), "val": pd.Series(ids[2:5]),
                          "test": pd.Series(ids[5:7] + ids[10:12])})
    os.makedirs(os.path.join(root, "splits"), exist_ok=True)
    split.to_csv(os.path.join(root, "splits", "splits_0.csv"))
    data = os.path.join(root, "data")
    g = torch.Generator().manual_seed(0)
    for k, i in enumerate(ids):
        DS.write_bag(data, i, torch.randn(20 + k, 512, generator=g), coords=np.arange((20 + k) * 2).reshape(-1, 2),
                     fmt=fmt if k % 2 == 0 else "npy")
    return ids, data


def test_splits_follow_reference_semantics(tmp_path):
    ids, data = _make_task(str(tmp_path))
    ds = DS.Generic_MIL_Dataset(csv_path=str(tmp_path / "dataset_csv" / "t.csv"), data_dir=data, shuffle=False, seed=1,
                                print_info=False, label_dict={"LUAD": 0, "LUSC": 1}, patient_strat=False, ignore=[])
    ds.load_from_h5(True)
    ds.load_full_path(True)
    tr, va, te = ds.return_splits(from_id=False, csv_path=str(tmp_path / "splits" / "splits_0.csv"), repeat_num=6)
    for sp in (tr, va, te):
        sp.load_full_path(True)
    # split membership is by slide id, ORDER is the slide table's (dataset_generic.py:206-207)
    assert tr.slide_data["slide_id"].tolist() == [ids[0], ids[1], ids[7], ids[9]]
    assert tr.slide_data["label"].tolist() == [0, 0, 1, 1]
    assert len(tr) == 6 and tr.real_len() == 4 and len(va) == 3 and va.real_len() == 3
    # repeat_num > real_len revisits from the start; idx >= repeat_num stops iteration
    feats, label, coords, path = tr[4]
    f0 = tr[0][0]
    assert torch.equal(feats, f0) and label == 0 and feats.dtype == torch.float32 and coords.shape == (feats.shape[0], 2)
    with pytest.raises(IndexError):
        tr[6]
    # repeat_num < real_len truncates (RCC-16: 50 listed, 48 used)
    tr.repeat_num = 3
    assert len(tr) == 3 and len(list(iter(lambda it=iter(range(3)): tr[next(it)], None))) == 3
    # the evaluation loops set repeat_num = real_len and restore it
    tr.repeat_num = tr.real_len()
    assert len(tr) == 4
    assert os.path.basename(path).startswith(ids[0])
    # default-collated through a DataLoader this is the item train() unpacks (main_moc.py:381-385)
    item = next(iter(torch.utils.data.DataLoader(va, batch_size=1, shuffle=False)))
    assert item[0].dim() == 3 and item[0].size(0) == 1 and item[1].shape == (1,) and isinstance(item[3][0], str)


def test_label_filtering_and_missing_files(tmp_path):
    ids, data = _make_task(str(tmp_path))
    ds = DS.Generic_MIL_Dataset(csv_path=str(tmp_path / "dataset_csv" / "t.csv"), data_dir=data, print_info=False,
                                label_dict={"LUAD": 0, "LUSC": 1}, ignore=["LUSC"])
    assert set(ds.slide_data["label"]) == {0} and len(ds.slide_data) == 7
    with pytest.raises(FileNotFoundError, match="no bag for slide"):
        DS.read_bag(data, "does-not-exist")


def test_summary_matches_reference_fixture(tmp_path):
    spec = importlib.util.spec_from_file_location("make_golden_inputs", os.path.join(H.GOLDEN_DIR, "make_golden.py"))
    src = open(spec.origin).read()
    ns = {"np": np, "os": os}
    start = src.index("def summary_inputs(td, kind):")
    exec(src[start:src.index("def gen_summary():")], ns)           # the input generator only (no reference needed)
    from moc_amd import run_moc
    g = H.golden("summary")
    for kind in ("full", "nozs", "ablation"):
        td = str(tmp_path / kind)
        os.makedirs(td)
        ns["summary_inputs"](td, kind)
        run_moc.summary(run_moc.get_args(["--summary", "--summary_dir", td]))
        for shot in (1, 2, 4, 8):
            df = pd.read_csv(os.path.join(td, f"summary_{shot}.csv"))
            assert list(df.columns) == list(g[f"{kind}_{shot}_cols"])
            assert df["fold"].tolist() == ["0", "1", "2", "3", "4", "mean"]
            np.testing.assert_allclose(df.drop(columns=["fold"]).to_numpy(dtype=np.float64), g[f"{kind}_{shot}_vals"], rtol=0, atol=1e-12)


def test_cli_flags_match_reference_defaults():
    from moc_amd import run_moc
    a = run_moc.get_args([])
    assert (a.fold, a.shot, a.topj, a.topk, a.result_dir, a.dataset, a.pretrain) == (0, 1, 10, 10, "results/moc_train", "nsclc", "conch")
    assert a.discard_classifiers == [] and a.ablation_study == "none" and a.check_zeroshot is True and not a.summary
    a = run_moc.get_args("--fold 3 --shot 8 --topj 400 --dataset rcc --discard_classifiers topk bottomk --disable_tqdm".split())
    assert a.fold == 3 and a.topj == 400 and a.dataset == "rcc" and a.discard_classifiers == ["topk", "bottomk"] and a.disable_tqdm


def test_direct_auc_equals_sklearn():
    """main_moc._auc replaces roc_auc_score on the evaluation path: same numbers, ties included."""
    import numpy as np
    import pytest
    from sklearn.metrics import roc_auc_score
    from moc_amd.main_moc import _auc
    rng = np.random.default_rng(3)
    for n in (2, 7, 202, 1000):
        y = rng.integers(0, 2, n)
        y[0], y[1] = 0, 1
        p = rng.random(n)
        assert abs(_auc(y, p) - roc_auc_score(y, p)) < 1e-12
        pq = np.round(p, 1)                                   # heavy ties
        assert abs(_auc(y, pq) - roc_auc_score(y, pq)) < 1e-12
    for C in (3, 5, 30):
        n = 40 * C
        y = rng.integers(0, C, n)
        y[:C] = np.arange(C)
        logits = rng.standard_normal((n, C)) + 1.5 * np.eye(C)[y]
        logits = np.round(logits, 1)                          # ties across rows
        p = np.exp(logits) / np.exp(logits).sum(1, keepdims=True)
        ref = roc_auc_score(y, p, multi_class="ovo", average="macro")
        assert abs(_auc(y, p) - ref) < 1e-12
    # inputs sklearn rejects (or answers with NaN + a warning, depending on its version) get sklearn's answer
    import warnings
    for y, p, kw in ((np.zeros(5, dtype=np.int64), np.linspace(0, 1, 5), {}),
                     (np.array([0, 1, 1, 0]), np.full((4, 3), 1 / 3), dict(multi_class="ovo", average="macro"))):
        def run(f, *a, **k):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                try:
                    return ("value", f(*a, **k))
                except ValueError as e:
                    return ("error", str(e))
        mine, theirs = run(_auc, y, p), run(roc_auc_score, y, p, **kw)
        assert mine[0] == theirs[0]
        assert mine[1] == theirs[1] or (isinstance(mine[1], float) and np.isnan(mine[1]) and np.isnan(theirs[1]))


def test_folds_are_dealt_to_the_ranks_exactly_once():
    from moc_amd.run_moc import folds_of_rank
    for world in (1, 2, 3, 8):
        got = [folds_of_rank("0,1,2,3,4", r, world) for r in range(world)]
        assert sorted(f for part in got for f in part) == [0, 1, 2, 3, 4]
        assert max(len(p) for p in got) - min(len(p) for p in got) <= 1
    assert folds_of_rank("3, 1", 0, 1) == [3, 1]
    with pytest.raises(AssertionError):
        folds_of_rank("1,1", 0, 1)
