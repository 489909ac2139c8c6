"""Row f2 pinned to the reference: moc_amd.datasets reproduces, entry by entry, what the reference's own
Generic_MIL_Dataset / Generic_Split (datasets/dataset_generic.py:40-231, :343-441, :484-504) produced for its own
dataset_csv/{nsclc,rcc}.csv x splits/*_fewshot/{1,2,4,8,16}shots/splits_{0-4}.csv -- tests/golden/splits.npz, written by
`make_golden.py splits` from the reference classes run in the build container; the CSV inputs are committed as data
under tests/golden/ref_data/."""
import os

import numpy as np
import pytest
import torch

import helpers as H
from moc_amd import datasets as DS

TASKS = {"nsclc": ({"LUAD": 0, "LUSC": 1}, 2), "rcc": ({"KICH": 0, "KIRC": 1, "KIRP": 2}, 3)}
SHOTS = (1, 2, 4, 8, 16)
DATA = os.path.join(H.GOLDEN_DIR, "ref_data")
_Z = torch.zeros(1, 4)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(H.GOLDEN_DIR, "splits.npz"))


def _visit_ids(sp):
    """Slide ids a shuffle=False loader visits: the index arithmetic of __getitem__ without touching a bag file."""
    seen = []
    real = DS.read_bag
    DS.read_bag = lambda data_dir, slide_id: (_Z, np.zeros((1, 2), dtype=np.int64), slide_id)
    try:
        for i in range(len(sp)):
            seen.append(sp[i][3])
        with pytest.raises(IndexError):
            sp[len(sp)]
    finally:
        DS.read_bag = real
    return seen


@pytest.mark.parametrize("task", sorted(TASKS))
def test_every_split_of_the_reference_is_reproduced(gold, task):
    label_dict, C = TASKS[task]
    ds = DS.Generic_MIL_Dataset(csv_path=os.path.join(DATA, "dataset_csv", task + ".csv"), data_dir="unused", shuffle=False,
                                seed=1, print_info=False, label_dict=label_dict, patient_strat=False, ignore=[])
    ds.load_from_h5(True)
    ds.load_full_path(True)
    table = gold[f"{task}/table_ids"].tolist()
    assert ds.slide_data["slide_id"].tolist() == table
    assert ds.slide_data["label"].tolist() == gold[f"{task}/table_labels"].tolist()
    assert len(ds) == int(gold[f"{task}/len"]) == ds.real_len()
    checked = 0
    for shot in SHOTS:
        for fold in range(5):
            csv = os.path.join(DATA, "splits", f"{task}_fewshot", f"{shot}shots", f"splits_{fold}.csv")
            splits = ds.return_splits(from_id=False, csv_path=csv, repeat_num=int(shot) * C)
            for name, sp in zip(("train", "val", "test"), splits):
                key = f"{task}/{shot}/{fold}/{name}"
                sp.load_full_path(True)
                want = [table[i] for i in gold[key + "/idx"]]
                assert sp.slide_data["slide_id"].tolist() == want, key          # membership AND order (table order)
                assert [int(v) for v in sp.slide_data["label"]] == gold[key + "/labels"].tolist(), key
                assert [len(sp), sp.real_len()] == gold[key + "/len_real"].tolist(), key
                assert [len(c) for c in sp.slide_cls_ids] == gold[key + "/cls_counts"].tolist(), key
                assert _visit_ids(sp) == [table[i] for i in gold[key + "/visit"]], key   # repeat_num wrap / truncation
                assert bool(gold[key + "/stops"])
                if name == "train":
                    keep = sp.repeat_num
                    sp.repeat_num = sp.real_len()                               # main_moc.py:469-471
                    assert len(sp) == int(gold[key + "/len_eval"]), key
                    sp.repeat_num = keep
                checked += 1
    assert checked == 75


def test_rcc_sixteen_shot_lists_fifty_and_uses_forty_eight(gold):
    """the split file lists more train slides than shot * n_classes; the loader stops at repeat_num."""
    hit = 0
    for fold in range(5):
        n_len, n_real = gold[f"rcc/16/{fold}/train/len_real"].tolist()
        assert n_len == 48
        if n_real > n_len:
            hit += 1
            assert len(gold[f"rcc/16/{fold}/train/visit"]) == 48
            assert gold[f"rcc/16/{fold}/train/visit"].tolist() == gold[f"rcc/16/{fold}/train/idx"][:48].tolist()
    assert hit > 0
