"""Bag ingestion for the MOC path.

Mirrors the part of the reference's datasets/dataset_generic.py that main_moc.py uses
(:209-220, :270-281): a slide table read from `dataset_csv/*.csv`, few-shot splits read from
`splits/*/splits_k.csv`, split objects with the `repeat_num` virtual length, and the item
contract `(features [N, D] f32, label, coords [N, 2], full_path)` (dataset_generic.py:380-433).
Same class / method names so the driver reads like the reference's.

Bag files: the CLAM layout `{data_dir}/h5_files/{slide_id}.h5` (datasets `features`, `coords`) when
h5py is importable; otherwise, or when those are absent, `{data_dir}/pt_files/{slide_id}.pt`
(a tensor, or a dict with `features`/`coords`) or `{data_dir}/npy_files/{slide_id}.npy`.

`to_resident` reads every slide of a split ONCE and packs it into HBM (main_moc.ResidentBags):
the reference re-reads each h5 file and pipes ~30 MB through a DataLoader worker on every visit.
"""
from __future__ import annotations

import os

import numpy as np
import pandas as pd
import torch

try:  # optional: absent in the build container
    import h5py
except ImportError:  # pragma: no cover
    h5py = None


def read_bag(data_dir: str, slide_id: str):
    """-> (features float32 [N, D] tensor, coords int64 [N, 2] ndarray, path)."""
    h5p = os.path.join(data_dir, "h5_files", f"{slide_id}.h5")
    if h5py is not None and os.path.exists(h5p):
        with h5py.File(h5p, "r") as f:
            return torch.from_numpy(f["features"][:]), f["coords"][:], h5p
    ptp = os.path.join(data_dir, "pt_files", f"{slide_id}.pt")
    if os.path.exists(ptp):
        obj = torch.load(ptp, map_location="cpu")
        if isinstance(obj, dict):
            feats = torch.as_tensor(obj["features"])
            c = obj.get("coords")
            coords = np.zeros((feats.shape[0], 2), dtype=np.int64) if c is None else np.asarray(c)
        else:
            feats, coords = obj, np.zeros((obj.shape[0], 2), dtype=np.int64)
        return feats, coords, ptp
    npp = os.path.join(data_dir, "npy_files", f"{slide_id}.npy")
    if os.path.exists(npp):
        feats = torch.from_numpy(np.load(npp))
        return feats, np.zeros((feats.shape[0], 2), dtype=np.int64), npp
    if os.path.exists(h5p):
        raise RuntimeError(f"{h5p} exists but h5py is not installed; convert to pt_files/ or npy_files/")
    raise FileNotFoundError(f"no bag for slide {slide_id!r} under {data_dir} (h5_files/, pt_files/, npy_files/)")


class Generic_Split:
    """One of train/val/test (dataset_generic.py:484-504 + the inherited item access)."""

    def __init__(self, slide_data: pd.DataFrame, data_dir=None, num_classes=2, bag_size=None, repeat_num=None):
        self.slide_data = slide_data.reset_index(drop=True)
        self.data_dir = data_dir
        self.num_classes = num_classes
        self.bag_size = bag_size
        self.repeat_num = repeat_num
        self.use_h5 = False
        self.return_full_path = False
        self.slide_cls_ids = [np.where(self.slide_data["label"] == i)[0] for i in range(num_classes)]

    def load_from_h5(self, toggle):
        self.use_h5 = toggle

    def load_full_path(self, toggle):
        self.return_full_path = toggle

    def real_len(self):
        return len(self.slide_data)

    def __len__(self):
        # dataset_generic.py:500-504 (`is not None`); the evaluation loops set it to real_len and back
        return self.repeat_num if self.repeat_num is not None else len(self.slide_data)

    def __getitem__(self, idx):
        n = len(self.slide_data)
        if self.repeat_num:
            if idx >= self.repeat_num:
                raise IndexError           # stop iteration
            idx = idx % n                  # repeat_num > n revisits; < n truncates (RCC-16: 50 listed, 48 used)
        elif idx >= n:
            raise IndexError
        slide_id = self.slide_data["slide_id"][idx]
        label = int(self.slide_data["label"][idx])
        feats, coords, path = read_bag(self.data_dir, slide_id)
        feats = feats.to(torch.float32)
        if self.return_full_path:
            return feats, label, coords, path
        return feats, label, coords


class Generic_MIL_Dataset:
    """Slide table + split factory (dataset_generic.py:38-147, :201-267, :343-393)."""

    def __init__(self, csv_path, data_dir=None, shuffle=False, seed=7, print_info=True, label_dict=None,
                 filter_dict=None, ignore=(), patient_strat=False, label_col=None, **kwargs):
        self.label_dict = dict(label_dict or {})
        self.num_classes = len(set(self.label_dict.values()))
        self.data_dir = data_dir
        self.seed, self.print_info, self.patient_strat = seed, print_info, patient_strat
        self.use_h5 = False
        self.return_full_path = False
        self.repeat_num = None
        df = pd.read_csv(csv_path, dtype=str)
        for key, vals in (filter_dict or {}).items():
            df = df[df[key].isin(vals)]
        label_col = label_col or "label"
        if label_col != "label":
            df["label"] = df[label_col].copy()
        df = df[~df["label"].isin(list(ignore))].reset_index(drop=True)
        df["label"] = [self.label_dict[k] for k in df["label"]]
        if shuffle:
            df = df.sample(frac=1.0, random_state=seed).reset_index(drop=True)
        self.slide_data = df
        if print_info:
            for i in range(self.num_classes):
                print("Slide-LVL; Number of samples registered in class %d: %d" % (i, int((df["label"] == i).sum())))

    def load_from_h5(self, toggle):
        self.use_h5 = toggle

    def load_full_path(self, toggle):
        self.return_full_path = toggle

    def real_len(self):
        return len(self.slide_data)

    def __len__(self):
        return self.repeat_num if self.repeat_num else len(self.slide_data)

    def get_split_from_df(self, all_splits, split_key="train", bag_size=None, repeat_num=None):
        ids = all_splits[split_key].dropna().reset_index(drop=True)
        if len(ids) == 0:
            return None
        mask = self.slide_data["slide_id"].isin(ids.tolist())       # table order, not split-file order
        return Generic_Split(self.slide_data[mask], data_dir=self.data_dir, num_classes=self.num_classes,
                             bag_size=bag_size, repeat_num=repeat_num)

    def return_splits(self, from_id=True, csv_path=None, bag_size=None, repeat_num=None, **kwargs):
        if from_id:
            raise NotImplementedError("only csv-defined splits are on the MOC path (main_moc.py:220)")
        assert csv_path
        all_splits = pd.read_csv(csv_path, dtype=self.slide_data["slide_id"].dtype)   # keep ids as strings
        return (self.get_split_from_df(all_splits, "train", bag_size, repeat_num),
                self.get_split_from_df(all_splits, "val"),
                self.get_split_from_df(all_splits, "test"))


def to_resident(split: Generic_Split, device, dtype=None, loader_seed_draw=False):
    """Read every slide of `split` once and keep the split packed in HBM.  `loader_seed_draw`: every pass makes the
    base-seed draw the reference's DataLoader makes (main_moc.ResidentBags)."""
    from .main_moc import ResidentBags
    bags, labels, paths = [], [], []
    n = split.real_len()
    for i in range(n):
        feats, coords, path = read_bag(split.data_dir, split.slide_data["slide_id"][i])
        bags.append(feats.to(torch.float32))
        labels.append(int(split.slide_data["label"][i]))
        paths.append(path)
    return ResidentBags(bags, labels, device, dtype=dtype, repeat_num=split.repeat_num, paths=paths,
                        loader_seed_draw=loader_seed_draw)


def to_sharded(split: Generic_Split, device, rank: int, world: int, dtype=None, train=False, group=None):
    """Multi-GPU form of to_resident: this rank reads ONLY the contiguous block of slides it will hold and keeps it
    packed in HBM -> moc_amd.dist.ShardedSplit (`train`: a SeqShardedBags for dist.train_seq, which also needs every
    slide's row count -- exchanged once through the process group)."""
    import torch.distributed as dist
    from . import dist as mdist
    n = split.real_len()
    labels = [int(v) for v in split.slide_data["label"]]
    if train and world > n:               # from (n, world) alone: the same on every rank, before anything collective
        raise ValueError(f"exact-sequential training: the train split has {n} slide(s) but the job has {world} ranks -- "
                         f"every rank must hold at least one slide; run with at most {n} rank(s)")
    blocks = mdist.block_lists(n, world)
    bags, paths = [], []
    for i in blocks[rank]:
        feats, coords, path = read_bag(split.data_dir, split.slide_data["slide_id"][i])
        bags.append(feats.to(torch.float32))
        paths.append(path)
    if not train:
        return mdist.ShardedSplit(bags, blocks[rank], labels, blocks, device, dtype=dtype, paths=paths)
    box = [None] * world
    dist.all_gather_object(box, [int(b.size(0)) for b in bags], group=group)
    sizes = [v for part in box for v in part]
    sh = mdist.SeqShardedBags(bags, sizes, labels, device, rank, world, dtype=dtype, paths=paths)
    if split.repeat_num and split.repeat_num != n:
        sh.repeat_num = split.repeat_num
    return sh


def write_bag(data_dir: str, slide_id: str, features: torch.Tensor, coords=None, fmt="pt"):
    """Test/demo helper: store one bag in one of the layouts read_bag understands."""
    feats = features.detach().cpu().to(torch.float32)
    coords = np.zeros((feats.shape[0], 2), dtype=np.int64) if coords is None else np.asarray(coords)
    if fmt == "h5":
        assert h5py is not None, "h5py is not installed"
        os.makedirs(os.path.join(data_dir, "h5_files"), exist_ok=True)
        with h5py.File(os.path.join(data_dir, "h5_files", f"{slide_id}.h5"), "w") as f:
            f.create_dataset("features", data=feats.numpy())
            f.create_dataset("coords", data=coords)
    elif fmt == "pt":
        os.makedirs(os.path.join(data_dir, "pt_files"), exist_ok=True)
        torch.save({"features": feats, "coords": torch.from_numpy(coords)},        # tensors only: loads with weights_only
                   os.path.join(data_dir, "pt_files", f"{slide_id}.pt"))
    elif fmt == "npy":
        os.makedirs(os.path.join(data_dir, "npy_files"), exist_ok=True)
        np.save(os.path.join(data_dir, "npy_files", f"{slide_id}.npy"), feats.numpy())
    else:
        raise ValueError(fmt)
