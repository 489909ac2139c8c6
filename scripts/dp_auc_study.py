#!/usr/bin/env python3
"""Does minibatch data parallelism keep the reference's AUC?  (SURVEY.md section 8e mode 2: "must be validated
against AUC within +-0.002")

The reference takes one Adam step per slide (main_moc.py:380-410).  A G-rank synchronous data-parallel run takes one
step per G slides.  This script runs the reference's 25-epoch loop (best-validation bookkeeping, main_moc.py:611-628)
on the GPU, sequentially (G = 1: the path the parity tests pin to the reference) and with minibatches of G in
{2, 4, 8} (moc_amd.dist.train_minibatch: the trajectory of train_dp at world G, by gradient accumulation on one GPU),
with the learning rate as is, x sqrt(G) and x G, and prints best-val AUC / test AUC at best val / best epoch:

  * the two tasks of tests/golden/driver.npz (8 and 6 train slides: what the reference's own main() was run on);
  * NSCLC-16-shot-sized synthetic tasks (32 train, 64 val, 202 test slides) at three difficulties (share of planted
    rows pointing at the wrong class 0.47 / 0.40 / 0.30), two seeds each.

    python scripts/dp_auc_study.py > gpurun_out/dp_auc_study.jsonl
"""
import json
import math
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moc_amd import dist as mdist, main_moc as M, synth  # noqa: E402


def run(task, G, lr, epochs=25, reseed=0):
    """`reseed` != 0: another (initialisation, mask stream) pair for the SAME task data -- the reference seeds neither
    (main_moc.py:315, :330), so two of its own runs differ exactly like this."""
    dev = torch.device("cuda:0")
    C, j, K, seed = task["C"], task["j"], task["K"], task["seed"] + 7777 * reseed
    M.set_classifier_bank(task["W"].to(dev), task["We"].to(dev))
    args = types.SimpleNamespace(disable_tqdm=True, n_classes=C, topj=j, topk=K, discard_classifiers=[],
                                 pretrain="conch", ablation_study="none")
    tr, va, te = task["loaders"]
    torch.manual_seed(seed)
    model = M.senet(512, 4).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=1e-4)
    torch.manual_seed(seed + 1)
    best_val, test_at_best, best_epoch, vals = 0.0, 0.0, 0, []
    for ep in range(epochs):
        if G == 1:
            M.train(model, tr, opt, dev, args)
        else:
            mdist.train_minibatch(model, tr, opt, dev, args, G)
        v = M.evaluation(model, va, dev, args)
        vals.append(v["auc"])
        if v["auc"] > best_val:
            best_val, best_epoch = v["auc"], ep
            test_at_best = M.evaluation(model, te, dev, args)["auc"]
    return {"best_val": best_val, "test_at_best_val": test_at_best, "best_epoch": best_epoch, "val_auc": [round(x, 4) for x in vals]}


def fixture_task(cid):
    g = np.load(os.path.join(ROOT, "tests", "golden", "driver.npz"))
    _, ntr, nva, nte, C, j, K, rep, seed = [int(v) for v in g["cases"][cid]]
    W, We = synth.make_bank(seed, 512, C)
    dev = torch.device("cuda:0")
    loaders = []
    for s_i in range(3):
        sizes = [int(v) for v in g[f"c{cid}_sizes{s_i}"]]
        bags, labels = synth.make_slide_set(seed + 1000 * (s_i + 1), sizes, 512, We, C, confusion=0.47, gain=0.12)
        loaders.append(M.ResidentBags(bags, labels, dev, repeat_num=rep if s_i == 0 else None))
    ref = {"best_val": float(g[f"c{cid}_result"][0]), "test_at_best_val": float(g[f"c{cid}_result"][1]),
           "best_epoch": int(g[f"c{cid}_result"][3])}
    return {"name": f"driver.npz case {cid} ({ntr} train slides, {rep} visits/epoch, {C}-way)", "C": C, "j": j, "K": K,
            "seed": seed, "W": W, "We": We, "loaders": loaders, "reference": ref}


def nsclc_task(seed, n=(32, 64, 202), mean_rows=2500, confusion=0.47, gain=0.12):
    C, j, K = 2, 400, 10
    W, We = synth.make_bank(seed, 512, C)
    dev = torch.device("cuda:0")
    loaders = []
    for s_i, m in enumerate(n):
        sizes = synth.bag_sizes(seed + 17 * s_i, m, mean_rows, fixed=False, lo=800, hi=8000)
        bags, labels = synth.make_slide_set(seed + 1000 * (s_i + 1), sizes, 512, We, C, confusion=confusion, gain=gain)
        loaders.append(M.ResidentBags(bags, labels, dev))
    return {"name": f"synthetic NSCLC 2-way 16-shot (32 / 64 / 202 slides of ~{mean_rows} rows, confusion {confusion}, gain {gain}), seed {seed}",
            "C": C, "j": j, "K": K, "seed": seed, "W": W, "We": We, "loaders": loaders, "reference": None}


def control(tasks, n_seeds=5):
    """The control the minibatch study lacked: the SEQUENTIAL loop's own run-to-run spread
    over (init, mask) seeds, per task; G = 2 / 4 / 8 at the same seeds beside it, learning rate as is."""
    for task in tasks:
        rows = {G: [run(task, G, 1e-3, reseed=k) for k in range(n_seeds)] for G in (1, 2, 4, 8)}
        out = {"control": True, "task": task["name"], "seeds": n_seeds}
        for G, rs in rows.items():
            bv = np.array([r["best_val"] for r in rs])
            ta = np.array([r["test_at_best_val"] for r in rs])
            out[f"G{G}"] = {"best_val": [round(float(v), 4) for v in bv], "test_at_best_val": [round(float(v), 4) for v in ta],
                            "best_val_range": round(float(bv.max() - bv.min()), 4), "test_range": round(float(ta.max() - ta.min()), 4),
                            "best_val_mean": round(float(bv.mean()), 4), "test_mean": round(float(ta.mean()), 4),
                            "best_val_std": round(float(bv.std(ddof=1)), 4), "test_std": round(float(ta.std(ddof=1)), 4)}
        # paired differences at equal seeds: minibatch minus sequential
        for G in (2, 4, 8):
            d = np.array([rows[G][k]["test_at_best_val"] - rows[1][k]["test_at_best_val"] for k in range(n_seeds)])
            out[f"G{G}"]["paired_d_test_mean"] = round(float(d.mean()), 4)
            out[f"G{G}"]["paired_d_test_max_abs"] = round(float(np.abs(d).max()), 4)
        print(json.dumps(out), flush=True)


def main():
    tasks = ([fixture_task(0), fixture_task(1)] + [nsclc_task(s) for s in (31000, 31001)] +
             [nsclc_task(s, confusion=0.40, gain=0.16) for s in (31010, 31011)] +
             [nsclc_task(s, confusion=0.30, gain=0.20) for s in (31020, 31021)])
    if "--control" in sys.argv:
        control(tasks)
        return
    for task in tasks:
        base = run(task, 1, 1e-3)
        print(json.dumps({"task": task["name"], "G": 1, "lr": 1e-3, **base, "reference_main": task["reference"]}), flush=True)
        for G in (2, 4, 8):
            for rule, lr in (("same", 1e-3), ("sqrt", 1e-3 * math.sqrt(G)), ("linear", 1e-3 * G)):
                r = run(task, G, lr)
                r["d_best_val"] = round(r["best_val"] - base["best_val"], 4)
                r["d_test_at_best_val"] = round(r["test_at_best_val"] - base["test_at_best_val"], 4)
                r.pop("val_auc")
                print(json.dumps({"task": task["name"], "G": G, "lr_rule": rule, "lr": round(lr, 6), **r}), flush=True)


if __name__ == "__main__":
    main()
