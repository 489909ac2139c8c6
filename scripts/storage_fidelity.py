#!/usr/bin/env python3
"""Does 16-bit bag STORAGE keep the reference's results?  

The reference runs on fp32 h5 embeddings (main_moc.py:329-337).  bench.py's headline stores the bags as bf16 (BASELINE
configs[1] names bf16) and computes in fp32.  Every 16-bit parity test so far fed the ORACLE the already-rounded values;
this script starts from the fp32 bags, lets the resident store round them (bf16 / fp16), and compares with what the
fp32 path -- the one pinned to the reference's own main() fixture -- gives for the same task:

  1. the two tasks of tests/golden/driver.npz through the reference's 25-epoch loop (main_moc.py:611-628): per-epoch
     validation AUC, best-val AUC, test AUC at best val, best epoch -- against the REFERENCE main()'s numbers;
  2. tests/golden/evaluation.npz cases through evaluation() (main_moc.py:462-520): loss / acc / AUC deviations from the
     reference's numbers;
  3. NSCLC-16-shot-sized synthetic tasks (32 / 64 / 202 slides) at three difficulties: best-val, test-at-best-val, and
     the largest |difference| of any pooled logit of the SAME (fp32-trained) model evaluated on fp32 vs 16-bit bags.

    python scripts/storage_fidelity.py > gpurun_out/storage_fidelity.jsonl
"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from moc_amd import main_moc as M, synth  # noqa: E402

DEV = torch.device("cuda:0")
DTYPES = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}


def _args(C, j, K, discard=()):
    return types.SimpleNamespace(disable_tqdm=True, n_classes=C, topj=j, topk=K, discard_classifiers=list(discard),
                                 pretrain="conch", ablation_study="none")


def pooled_logits(model, loader, args):
    real, keep = loader.dataset.real_len(), loader.dataset.repeat_num
    loader.dataset.repeat_num = real
    try:
        with torch.no_grad():
            p, _, _ = M._eval_pass(loader, DEV, args, "eval", model=model)
    finally:
        loader.dataset.repeat_num = keep
    return p


def run_loop(task, dtype, epochs=25):
    """The reference's epoch loop at one storage type -> dict + the trained model and loaders."""
    C, j, K, seed = task["C"], task["j"], task["K"], task["seed"]
    M.set_classifier_bank(task["W"].to(DEV), task["We"].to(DEV))
    args = _args(C, j, K)
    loaders = [M.ResidentBags(b, l, DEV, dtype=DTYPES[dtype], repeat_num=rep) for (b, l), rep in zip(task["splits"], task["repeat"])]
    tr, va, te = loaders
    torch.manual_seed(seed)
    model = M.senet(512, 4).to(DEV)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    torch.manual_seed(seed + 1)
    best_val, test_at_best, best_epoch, vals = 0.0, 0.0, 0, []
    for ep in range(epochs):
        M.train(model, tr, opt, DEV, args)
        v = M.evaluation(model, va, DEV, args)
        vals.append(v["auc"])
        if v["auc"] > best_val:
            best_val, best_epoch = v["auc"], ep
            test_at_best = M.evaluation(model, te, DEV, args)["auc"]
    return ({"best_val": best_val, "test_at_best_val": test_at_best, "best_epoch": best_epoch, "val_auc": vals}, model, loaders, args)


def fixture_task(cid):
    g = np.load(os.path.join(ROOT, "tests", "golden", "driver.npz"))
    _, ntr, nva, nte, C, j, K, rep, seed = [int(v) for v in g["cases"][cid]]
    W, We = synth.make_bank(seed, 512, C)
    splits = []
    for s_i in range(3):
        sizes = [int(v) for v in g[f"c{cid}_sizes{s_i}"]]
        splits.append(synth.make_slide_set(seed + 1000 * (s_i + 1), sizes, 512, We, C, confusion=0.47, gain=0.12))
    ref = {"best_val": float(g[f"c{cid}_result"][0]), "test_at_best_val": float(g[f"c{cid}_result"][1]),
           "best_epoch": int(g[f"c{cid}_result"][3]), "val_auc": [float(v) for v in g[f"c{cid}_val_auc"]]}
    return {"name": f"driver.npz case {cid} ({ntr}/{nva}/{nte} slides, {C}-way)", "C": C, "j": j, "K": K, "seed": seed, "W": W, "We": We,
            "splits": splits, "repeat": [rep, None, None], "reference": ref}


def nsclc_task(seed, n=(32, 64, 202), mean_rows=2500, confusion=0.47, gain=0.12):
    C, j, K = 2, 400, 10
    W, We = synth.make_bank(seed, 512, C)
    splits = []
    for s_i, m in enumerate(n):
        sizes = synth.bag_sizes(seed + 17 * s_i, m, mean_rows, fixed=False, lo=max(800, mean_rows // 3), hi=max(8000, 3 * mean_rows))
        splits.append(synth.make_slide_set(seed + 1000 * (s_i + 1), sizes, 512, We, C, confusion=confusion, gain=gain))
    return {"name": f"synthetic NSCLC 2-way 16-shot (32/64/202 slides of ~{mean_rows} rows, confusion {confusion}, gain {gain}), seed {seed}",
            "C": C, "j": j, "K": K, "seed": seed, "W": W, "We": We, "splits": splits, "repeat": [None, None, None], "reference": None}


def study_loop(task):
    base, model32, loaders32, args = run_loop(task, "fp32")
    row = {"kind": "loop", "task": task["name"], "storage": "fp32", **{k: base[k] for k in ("best_val", "test_at_best_val", "best_epoch")},
           "val_auc": [round(v, 4) for v in base["val_auc"]], "reference_main": task["reference"] and {k: task["reference"][k] for k in ("best_val", "test_at_best_val", "best_epoch")}}
    print(json.dumps(row), flush=True)
    p32 = pooled_logits(model32, loaders32[2], args)
    ref_vals = task["reference"]["val_auc"] if task["reference"] else base["val_auc"]
    ref_best = task["reference"] or base
    for dt in ("bf16", "fp16"):
        r, _, _, _ = run_loop(task, dt)
        # the fp32-trained model on 16-bit copies of the test bags: what rounding alone does to the pooled logits
        te16 = M.ResidentBags(*task["splits"][2], DEV, dtype=DTYPES[dt])
        p16 = pooled_logits(model32, te16, args)
        del te16
        out = {"kind": "loop", "task": task["name"], "storage": dt, "best_val": r["best_val"], "test_at_best_val": r["test_at_best_val"],
               "best_epoch": r["best_epoch"],
               "max_abs_d_val_auc_per_epoch": round(float(np.max(np.abs(np.asarray(r["val_auc"]) - np.asarray(ref_vals)))), 6),
               "d_best_val": round(r["best_val"] - ref_best["best_val"], 6),
               "d_test_at_best_val": round(r["test_at_best_val"] - ref_best["test_at_best_val"], 6),
               "same_best_epoch": r["best_epoch"] == ref_best["best_epoch"],
               "max_abs_d_pooled_logit_same_model": float((p16 - p32).abs().max()),
               "compared_with": "reference main()" if task["reference"] else "fp32 storage run"}
        print(json.dumps(out), flush=True)


def study_evaluation_fixtures():
    import helpers as H
    g = H.golden("evaluation")
    for cid, ns, N, C, j, K, dmask, repeat_num, seed in g["cases"]:
        ns, N, C, j, K = int(ns), int(N), int(C), int(j), int(K)
        W, We = synth.make_bank(seed, 512, C)
        bags, labels = synth.make_slide_set(seed + 100, [N] * ns, 512, We, C)
        M.set_classifier_bank(W.to(DEV), We.to(DEV))
        args = _args(C, j, K, H.discard_from_mask(dmask))
        exp = [float(v) for v in g[f"c{cid}_eval"]]
        for dt in ("fp32", "bf16", "fp16"):
            torch.manual_seed(int(seed))
            model = M.senet(512, 4).to(DEV)
            res = M.ResidentBags(bags, labels, DEV, dtype=DTYPES[dt], repeat_num=int(repeat_num) or None)
            got = M.evaluation(model, res, DEV, args)
            print(json.dumps({"kind": "evaluation fixture", "case": int(cid), "slides": ns, "rows": N, "C": C, "topj": j, "topk": K,
                              "storage": dt, "d_loss": got["loss"] - exp[0], "d_acc": got["acc"] - exp[1], "d_auc": got["auc"] - exp[2],
                              "reference": exp}), flush=True)


def main():
    study_evaluation_fixtures()
    tasks = [fixture_task(0), fixture_task(1)]
    if "--quick" not in sys.argv:
        tasks += ([nsclc_task(s) for s in (31000, 31001)] + [nsclc_task(s, confusion=0.40, gain=0.16) for s in (31010, 31011)] +
                  [nsclc_task(s, confusion=0.30, gain=0.20) for s in (31020, 31021)] +
                  [nsclc_task(31030, mean_rows=15000, confusion=0.47, gain=0.12)])
    for t in tasks:
        study_loop(t)


if __name__ == "__main__":
    main()
