#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python.

Run in the build container only (needs /root/reference; never on the GPU box):

    python -B tests/golden/make_golden.py

The reference cannot travel, so what is committed is data: seeds/configs of the
synthetic inputs (moc_amd.synth, numpy Philox) plus the outputs the reference
produced for them.  tests/test_oracle_golden.py pins oracle/moc_oracle.py to
these files; the -m gpu tests compare the HIP path with them.

How the reference is reached:
  * utils/patch_selection_classifier{,_index}.py import as-is (torch only);
  * main_moc.py has import-time side effects (argparse, checkpoint load), so
    the definitions senet / slide_process / train / zs_evaluation / evaluation /
    ablation_evaluation are taken from its AST and exec'd unmodified in a
    namespace holding the names they use.
"""
import ast
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from sklearn.metrics import roc_auc_score
from tqdm import tqdm

from moc_amd import synth

from utils.patch_selection_classifier_index import (  # noqa: E402  (reference)
    index_topj_classifier, index_delta_diff_classifier,
    index_delta_softmax_classifier, index_bottomk_irrel_classifier)
from utils.patch_selection_classifier import (  # noqa: E402  (reference)
    topj_pooling, delta_softmax_classifier_pooling, delta_diff_classifier_pooling,
    bottomk_irrel_classifier_pooling)

torch.set_num_threads(1)   # fixed reduction order inside aten on this host

WANTED = {"senet", "slide_process", "train", "zs_evaluation", "evaluation", "ablation_evaluation", "main"}


def load_reference_main():
    tree = ast.parse(open(os.path.join(REF, "main_moc.py")).read())
    body = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in WANTED]
    import json
    ns = dict(torch=torch, nn=nn, F=F, np=np, tqdm=tqdm, roc_auc_score=roc_auc_score, os=os, json=json,
              index_topj_classifier=index_topj_classifier,
              index_delta_diff_classifier=index_delta_diff_classifier,
              index_delta_softmax_classifier=index_delta_softmax_classifier,
              index_bottomk_irrel_classifier=index_bottomk_irrel_classifier,
              topj_pooling=topj_pooling,
              delta_softmax_classifier_pooling=delta_softmax_classifier_pooling,
              delta_diff_classifier_pooling=delta_diff_classifier_pooling,
              bottomk_irrel_classifier_pooling=bottomk_irrel_classifier_pooling)
    exec(compile(ast.Module(body=body, type_ignores=[]), "main_moc.py", "exec"), ns)
    return ns


class FakeDataset:
    """The attributes evaluation()/zs_evaluation() touch (dataset_generic.py:380-393)."""

    def __init__(self, bags, labels, repeat_num=None):
        self.bags, self.labels, self.repeat_num = bags, labels, repeat_num

    def real_len(self):
        return len(self.bags)

    def __len__(self):
        return self.repeat_num if self.repeat_num else len(self.bags)


class FakeLoader:
    """batch_size=1 default-collate items, as main_moc.py:381-385 unpacks them."""

    def __init__(self, bags, labels, repeat_num=None):
        self.dataset = FakeDataset(bags, labels, repeat_num)

    def __iter__(self):
        ds = self.dataset
        for i in range(len(ds)):
            k = i % ds.real_len()
            x = ds.bags[k]
            yield (x.unsqueeze(0), torch.tensor([ds.labels[k]]),
                   torch.zeros(1, x.size(0), 2), [f"slide_{k}.h5"])

    def __len__(self):
        return len(self.dataset)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def i32(t):
    return t.to(torch.int32).numpy()


# --------------------------------------------------------------------------
def gen_selectors():
    """(1) the four selectors on seeded bags; index tensors value-ordered."""
    out = {}
    cases = []
    cid = 0
    for N in (128, 1000, 4096):
        for C in (2, 3, 30):
            for j in (10, 400):
                seed = 7000 + cid
                W, We = synth.make_bank(seed, 512, C)
                x = synth.make_bag(seed + 500, N, 512, We, C, label=cid % C)
                lg, lge = x @ W, x @ We
                out[f"c{cid}_top"] = i32(index_topj_classifier(lg, [j]))
                out[f"c{cid}_softmax"] = i32(index_delta_softmax_classifier(lg, [j]))
                out[f"c{cid}_gap"] = i32(index_delta_diff_classifier(lg, [j]))
                out[f"c{cid}_lowbg"] = i32(index_bottomk_irrel_classifier(lge, [j], C))
                out[f"c{cid}_logits_ext"] = lge.numpy()
                cases.append((cid, N, C, j, seed))
                cid += 1
    out["cases"] = np.array(cases, dtype=np.int64)
    save("selectors", **out)


def gen_slide_process(ref):
    """(2) slide_process dicts: no mask, reference-drawn mask, discard subsets."""
    out, cases = {}, []
    cfgs = [  # (N, C, j, random_mask, discard)
        (1000, 2, 100, False, []),
        (1000, 2, 100, True, []),
        (3000, 3, 400, True, []),
        (300, 2, 400, True, []),          # N' < j  -> every kept row selected
        (2000, 30, 50, False, []),
        (1500, 2, 100, False, ["delta_diff"]),
        (1500, 2, 100, True, ["topk", "bottomk"]),
        (1500, 3, 100, False, ["delta_softmax", "delta_diff", "bottomk"]),
        (257, 2, 10, True, []),
    ]
    for cid, (N, C, j, rm, discard) in enumerate(cfgs):
        seed = 8000 + cid
        W, We = synth.make_bank(seed, 512, C)
        x = synth.make_bag(seed + 500, N, 512, We, C, label=cid % C)
        torch.manual_seed(seed)
        r = ref["slide_process"](x, W, We, C, topj=j, random_mask=rm, discard_classifiers=list(discard))
        torch.manual_seed(seed)
        mask = (torch.rand(N) > 0.5) if rm else torch.ones(N, dtype=torch.bool)
        kept = x[mask]
        assert torch.equal(kept[r["selected_index"]], r["selected_feat"])
        out[f"c{cid}_mask"] = np.packbits(mask.numpy())
        out[f"c{cid}_selected_index"] = np.asarray(r["selected_index"], dtype=np.int32)
        out[f"c{cid}_top"] = r["logits_top_classifier"].numpy()
        out[f"c{cid}_softmax"] = r["logits_delta_softmax_classifier"].numpy()
        out[f"c{cid}_gap"] = r["logits_delta_diff_classifier"].numpy()
        out[f"c{cid}_lowbg"] = r["logits_bottomk_irrel_classifier"].numpy()
        dmask = sum(1 << k for k, n in enumerate(("topk", "delta_softmax", "delta_diff", "bottomk")) if n in discard)
        cases.append((cid, N, C, j, int(rm), dmask, seed))
    out["cases"] = np.array(cases, dtype=np.int64)
    save("slide_process", **out)


def gen_pooling():
    """(3) topj_pooling and the three zs pooling variants, K in {1,10}, S<K edge."""
    out, cases = {}, []
    cid = 0
    for N, C in ((5, 2), (64, 3), (2000, 2), (2000, 30)):
        seed = 9000 + cid
        W, We = synth.make_bank(seed, 512, C)
        x = synth.make_bag(seed + 500, N, 512, We, C, label=0)
        lg, lge = x @ W, x @ We
        for K in (1, 10):
            out[f"c{cid}_K{K}_topj"] = topj_pooling(lg, [K])[1][K].numpy()
            out[f"c{cid}_K{K}_softmax"] = delta_softmax_classifier_pooling(lg, [K])[1][K].numpy()
            out[f"c{cid}_K{K}_gap"] = delta_diff_classifier_pooling(lg, [K])[1][K].numpy()
            out[f"c{cid}_K{K}_lowbg"] = bottomk_irrel_classifier_pooling(lge, [K], coords_list=C)[1][K].numpy()
        p, pooled, idx = topj_pooling(lg, [10], return_indices=True)
        out[f"c{cid}_topj_idx"] = i32(idx)
        out[f"c{cid}_topj_pred"] = i32(p[10])
        cases.append((cid, N, C, seed))
        cid += 1
    out["cases"] = np.array(cases, dtype=np.int64)
    save("pooling", **out)


def gen_detection():
    """`detection=True` of the two low-background helpers (index.py:65-68, :83-84; classifier.py:146-149, :161-162): one
    foreground column, rows also ranked by their largest background logit; with and without an explicit bottomk."""
    out, cases = {}, []
    cid = 0
    for N, Ct, j, bottomk in ((40, 5, 10, None), (1000, 6, 10, None), (1000, 6, 100, 300), (3000, 34, 400, None), (7, 3, 10, None)):
        seed = 9500 + cid
        W, We = synth.make_bank(seed, 512, Ct - 4 if Ct > 5 else 2)
        x = synth.make_bag(seed + 500, N, 512, We, We.size(1) - 4 if Ct > 5 else 2, label=0)
        lge = (x @ We)[:, :Ct].contiguous()
        kw = {} if bottomk is None else {"bottomk": bottomk}
        out[f"c{cid}_logits_ext"] = lge.numpy()
        out[f"c{cid}_idx"] = i32(index_bottomk_irrel_classifier(lge, [j], 1, detection=True, **kw))
        p, pooled, idx = bottomk_irrel_classifier_pooling(lge, [1, j], return_indices=True, coords_list=1, detection=True, **kw)
        out[f"c{cid}_pool_idx"] = i32(idx)
        out[f"c{cid}_pooled_1"] = pooled[1].numpy()
        out[f"c{cid}_pooled_j"] = pooled[j].numpy()
        out[f"c{cid}_pred_j"] = i32(p[j])
        cases.append((cid, N, Ct, j, -1 if bottomk is None else bottomk, seed))
        cid += 1
    out["cases"] = np.array(cases, dtype=np.int64)
    save("detection", **out)


def _args(C, j, K, discard=()):
    return types.SimpleNamespace(disable_tqdm=True, n_classes=C, topj=j, topk=K,
                                 discard_classifiers=list(discard), pretrain="conch",
                                 ablation_study="none")


def flat_params(model):
    return torch.cat([p.detach().reshape(-1) for p in model.parameters()]).numpy().copy()


def flat_state(opt, key):
    return torch.cat([opt.state[p][key].reshape(-1) for g in opt.param_groups for p in g["params"]]).numpy().copy()


def gen_train(ref):
    """(4) consecutive train steps from a seeded senet."""
    out, cases = {}, []
    cfgs = [  # (n_slides, N, C, j, K, discard)
        (5, 1200, 2, 100, 10, []),
        (6, 1500, 3, 400, 10, []),
        (4, 256, 2, 400, 10, []),            # cfg-1 like: N' < j
        (4, 1000, 2, 100, 10, ["delta_diff"]),
        (30, 800, 30, 50, 10, []),
    ]
    for cid, (ns, N, C, j, K, discard) in enumerate(cfgs):
        seed = 10000 + cid
        W, We = synth.make_bank(seed, 512, C)
        bags, labels = synth.make_slide_set(seed + 100, [N] * ns, 512, We, C)
        ref["zeroshot_weights"], ref["zeroshot_weights_ext"] = W, We
        torch.manual_seed(seed)
        model = ref["senet"](512, 4)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
        out[f"c{cid}_init"] = flat_params(model)
        # one step at a time so per-step state can be recorded; the RNG stream is
        # the same as one train() over all slides (nothing else draws from it)
        torch.manual_seed(seed + 1)
        masks = []
        st = torch.get_rng_state()
        for b in bags:
            masks.append(torch.rand(b.size(0)) > 0.5)
        torch.set_rng_state(st)
        for s in range(ns):
            loader = FakeLoader([bags[s]], [labels[s]])
            ref["train"](model, loader, opt, "cpu", _args(C, j, K, discard))
            if s == 0:
                out[f"c{cid}_grad1"] = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).numpy().copy()
            if s in (0, ns - 1):
                out[f"c{cid}_params_s{s}"] = flat_params(model)
                out[f"c{cid}_m_s{s}"] = flat_state(opt, "exp_avg")
                out[f"c{cid}_v_s{s}"] = flat_state(opt, "exp_avg_sq")
        # per-step loss / pooled logits: one-pass rerun from the same init with
        # F.cross_entropy observed (also checks stepwise == one train() call)
        torch.manual_seed(seed)
        model2 = ref["senet"](512, 4)
        opt2 = torch.optim.Adam(model2.parameters(), lr=1e-3, weight_decay=1e-4)
        torch.manual_seed(seed + 1)
        rec = []
        orig = F.cross_entropy

        def hooked(inp, tgt, *a, **k):
            v = orig(inp, tgt, *a, **k)
            rec.append((float(v), inp.detach().numpy().copy()))
            return v
        F.cross_entropy = hooked
        try:
            ref["train"](model2, FakeLoader(bags, labels), opt2, "cpu", _args(C, j, K, discard))
        finally:
            F.cross_entropy = orig
        assert np.array_equal(flat_params(model2), out[f"c{cid}_params_s{ns - 1}"]), "stepwise != one-pass"
        out[f"c{cid}_loss"] = np.array([r[0] for r in rec], dtype=np.float64)
        out[f"c{cid}_pooled"] = np.concatenate([r[1] for r in rec], 0)
        out[f"c{cid}_masks"] = np.concatenate([np.packbits(m.numpy()) for m in masks])
        dmask = sum(1 << k for k, n in enumerate(("topk", "delta_softmax", "delta_diff", "bottomk")) if n in discard)
        cases.append((cid, ns, N, C, j, K, dmask, seed))
    out["cases"] = np.array(cases, dtype=np.int64)
    save("train", **out)


def gen_eval(ref):
    """(5) evaluation / zs_evaluation / ablation_evaluation dicts."""
    out, cases = {}, []
    cfgs = [(16, 900, 2, 100, 10, [], None), (18, 700, 3, 400, 10, [], 12),
            (16, 900, 2, 100, 10, ["delta_softmax"], None), (60, 400, 30, 20, 10, [], None)]
    for cid, (ns, N, C, j, K, discard, repeat_num) in enumerate(cfgs):
        seed = 11000 + cid
        W, We = synth.make_bank(seed, 512, C)
        bags, labels = synth.make_slide_set(seed + 100, [N] * ns, 512, We, C)
        ref["zeroshot_weights"], ref["zeroshot_weights_ext"] = W, We
        torch.manual_seed(seed)
        model = ref["senet"](512, 4)
        out[f"c{cid}_init"] = flat_params(model)
        a = _args(C, j, K, discard)
        ev = ref["evaluation"](model, FakeLoader(bags, labels, repeat_num), "cpu", a)
        out[f"c{cid}_eval"] = np.array([ev["loss"], ev["acc"], ev["auc"]], dtype=np.float64)
        pools = (("topj", topj_pooling), ("delta_softmax", delta_softmax_classifier_pooling),
                 ("delta_diff", delta_diff_classifier_pooling), ("bottomk", bottomk_irrel_classifier_pooling))
        for name, fn in pools:
            zs = ref["zs_evaluation"](FakeLoader(bags, labels, repeat_num), "cpu", a, pooling_func=fn)
            out[f"c{cid}_zs_{name}"] = np.array([zs["loss"], zs["acc"], zs["auc"]], dtype=np.float64)
        for mode in ("avg", "sum", "max"):
            a.ablation_study = mode
            ab = ref["ablation_evaluation"](FakeLoader(bags, labels, repeat_num), "cpu", a)
            out[f"c{cid}_abl_{mode}"] = np.array([ab["loss"], ab["acc"], ab["auc"]], dtype=np.float64)
        dmask = sum(1 << k for k, n in enumerate(("topk", "delta_softmax", "delta_diff", "bottomk")) if n in discard)
        cases.append((cid, ns, N, C, j, K, dmask, repeat_num or 0, seed))
    out["cases"] = np.array(cases, dtype=np.int64)
    save("evaluation", **out)


def gen_driver(ref):
    """(6) the reference's main(): zero-shot evals, 25 epochs, best-val bookkeeping, result JSON."""
    import json
    import tempfile
    out, cases = {}, []
    cfgs = [(8, 12, 16, 2, 100, 10, 8), (6, 12, 15, 3, 60, 10, 9)]   # n_train, n_val, n_test, C, topj, topk, repeat_num
    for cid, (ntr, nva, nte, C, j, K, rep) in enumerate(cfgs):
        seed = 12000 + cid
        W, We = synth.make_bank(seed, 512, C)
        sets = []
        for s_i, n in enumerate((ntr, nva, nte)):
            sizes = synth.bag_sizes(seed + 10 * s_i, n, 400, fixed=False, lo=150, hi=900)
            sets.append(synth.make_slide_set(seed + 1000 * (s_i + 1), sizes, 512, We, C, confusion=0.47, gain=0.12))
        ref["zeroshot_weights"], ref["zeroshot_weights_ext"] = W, We
        ref["train_loader"] = FakeLoader(*sets[0], rep)
        ref["val_loader"] = FakeLoader(*sets[1])
        ref["test_loader"] = FakeLoader(*sets[2])
        ref["device"] = "cpu"
        torch.manual_seed(seed)
        ref["model"] = ref["senet"](512, 4)
        ref["optimizer"] = torch.optim.Adam(ref["model"].parameters(), lr=1e-3, weight_decay=1e-4)
        val_aucs, orig_eval = [], ref["evaluation"]

        def logged(model, loader, device, args, _o=orig_eval):
            r = _o(model, loader, device, args)
            if loader is ref["val_loader"]:
                val_aucs.append(r["auc"])
            return r
        ref["evaluation"] = logged
        with tempfile.TemporaryDirectory() as td:
            a = _args(C, j, K)
            a.result_dir, a.shot, a.fold, a.check_zeroshot = td, 4, 0, True
            torch.manual_seed(seed + 1)
            ref["main"](a)
            res = json.load(open(os.path.join(td, "best_results_shot_4_fold_0.json")))
            zs = json.load(open(os.path.join(td, "zs_results_shot_4_fold_0.json")))
            best_state = torch.load(res["best_model_path"])
        ref["evaluation"] = orig_eval
        out[f"c{cid}_result"] = np.array([res["best_val"], res["test_at_best_val"], res["test_acc_at_best_val"], res["best_epoch"]])
        out[f"c{cid}_zs"] = np.array([[zs[k]["loss"], zs[k]["acc"], zs[k]["auc"]] for k in ("zs_train", "zs_val", "zs_test")])
        out[f"c{cid}_val_auc"] = np.array(val_aucs)
        out[f"c{cid}_best_params"] = torch.cat([v.reshape(-1) for v in best_state.values()]).numpy()
        out[f"c{cid}_final_params"] = flat_params(ref["model"])
        for s_i, (bags, _) in enumerate(sets):
            out[f"c{cid}_sizes{s_i}"] = np.array([b.size(0) for b in bags])
        cases.append((cid, ntr, nva, nte, C, j, K, rep, seed))
    out["cases"] = np.array(cases, dtype=np.int64)
    save("driver", **out)


def summary_inputs(td, kind):
    """Deterministic fake result files for --summary: kind in {full, nozs, ablation}."""
    import json
    rs = np.random.RandomState(5)
    for shot in (1, 2, 4, 8):
        d = os.path.join(td, f"{shot}_shot")
        os.makedirs(d, exist_ok=True)
        for fold in range(5):
            v = rs.rand(6).round(6).tolist()
            if kind == "ablation":
                obj, name = {"loss": v[0], "acc": v[1], "auc": v[2]}, f"ablation_results_avg_shot_{shot}_fold_{fold}.json"
            else:
                obj = {"zero_shot_train": -1, "zero_shot_val": -1,
                       "zero_shot_test": {"loss": v[0], "acc": v[1], "auc": v[2]} if kind == "full" else -1,
                       "best_val": v[3], "test_at_best_val": v[4], "test_acc_at_best_val": v[5], "best_epoch": fold}
                name = f"best_results_shot_{shot}_fold_{fold}.json"
            json.dump(obj, open(os.path.join(d, name), "w"))


def gen_summary():
    """(7) --summary: the reference's top-level `if args.summary:` block (main_moc.py:53-127), run on fake results."""
    import glob as _glob
    import json
    import tempfile
    import pandas as pd
    tree = ast.parse(open(os.path.join(REF, "main_moc.py")).read())
    node = [n for n in tree.body if isinstance(n, ast.If) and "summary" in ast.unparse(n.test)][0]
    body = [n for n in node.body if not (isinstance(n, ast.Expr) and "exit" in ast.unparse(n))]
    code = compile(ast.Module(body=body, type_ignores=[]), "main_moc.py:summary", "exec")
    out = {}
    for kind in ("full", "nozs", "ablation"):
        with tempfile.TemporaryDirectory() as td:
            summary_inputs(td, kind)
            exec(code, dict(args=types.SimpleNamespace(summary_dir=td), os=os, json=json, np=np, pd=pd, glob=_glob.glob))
            for shot in (1, 2, 4, 8):
                df = pd.read_csv(os.path.join(td, f"summary_{shot}.csv"))
                out[f"{kind}_{shot}_cols"] = np.array(list(df.columns))
                out[f"{kind}_{shot}_vals"] = df.drop(columns=["fold"]).to_numpy(dtype=np.float64)
    save("summary", **out)


# ----------------------------------------------------------------------------------------------
# the baseline models and trainer hooks.  models/model_mil.py imports
# nystrom_attention, models/model_adapters.py imports openslide and utils/core_utils.py imports every
# model family -- none importable here -- so the definitions are taken from the AST, as for main_moc.py.
def _extract(path, names, ns):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    body = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert {n.name for n in body} == set(names), (path, names)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


def load_reference_baselines():
    from sklearn.metrics import auc as calc_auc
    from sklearn.metrics import roc_curve
    from sklearn.preprocessing import label_binarize
    base = dict(torch=torch, nn=nn, F=F, np=np, os=os)
    util = _extract("utils/utils.py", ["detect_nan", "calculate_error"], dict(base))
    mil = _extract("models/model_mil.py", ["initialize_weights", "MIL_fc", "MIL_fc_mc"], dict(base))
    ada = _extract("models/model_adapters.py",
                   ["Linear_Adapter", "uncertainty", "Conch_CLIP_Ada", "Conch_TIP_Ada", "load_balancing_loss_func",
                    "SwitchGate", "Conch_MOE_CLIP_Ada", "Conch_AMUVanilla_Ada", "Conch_AMUTip_Ada"],
                   dict(base, detect_nan=util["detect_nan"]))
    class _Np:                       # utils/core_utils.py:71 says `np.Inf`, an alias NumPy 2 dropped: same constant
        Inf = np.inf

        def __getattr__(self, k):
            return getattr(np, k)
    core = _extract("utils/core_utils.py", ["Accuracy_Logger", "EarlyStopping", "train_loop", "validate", "summary"],
                    dict(base, np=_Np(), calculate_error=util["calculate_error"], roc_auc_score=roc_auc_score, roc_curve=roc_curve,
                         calc_auc=calc_auc, label_binarize=label_binarize))
    return mil, ada, core


sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers_baselines import (BASELINE_CASES, HOOK_CASES, EARLY_SEQS, CLAM_CASES, CLAM_HOOK_CASES, build_case, run_case,  # noqa: E402
                               run_clam_case, run_clam_hooks, psig as _psig, randn as _randn, hook_bags, Loader)


def gen_clam():
    """the reference's CLAM_SB / CLAM_MB (models/model_clam.py imports
    utils.utils, which imports h5py-era helpers: AST extraction again)."""
    base = dict(torch=torch, nn=nn, F=F, np=np, os=os)
    util = _extract("utils/utils.py", ["initialize_weights"], dict(base))
    ns = _extract("models/model_clam.py", ["Attn_Net", "Attn_Net_Gated", "CLAM_SB", "CLAM_MB"],
                  dict(base, initialize_weights=util["initialize_weights"]))
    arrays = {"cases": np.asarray([c[0] for c in CLAM_CASES])}
    for i, (name, kind, kw, N, label, fkw) in enumerate(CLAM_CASES):
        for k, v in run_clam_case(ns, kind, kw, N, label, fkw, 7000 + 13 * i).items():
            arrays[f"{name}:{k}"] = v
    save("clam", **arrays)


def gen_clam_hooks():
    """SURVEY.md section 8 rows f3 x f4: the reference's train_loop_clam / validate_clam (utils/core_utils.py:294-370,
    :558-656) driving the reference's CLAM_SB / CLAM_MB, three epochs with early stopping + summary."""
    import tempfile
    from sklearn.metrics import auc as calc_auc
    from sklearn.metrics import roc_curve
    from sklearn.preprocessing import label_binarize
    base = dict(torch=torch, nn=nn, F=F, np=np, os=os)
    util = _extract("utils/utils.py", ["initialize_weights", "calculate_error"], dict(base))
    clam = _extract("models/model_clam.py", ["Attn_Net", "Attn_Net_Gated", "CLAM_SB", "CLAM_MB"],
                    dict(base, initialize_weights=util["initialize_weights"]))

    class _Np:                       # utils/core_utils.py:71 says `np.Inf`, an alias NumPy 2 dropped: same constant
        Inf = np.inf

        def __getattr__(self, k):
            return getattr(np, k)
    core = _extract("utils/core_utils.py", ["Accuracy_Logger", "EarlyStopping", "train_loop_clam", "validate_clam", "summary"],
                    dict(base, np=_Np(), calculate_error=util["calculate_error"], roc_auc_score=roc_auc_score, roc_curve=roc_curve,
                         calc_auc=calc_auc, label_binarize=label_binarize))
    arrays = {"cases": np.asarray([c[0] for c in CLAM_HOOK_CASES])}
    for tag, kind, kw, bw in CLAM_HOOK_CASES:
        with tempfile.TemporaryDirectory() as td:
            for k, v in run_clam_hooks(core, clam, tag, kind, kw, bw, "cpu", td).items():
                arrays[f"{tag}:{k}"] = v
    save("clam_hooks", **arrays)


def gen_baselines():
    mil, ada, core = load_reference_baselines()
    arrays = {"cases": np.asarray([c[0] for c in BASELINE_CASES])}
    for i, (name, kind, kw, N, label) in enumerate(BASELINE_CASES):
        seed = 4000 + 17 * i
        ns = mil if kind.startswith("MIL") else ada
        cls, kwargs = build_case(ns, kind, kw, seed)
        model = cls(**kwargs)
        arrays[f"{name}:psig"] = _psig(model)
        for k, v in run_case(model, kind, N, label, seed).items():
            arrays[f"{name}:{k}"] = v
    # ---- trainer hooks: one epoch of train_loop, validate with early stopping, summary
    import contextlib
    import io
    import tempfile
    import pandas as pd

    for tag, kind, kw, d, C in HOOK_CASES:
        torch.manual_seed(77)
        model = mil[kind](**kw)
        opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-5)
        tr, va = Loader(hook_bags(6000, 8, d, C)), Loader(hook_bags(6100, 8, d, C))
        va.dataset = types.SimpleNamespace(slide_data=pd.DataFrame({"slide_id": [f"s{k}" for k in range(len(va))]}))
        loss_fn = nn.CrossEntropyLoss()
        with tempfile.TemporaryDirectory() as td, contextlib.redirect_stdout(io.StringIO()):
            stop = core["EarlyStopping"](patience=2, stop_epoch=1, verbose=True)
            trace = []
            for epoch in range(5):
                core["train_loop"](epoch, model, tr, opt, C, None, loss_fn)
                fired = core["validate"](0, epoch, model, va, C, stop, None, loss_fn, td)
                trace.append([float(fired), stop.counter, float(stop.best_score), float(stop.val_loss_min)])
                if fired:
                    break
            res, err, auc, logger = core["summary"](model, va, C)
        arrays[f"{tag}:psig"] = _psig(model)
        arrays[f"{tag}:trace"] = np.asarray(trace)
        arrays[f"{tag}:summary"] = np.asarray([err, auc])
        arrays[f"{tag}:acc"] = np.asarray([[c if c is not None else -1 for c in (logger.get_summary(i)[1], logger.get_summary(i)[2])] for i in range(C)], dtype=np.float64)
        arrays[f"{tag}:probs"] = np.stack([res[f"s{k}"]["prob"].reshape(-1) for k in range(len(va))])
    # ---- EarlyStopping alone, loss-driven and criteria-driven sequences
    class Dummy(nn.Module):
        def __init__(self):
            super().__init__()
            self.w = nn.Parameter(torch.zeros(1))
    for k, seq in EARLY_SEQS.items():
        with tempfile.TemporaryDirectory() as td, contextlib.redirect_stdout(io.StringIO()):
            es = core["EarlyStopping"](patience=2, stop_epoch=3, verbose=True)
            tr = []
            for e, l, c in seq:
                es(e, l, Dummy(), ckpt_name=os.path.join(td, "c.pt"), criteria=c)
                tr.append([es.counter, float(es.best_score), float(es.early_stop), float(es.val_loss_min)])
        arrays[f"early:{k}"] = np.asarray(tr)
        arrays[f"early:{k}:in"] = np.asarray([[e, l, -1.0 if c is None else c] for e, l, c in seq])
    save("baselines", **arrays)


# the split semantics of datasets/dataset_generic.py.  The module imports cv2, h5py and
# utils.utils (torchvision) -- none importable here -- so the dataset classes are taken from the AST like main_moc.py's
# definitions; constructing them and cutting splits needs pandas only (no item is read from a bag file: with
# data_dir=None the reference's __getitem__ returns (slide_id, label), dataset_generic.py:412-413, which is exactly the
# visiting order a shuffle=False DataLoader sees).
SPLIT_TASKS = {"nsclc": ({"LUAD": 0, "LUSC": 1}, 2), "rcc": ({"KICH": 0, "KIRC": 1, "KIRP": 2}, 3)}
SPLIT_SHOTS = (1, 2, 4, 8, 16)
SPLIT_FOLDS = range(5)


def load_reference_datasets():
    from scipy import stats
    from torch.utils.data import Dataset
    import pandas as pd
    ns = dict(torch=torch, np=np, pd=pd, os=os, stats=stats, Dataset=Dataset)
    return _extract("datasets/dataset_generic.py",
                    ["Generic_WSI_Classification_Dataset", "Generic_MIL_Dataset", "Generic_MIL_Dataset_ViLa", "Generic_Split"], ns)


def gen_splits():
    """return_splits(from_id=False, csv_path=..., repeat_num=shot*C) (main_moc.py:209-220, :270-281) on the reference's
    own dataset_csv/{nsclc,rcc}.csv x splits/*_fewshot/{1,2,4,8,16}shots/splits_{0-4}.csv.  Per split: the slides as
    positions in the slide table (in the split's own order), labels, len(), real_len(); for the train split also the
    slide visited at every index 0..len-1 (the repeat_num wrap / truncation) and len() after the evaluation loops'
    `repeat_num = real_len()` (main_moc.py:469-471).  The CSV inputs are copied next to the fixture as data."""
    import shutil
    ref = load_reference_datasets()
    out = {}
    data_root = os.path.join(HERE, "ref_data")
    for task, (label_dict, C) in SPLIT_TASKS.items():
        os.makedirs(os.path.join(data_root, "dataset_csv"), exist_ok=True)
        shutil.copyfile(os.path.join(REF, "dataset_csv", task + ".csv"), os.path.join(data_root, "dataset_csv", task + ".csv"))
        ds = ref["Generic_MIL_Dataset"](csv_path=os.path.join(REF, "dataset_csv", task + ".csv"), data_dir=None, shuffle=False,
                                        seed=1, print_info=False, label_dict=label_dict, patient_strat=False, ignore=[])
        ds.load_from_h5(True)
        ds.load_full_path(True)
        table = ds.slide_data["slide_id"].tolist()
        pos = {s: i for i, s in enumerate(table)}
        assert len(pos) == len(table)
        out[f"{task}/table_ids"] = np.asarray(table)
        out[f"{task}/table_labels"] = np.asarray(ds.slide_data["label"].tolist(), dtype=np.int64)
        out[f"{task}/len"] = np.int64(len(ds))
        for shot in SPLIT_SHOTS:
            for fold in SPLIT_FOLDS:
                rel = os.path.join("splits", f"{task}_fewshot", f"{shot}shots", f"splits_{fold}.csv")
                os.makedirs(os.path.dirname(os.path.join(data_root, rel)), exist_ok=True)
                shutil.copyfile(os.path.join(REF, rel), os.path.join(data_root, rel))
                splits = ds.return_splits(from_id=False, csv_path=os.path.join(REF, rel), repeat_num=int(shot) * C)
                for name, sp in zip(("train", "val", "test"), splits):
                    key = f"{task}/{shot}/{fold}/{name}"
                    sp.load_full_path(True)
                    out[key + "/idx"] = np.asarray([pos[s] for s in sp.slide_data["slide_id"].tolist()], dtype=np.int32)
                    out[key + "/labels"] = np.asarray(sp.slide_data["label"].tolist(), dtype=np.int64)
                    out[key + "/len_real"] = np.asarray([len(sp), sp.real_len()], dtype=np.int64)
                    out[key + "/cls_counts"] = np.asarray([len(c) for c in sp.slide_cls_ids], dtype=np.int64)
                    # what a shuffle=False loader visits: indices 0 .. len-1 (IndexError ends it, :382-383)
                    visit = [sp[i] for i in range(len(sp))]
                    out[key + "/visit"] = np.asarray([pos[v[0]] for v in visit], dtype=np.int32)
                    assert [int(v[1]) for v in visit] == [int(out[f"{task}/table_labels"][pos[v[0]]]) for v in visit]
                    try:
                        sp[len(sp)]
                        stops = False
                    except IndexError:
                        stops = True
                    out[key + "/stops"] = np.bool_(stops)
                    if name == "train":           # the evaluation loops' toggle (main_moc.py:469-471, :499)
                        keep = sp.repeat_num
                        sp.repeat_num = sp.real_len()
                        out[key + "/len_eval"] = np.int64(len(sp))
                        sp.repeat_num = keep
    save("splits", **out)


if __name__ == "__main__":
    only = sys.argv[1:]          # e.g. `make_golden.py driver` regenerates one fixture
    ref = load_reference_main()
    if only:
        for name in only:
            fn = globals()["gen_" + name]
            fn(ref) if fn.__code__.co_argcount else fn()
        sys.exit(0)
    gen_selectors()
    gen_slide_process(ref)
    gen_pooling()
    gen_train(ref)
    gen_eval(ref)
    gen_driver(ref)
    gen_summary()
    gen_baselines()
    gen_clam()
    gen_clam_hooks()
    gen_splits()
    gen_detection()
