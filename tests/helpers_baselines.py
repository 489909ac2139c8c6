"""Shared by tests/golden/make_golden.py (run on the reference's classes) and the baseline tests (run on
moc_amd's): the case table, how a case is constructed and exercised, and the signatures compared."""
import numpy as np
import torch
import torch.nn.functional as F

from moc_amd import synth

def randn(seed, *shape):
    return torch.from_numpy(synth._rng(seed).standard_normal(shape).astype(np.float32))


def psig(model):
    """Order-stable signature of the parameters: (sum, abs-sum, weighted sum) per tensor."""
    out = []
    for _, p in sorted(model.named_parameters()):
        v = p.detach().cpu().double().reshape(-1)
        out.append([float(v.sum()), float(v.abs().sum()), float((v * torch.arange(1, v.numel() + 1, dtype=torch.float64)).sum() / v.numel())])
    return np.asarray(out)


def gsig(model):
    """Gradients: small tensors whole, big ones as their row / column sums."""
    out = {}
    for n, p in sorted(model.named_parameters()):
        g = torch.zeros_like(p) if p.grad is None else p.grad.detach()
        if g.numel() <= 4096:
            out[n] = g.cpu().numpy().copy()
        else:
            out[n + ":rows"] = g.sum(1).cpu().numpy().copy()
            out[n + ":cols"] = g.sum(0).cpu().numpy().copy()
    return out


BASELINE_CASES = [
    # name, kind, ctor kwargs (classifier/sample tensors are built from seeds in build_case), N, label
    ("mil_fc", "MIL_fc", dict(size_arg="benchmark"), 300, 1),
    ("mil_fc_small", "MIL_fc", dict(size_arg="small"), 37, 0),
    ("mil_mc", "MIL_fc_mc", dict(n_classes=4), 200, 2),
    ("clip", "Conch_CLIP_Ada", dict(num_classes=3, topj=10), 400, 1),
    ("clip_short", "Conch_CLIP_Ada", dict(num_classes=2, topj=10), 6, 0),          # N < topj
    ("tip", "Conch_TIP_Ada", dict(num_classes=3), 350, 2),
    ("tip_cache", "Conch_TIP_Ada", dict(num_classes=3, samples=12), 350, 0),
    ("moe", "Conch_MOE_CLIP_Ada", dict(ada_num=3, topj=8, num_classes=3), 300, 1),
    ("moe_switch", "Conch_MOE_CLIP_Ada", dict(ada_num=4, topj=10, num_classes=2, use_switch_gate=True, use_balance_loss=True), 300, 1),
    ("moe_router", "Conch_MOE_CLIP_Ada", dict(ada_num=3, topj=10, num_classes=2, router=True, router_trainable=False), 250, 0),
    ("amu_none", "Conch_AMUVanilla_Ada", dict(num_classes=3), 300, 1),
    ("amu_entropy", "Conch_AMUVanilla_Ada", dict(num_classes=3, uncertainty_type="entropy", uncertainty_power=0.5), 300, 1),
    ("amu_energy", "Conch_AMUVanilla_Ada", dict(num_classes=3, uncertainty_type="energy", uncertainty_power=2.0), 300, 2),
    ("amu_max", "Conch_AMUVanilla_Ada", dict(num_classes=3, uncertainty_type="max"), 300, 0),
    ("amu_maxmin", "Conch_AMUVanilla_Ada", dict(num_classes=3, uncertainty_type="max-min"), 300, 0),
    ("amu_var", "Conch_AMUVanilla_Ada", dict(num_classes=3, uncertainty_type="var"), 300, 0),
    ("amu_top5", "Conch_AMUVanilla_Ada", dict(num_classes=6, uncertainty_type="top5"), 300, 4),
    ("amu_moment", "Conch_AMUVanilla_Ada", dict(num_classes=3, uncertainty_type="moment"), 300, 0),
    ("amutip", "Conch_AMUTip_Ada", dict(num_classes=3, samples=9), 280, 2),
]


def build_case(ns, kind, kw, seed, device="cpu"):
    """Construct `kind` from namespace `ns` (reference or product) the same way for both sides.
    Tensors handed to the constructor live on `device`; the module itself is built on the CPU under
    the seed (same RNG stream on both sides) and moved by the caller."""
    kw = dict(kw)
    C = kw.get("num_classes", kw.get("n_classes", 2))
    if kind.startswith("Conch"):
        _, We = synth.make_bank(seed, 512, C)
        kw["classifier_tensor"] = We[:, :C].contiguous().to(device)
        if kind == "Conch_MOE_CLIP_Ada":
            kw.pop("num_classes")
            if kw.pop("router", False):
                kw["router_tensor"] = (randn(seed + 5, 512, kw["ada_num"]) * 0.05).to(device)
        ns_samples = kw.pop("samples", None)
        if ns_samples:
            lab = [i % C for i in range(ns_samples)]
            kw["sample_features"] = (randn(seed + 1, ns_samples, 512).to(device), lab)
            if kind == "Conch_AMUTip_Ada":
                kw["aux_sample_features"] = (randn(seed + 2, ns_samples, 1024).to(device), lab)
    torch.manual_seed(seed)
    return getattr(ns, kind) if not isinstance(ns, dict) else ns[kind], kw


def run_case(model, kind, N, label, seed, device="cpu"):
    """-> dict of outputs and gradient signatures (numpy, host)."""
    d_in = {"MIL_fc": None, "MIL_fc_mc": 1024}.get(kind, 512)
    if kind == "MIL_fc":
        d_in = model.classifier[0].in_features
    x = randn(seed + 100, N, d_in).to(device)
    out = {}
    y = torch.tensor([label], device=device)
    if kind.startswith("MIL"):
        top, prob, yhat, yprobs, _ = model(x)
        out.update(top=top.detach().cpu().numpy(), prob=prob.detach().cpu().numpy(), yhat=yhat.cpu().numpy().reshape(-1),
                   yprobs=yprobs.detach().cpu().numpy())
        loss = F.cross_entropy(top, y)
    elif "AMU" in kind:
        aux = randn(seed + 200, N, 1024).to(device)
        res = model(x.clone(), aux.clone())
        pooled = res[0] if isinstance(res, tuple) else res
        out["pooled"] = pooled.detach().cpu().numpy()
        loss = F.cross_entropy(pooled, y)
        if isinstance(res, tuple):
            out["pooled_aux"] = res[1].detach().cpu().numpy()
            loss = loss + 0.5 * F.cross_entropy(res[1], y)
        out["zero_shot"] = model.forward_disable_ada(x.clone(), aux.clone()).detach().cpu().numpy()
    else:
        res = model(x.clone())
        pooled = res[0] if isinstance(res, tuple) else res
        out["pooled"] = pooled.detach().cpu().numpy()
        loss = F.cross_entropy(pooled, y)
        if isinstance(res, tuple):
            out["balance"] = np.asarray([float(res[1].detach())])
            loss = loss + 0.1 * res[1]
        out["zero_shot"] = model.forward_disable_ada(x.clone()).detach().cpu().numpy()
    out["loss"] = np.asarray([float(loss.detach())])
    loss.backward()
    for k, v in gsig(model).items():
        out["grad:" + k] = v
    return out



HOOK_CASES = (("hooks2", "MIL_fc", dict(size_arg="benchmark"), 384, 2),
              ("hooks4", "MIL_fc_mc", dict(n_classes=4), 1024, 4))

EARLY_SEQS = {"loss": [(e, l, None) for e, l in enumerate([1.0, 0.9, 0.95, 0.97, 0.8, 0.85, 0.9, 0.95])],
              "crit": [(e, 1.0, c) for e, c in enumerate([0.5, 0.0, 0.6, 0.6, 0.55, 0.7, 0.65, 0.6, 0.5])]}


def hook_bags(seed, n, d, C):
    return [(randn(seed + k, 40 + 7 * k, d), torch.tensor([k % C])) for k in range(n)]


def check_case(got: dict, gold, name: str, atol=2e-5):
    """Every array the reference produced for case `name` against what `got` holds."""
    keys = [k for k in gold.files if k.startswith(name + ":") and k != name + ":psig"]
    assert keys, name
    for k in keys:
        sub = k[len(name) + 1:]
        assert sub in got, f"{name}: missing output {sub}"
        exp, val = gold[k], np.asarray(got[sub])
        if sub in ("yhat", "inst_labels", "inst_preds"):
            assert np.array_equal(val.reshape(-1), exp.reshape(-1)), (name, val, exp)
        else:
            scale = max(1.0, float(np.abs(exp).max()))
            np.testing.assert_allclose(val.reshape(exp.shape), exp, atol=atol * scale, rtol=0, err_msg=f"{name}:{sub}")


class Loader(list):
    """A list of (data, label) with room for the `.dataset` attribute `summary` reads."""


# ---------------------------------------------------------------- row f4: CLAM (gated-attention pooling)
CLAM_CASES = [
    # name, class, ctor kwargs, N, label, forward kwargs
    ("sb_small", "CLAM_SB", dict(size_arg="small", n_classes=2), 300, 1, dict()),
    ("sb_conch_inst", "CLAM_SB", dict(size_arg="conch", n_classes=3, subtyping=True, k_sample=8), 333, 2, dict(instance_eval=True, return_features=True)),
    ("sb_bench_inst", "CLAM_SB", dict(size_arg="benchmark", n_classes=2, k_sample=8), 70, 0, dict(instance_eval=True)),
    ("sb_tiny", "CLAM_SB", dict(size_arg="benchmark", n_classes=2, k_sample=8), 5, 1, dict(instance_eval=True)),      # N < k_sample
    ("mb_small", "CLAM_MB", dict(size_arg="small", n_classes=3), 257, 1, dict()),
    ("mb_conch_inst", "CLAM_MB", dict(size_arg="conch", n_classes=4, subtyping=True), 400, 3, dict(instance_eval=True, return_features=True)),
    ("mb_big", "CLAM_MB", dict(size_arg="big", n_classes=2), 129, 0, dict(instance_eval=True)),
    # attention without gating (Attn_Net, models/model_clam.py:15-33)
    ("sb_nogate", "CLAM_SB", dict(gate=False, size_arg="small", n_classes=2), 210, 1, dict(instance_eval=True)),
    ("mb_nogate", "CLAM_MB", dict(gate=False, size_arg="conch", n_classes=3, subtyping=True), 190, 2, dict(instance_eval=True, return_features=True)),
    # dropout (p = 0.25 after the first layer and inside the attention network), TRAINING mode: the masks come from
    # the seeded CPU generator, so this case pins where the Dropout layers sit -- on the CPU only
    ("sb_dropout_train", "CLAM_SB", dict(size_arg="small", dropout=True, n_classes=2), 150, 1, dict(instance_eval=True)),
    ("mb_nogate_dropout_train", "CLAM_MB", dict(gate=False, size_arg="small", dropout=True, n_classes=2), 140, 0, dict()),
]
CLAM_CPU_ONLY = {"sb_dropout_train", "mb_nogate_dropout_train"}     # torch's dropout masks differ between CPU and GPU generators

# trainer hooks of the CLAM models (utils/core_utils.py:294-370, :558-656): tag, class, ctor kwargs, bag_weight
CLAM_HOOK_CASES = [
    ("hook_clam_sb", "CLAM_SB", dict(size_arg="benchmark", n_classes=2, k_sample=4), 0.7),
    ("hook_clam_mb", "CLAM_MB", dict(size_arg="benchmark", n_classes=3, k_sample=4, subtyping=True), 0.5),
]


def run_clam_hooks(core, clam_ns, tag, kind, kw, bag_weight, device, tmpdir):
    """Three epochs of train_loop_clam + validate_clam (early stopping on the AUC criterion) + summary, the same way
    for the reference's functions (fixture generation) and for moc_amd's (tests).  `core` / `clam_ns`: namespaces or
    dicts holding the functions / classes."""
    import contextlib
    import io
    import os
    import types

    import pandas as pd
    import torch.nn as nn
    get = (lambda ns, k: ns[k]) if isinstance(core, dict) else (lambda ns, k: getattr(ns, k))
    C = kw["n_classes"]
    torch.manual_seed(91)
    model = get(clam_ns, kind)(**kw, instance_loss_fn=nn.CrossEntropyLoss()).to(device)
    opt = torch.optim.Adam(model.parameters(), lr=2e-4, weight_decay=1e-5)
    d = model.attention_net[0].in_features
    tr, va = Loader(hook_bags(6500, 9, d, C)), Loader(hook_bags(6600, 9, d, C))
    va.dataset = types.SimpleNamespace(slide_data=pd.DataFrame({"slide_id": [f"s{k}" for k in range(len(va))]}))
    loss_fn = nn.CrossEntropyLoss()
    with contextlib.redirect_stdout(io.StringIO()):
        stop = get(core, "EarlyStopping")(patience=2, stop_epoch=1, verbose=True)
        trace = []
        for epoch in range(3):
            get(core, "train_loop_clam")(epoch, model, tr, opt, C, bag_weight, None, loss_fn)
            fired = get(core, "validate_clam")(0, epoch, model, va, C, stop, None, loss_fn, str(tmpdir))
            trace.append([float(fired), stop.counter, float(stop.best_score), float(stop.val_loss_min)])
            if fired:
                break
        res, err, auc, logger = get(core, "summary")(model, va, C)
    assert os.path.exists(os.path.join(str(tmpdir), "s_0_checkpoint.pt"))
    # the bias of the attention scores has a gradient that is zero by construction (softmax over the patches is
    # shift invariant): under Adam it performs a random walk on rounding noise -- in the reference as here.  Its psig
    # row is recorded but not compared.
    noise_rows = np.asarray([i for i, (n, _) in enumerate(sorted(model.named_parameters()))
                             if n.endswith("attention_c.bias") or (".module." in n and n.endswith("bias") and n.split(".")[-2] != "0")])
    return dict(psig=psig(model), noise_rows=noise_rows, trace=np.asarray(trace), summary=np.asarray([err, auc]),
                acc=np.asarray([[logger.get_summary(i)[1], logger.get_summary(i)[2]] for i in range(C)], dtype=np.float64),
                probs=np.stack([res[f"s{k}"]["prob"].reshape(-1) for k in range(len(va))]))


def run_clam_case(ns, kind, kw, N, label, fkw, seed, device="cpu"):
    get = (lambda k: ns[k]) if isinstance(ns, dict) else (lambda k: getattr(ns, k))
    torch.manual_seed(seed)
    model = get(kind)(**kw)
    sig = psig(model)
    model = model.to(device)
    d_in = model.attention_net[0].in_features
    x = randn(seed + 100, N, d_in).to(device)
    y = torch.tensor([label], device=device)
    out = {"psig": sig}
    logits, prob, yhat, A_raw, res = model(x, label=y, **fkw)
    out.update(logits=logits.detach().cpu().numpy(), prob=prob.detach().cpu().numpy(), yhat=yhat.cpu().numpy().reshape(-1),
               A_raw=A_raw.detach().cpu().numpy())
    loss = F.cross_entropy(logits, y)
    if fkw.get("instance_eval"):
        out["instance_loss"] = np.asarray([float(res["instance_loss"].detach())])
        out["inst_labels"] = np.asarray(res["inst_labels"]).reshape(-1)
        out["inst_preds"] = np.asarray(res["inst_preds"]).reshape(-1)
        loss = loss + 0.3 * res["instance_loss"]
    if fkw.get("return_features"):
        out["features"] = res["features"].detach().cpu().numpy()
    out["loss"] = np.asarray([float(loss.detach())])
    loss.backward()
    for k, v in gsig(model).items():
        out["grad:" + k] = v
    out["attention_only"] = model(x, attention_only=True).detach().cpu().numpy()
    out["patch_level"] = model.forward_patch_level(x).detach().cpu().numpy()[:16]
    return out
