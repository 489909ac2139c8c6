"""SURVEY.md section 8 row f3 on the CPU: moc_amd's baseline models and trainer hooks against the
fixtures the reference's own classes produced (tests/golden/baselines.npz), with the HIP pooling steps
replaced by their CPU restatement (oracle/baselines_oracle.py).  Pins layers, mixing formulas,
uncertainty factors, constructor RNG order, EarlyStopping and the train / validate / summary loops."""
import contextlib
import io
import os
import types

import numpy as np
import pandas as pd
import pytest
import torch
import torch.nn as nn

import helpers as H
import helpers_baselines as HB
from oracle import baselines_oracle as BO

GOLD = H.golden("baselines")


def _ns(kind):
    import moc_amd.model_adapters as A
    import moc_amd.model_mil as Mm
    return Mm if kind.startswith("MIL") else A


@pytest.mark.parametrize("i", range(len(HB.BASELINE_CASES)), ids=[c[0] for c in HB.BASELINE_CASES])
def test_baseline_models_match_reference_outputs(i):
    name, kind, kw, N, label = HB.BASELINE_CASES[i]
    seed = 4000 + 17 * i
    with BO.patched(), contextlib.redirect_stdout(io.StringIO()):
        cls, kwargs = HB.build_case(_ns(kind), kind, kw, seed)
        model = cls(**kwargs)
        np.testing.assert_allclose(HB.psig(model), GOLD[f"{name}:psig"], rtol=1e-6, atol=1e-7,
                                   err_msg=f"{name}: same seed, different parameters (constructor RNG order)")
        got = HB.run_case(model, kind, N, label, seed)
    HB.check_case(got, GOLD, name)


def run_hooks(core, mil_ns, tag, kind, kw, d, C, device, tmpdir):
    torch.manual_seed(77)
    model = getattr(mil_ns, kind)(**kw).to(device)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-5)
    tr, va = HB.Loader(HB.hook_bags(6000, 8, d, C)), HB.Loader(HB.hook_bags(6100, 8, d, C))
    va.dataset = types.SimpleNamespace(slide_data=pd.DataFrame({"slide_id": [f"s{k}" for k in range(len(va))]}))
    loss_fn = nn.CrossEntropyLoss()
    with contextlib.redirect_stdout(io.StringIO()):
        stop = core.EarlyStopping(patience=2, stop_epoch=1, verbose=True)
        trace = []
        for epoch in range(5):
            core.train_loop(epoch, model, tr, opt, C, None, loss_fn)
            fired = core.validate(0, epoch, model, va, C, stop, None, loss_fn, str(tmpdir))
            trace.append([float(fired), stop.counter, float(stop.best_score), float(stop.val_loss_min)])
            if fired:
                break
        res, err, auc, logger = core.summary(model, va, C)
    assert os.path.exists(os.path.join(str(tmpdir), "s_0_checkpoint.pt"))
    return dict(psig=HB.psig(model), trace=np.asarray(trace), summary=np.asarray([err, auc]),
                acc=np.asarray([[logger.get_summary(i)[1], logger.get_summary(i)[2]] for i in range(C)], dtype=np.float64),
                probs=np.stack([res[f"s{k}"]["prob"].reshape(-1) for k in range(len(va))]))


def check_hooks(got, tag, tol=2e-5):
    np.testing.assert_allclose(got["trace"], GOLD[f"{tag}:trace"], atol=tol, rtol=0, err_msg=f"{tag}: validate / early-stopping trace")
    np.testing.assert_allclose(got["summary"], GOLD[f"{tag}:summary"], atol=tol, rtol=0)
    np.testing.assert_array_equal(got["acc"], GOLD[f"{tag}:acc"])
    np.testing.assert_allclose(got["probs"], GOLD[f"{tag}:probs"], atol=tol, rtol=0)
    np.testing.assert_allclose(got["psig"], GOLD[f"{tag}:psig"], rtol=2e-5, atol=1e-6, err_msg=f"{tag}: parameters after training")


@pytest.mark.parametrize("case", HB.HOOK_CASES, ids=[c[0] for c in HB.HOOK_CASES])
def test_trainer_hooks_match_reference(case, tmp_path, monkeypatch):
    import moc_amd.core_utils as core
    import moc_amd.model_mil as Mm
    monkeypatch.setattr(core, "_device", lambda: torch.device("cpu"))     # the product insists on a GPU; this is the CPU pin
    with BO.patched():
        got = run_hooks(core, Mm, *case, torch.device("cpu"), tmp_path)
    check_hooks(got, case[0])


@pytest.mark.parametrize("kind", list(HB.EARLY_SEQS))
def test_early_stopping_sequences(kind, tmp_path):
    import moc_amd.core_utils as core

    class Dummy(nn.Module):
        def __init__(self):
            super().__init__()
            self.w = nn.Parameter(torch.zeros(1))
    with contextlib.redirect_stdout(io.StringIO()):
        es = core.EarlyStopping(patience=2, stop_epoch=3, verbose=True)
        tr = []
        for e, l, c in HB.EARLY_SEQS[kind]:
            es(e, l, Dummy(), ckpt_name=os.path.join(str(tmp_path), "c.pt"), criteria=c)
            tr.append([es.counter, float(es.best_score), float(es.early_stop), float(es.val_loss_min)])
    np.testing.assert_allclose(np.asarray(tr), GOLD[f"early:{kind}"], atol=1e-12, rtol=0)


def test_nystrom_attention_restatement_limits():
    """moc_amd/nystrom.py restates a third-party layer no fixture can pin (the package is absent everywhere).  What can
    be checked: with one token per landmark and a converged pseudo-inverse it IS softmax attention; the iteration
    converges to torch.linalg.pinv; front padding keeps the last n tokens; parameter names are the package's."""
    from moc_amd.nystrom import NystromAttention, moore_penrose_iter_pinv
    torch.manual_seed(0)
    a = torch.softmax(4.0 * torch.randn(2, 3, 12, 12), -1) + 0.5 * torch.eye(12)      # well conditioned
    np.testing.assert_allclose(moore_penrose_iter_pinv(a, 30).numpy(), torch.linalg.pinv(a).numpy(), atol=2e-4)
    att = NystromAttention(dim=64, dim_head=8, heads=8, num_landmarks=32, pinv_iterations=30, residual=False).eval()
    x = torch.randn(2, 32, 64)
    with torch.no_grad():
        y = att(x)
        q, k, v = (t.reshape(2, 32, 8, 8).transpose(1, 2) for t in att.to_qkv(x).chunk(3, -1))
        ref = att.to_out((torch.softmax(q * att.scale @ k.transpose(-1, -2), -1) @ v).transpose(1, 2).reshape(2, 32, 64))
    np.testing.assert_allclose(y.numpy(), ref.numpy(), atol=2e-4)
    att = NystromAttention(dim=64, dim_head=8, heads=8, num_landmarks=16, residual=True).eval()
    with torch.no_grad():
        assert att(torch.randn(1, 37, 64)).shape == (1, 37, 64)                       # 37 -> padded to 48 at the front
    assert sorted(att.state_dict()) == ["res_conv.weight", "to_out.0.bias", "to_out.0.weight", "to_qkv.weight"]
    assert att.res_conv.weight.shape == (8, 1, 33, 1)


def test_transmil_keeps_the_reference_contract():
    """models/model_mil.py:142-273: module / parameter names, the 5-tuple, wrap-around padding to a square, the patch-
    level head -- parity with a reference NUMBER is unpinned (see moc_amd/nystrom.py)."""
    from moc_amd.model_mil import TransMIL
    torch.manual_seed(3)
    m = TransMIL(n_classes=3, size_arg="conch", dropout=True).eval()
    names = set(m.state_dict())
    assert {"cls_token", "_fc1.0.weight", "_fc2.bias", "norm.weight", "pos_layer.proj.weight", "pos_layer.proj1.bias",
            "pos_layer.proj2.weight", "layer1.norm.weight", "layer1.attn.to_qkv.weight", "layer2.attn.to_out.0.bias",
            "layer2.attn.res_conv.weight"} <= names and len(names) == 25
    x = torch.randn(300, 512)
    with torch.no_grad():
        logits, prob, yhat, a, b = m(x)
        again = m(x.unsqueeze(0))[0]
        patch = m.forward_patch_level(x)
    assert logits.shape == (1, 3) and prob.shape == (1, 3) and yhat.shape == (1,) and a is None and b is None
    assert torch.allclose(prob.sum(), torch.tensor(1.0)) and torch.equal(logits, again) and patch.shape == (300, 3)
    m.train()
    out = m(torch.randn(50, 512))[0]
    out.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_product_pooling_refuses_cpu_tensors():
    from moc_amd.pool_autograd import topk_mean_pool
    with pytest.raises(AssertionError):
        topk_mean_pool(torch.zeros(5, 2), 3)


# ---------------------------------------------------------------- row f4: CLAM around the gated-attention step
CLAM = H.golden("clam")


def run_clam(i, device):
    import moc_amd.model_clam as Mc
    name, kind, kw, N, label, fkw = HB.CLAM_CASES[i]
    return name, HB.run_clam_case(Mc, kind, kw, N, label, fkw, 7000 + 13 * i, device=device)


def check_clam(name, got, atol):
    np.testing.assert_allclose(got.pop("psig"), CLAM[f"{name}:psig"], rtol=1e-6, atol=1e-7,
                               err_msg=f"{name}: same seed, different parameters")
    HB.check_case(got, CLAM, name, atol=atol)


@pytest.mark.parametrize("i", range(len(HB.CLAM_CASES)), ids=[c[0] for c in HB.CLAM_CASES])
def test_clam_models_match_reference_outputs(i):
    with BO.patched():
        name, got = run_clam(i, "cpu")
    check_clam(name, got, 2e-5)


# ---------------------------------------------------------------- rows f3 x f4: the CLAM trainer hooks
CLAM_HOOKS = H.golden("clam_hooks")


def check_clam_hooks(got, tag, tol=2e-5):
    np.testing.assert_allclose(got["trace"], CLAM_HOOKS[f"{tag}:trace"], atol=tol, rtol=0, err_msg=f"{tag}: validate_clam / early-stopping trace")
    np.testing.assert_allclose(got["summary"], CLAM_HOOKS[f"{tag}:summary"], atol=tol, rtol=0)
    np.testing.assert_array_equal(got["acc"], CLAM_HOOKS[f"{tag}:acc"])
    np.testing.assert_allclose(got["probs"], CLAM_HOOKS[f"{tag}:probs"], atol=tol, rtol=0)
    keep = np.setdiff1d(np.arange(got["psig"].shape[0]), got["noise_rows"])
    assert len(keep) >= got["psig"].shape[0] - 1
    np.testing.assert_allclose(got["psig"][keep], CLAM_HOOKS[f"{tag}:psig"][keep], rtol=2e-5, atol=2e-6, err_msg=f"{tag}: parameters after training")


@pytest.mark.parametrize("case", HB.CLAM_HOOK_CASES, ids=[c[0] for c in HB.CLAM_HOOK_CASES])
def test_clam_trainer_hooks_match_reference(case, tmp_path, monkeypatch):
    """train_loop_clam (bag loss + instance loss mixed by bag_weight), validate_clam (AUC-criterion early stopping) and
    summary, with CLAM_SB / CLAM_MB, against what the reference's own functions and classes produced."""
    import moc_amd.core_utils as core
    import moc_amd.model_clam as Mc
    monkeypatch.setattr(core, "_device", lambda: torch.device("cpu"))
    with BO.patched():
        got = HB.run_clam_hooks(core, Mc, *case, torch.device("cpu"), tmp_path)
    check_clam_hooks(got, case[0])


def test_core_utils_train_refuses_what_is_not_on_the_path(tmp_path):
    import moc_amd.core_utils as core
    args = types.SimpleNamespace(model_type="vila", n_classes=2, results_dir=str(tmp_path))
    with pytest.raises(AssertionError, match="not on this path"):
        core.train((None, None, None), 0, args)
