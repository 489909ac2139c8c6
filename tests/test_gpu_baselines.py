"""SURVEY.md section 8 row f3 on the GPU: the same cases as tests/test_baselines_cpu.py, with the top-j
pooling / top-instance selection on the HIP path (moc_topk_mean through moc_amd.pool_autograd), forward
AND backward, against the fixtures the reference's own classes produced."""
import contextlib
import io

import numpy as np
import pytest
import torch

import helpers as H
import helpers_baselines as HB
from test_baselines_cpu import GOLD, _ns, check_hooks, run_hooks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("i", range(len(HB.BASELINE_CASES)), ids=[c[0] for c in HB.BASELINE_CASES])
def test_baseline_models_on_the_hip_path(gpu_device, i):
    name, kind, kw, N, label = HB.BASELINE_CASES[i]
    seed = 4000 + 17 * i
    dev = torch.device("cuda:0")
    with contextlib.redirect_stdout(io.StringIO()):
        cls, kwargs = HB.build_case(_ns(kind), kind, kw, seed, device=dev)
        model = cls(**kwargs).to(dev)
        # cached-sample initialisation standardises the samples on the GPU here (mean / std reductions in a
        # different order than on the CPU): compare relative to each tensor's absolute sum
        sig, exp = HB.psig(model.cpu()), GOLD[f"{name}:psig"]
        assert np.all(np.abs(sig - exp) <= 5e-6 * exp[:, 1:2] + 1e-7), (name, sig, exp)
        got = HB.run_case(model.to(dev), kind, N, label, seed, device=dev)
    HB.check_case(got, GOLD, name, atol=5e-5)


@pytest.mark.parametrize("case", HB.HOOK_CASES, ids=[c[0] for c in HB.HOOK_CASES])
def test_trainer_hooks_on_the_hip_path(gpu_device, case, tmp_path):
    import moc_amd.core_utils as core
    import moc_amd.model_mil as Mm
    got = run_hooks(core, Mm, *case, torch.device("cuda:0"), tmp_path)
    check_hooks(got, case[0], tol=1e-4)


def test_topk_mean_pool_gradient_is_the_gather_gradient(gpu_device):
    from moc_amd.pool_autograd import topk_mean_pool
    dev = torch.device("cuda:0")
    x = HB.randn(5, 500, 7).to(dev).requires_grad_(True)
    w = torch.arange(1, 8, device=dev, dtype=torch.float32)
    (topk_mean_pool(x, 13) * w).sum().backward()
    ref = x.detach().cpu().clone().requires_grad_(True)
    (ref.topk(13, 0)[0].mean(0, keepdim=True) * w.cpu()).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), ref.grad.numpy(), atol=1e-7)


# ---------------------------------------------------------------- row f4: gated-attention pooling kernel
@pytest.mark.parametrize("N,L,D,K", [(1, 512, 256, 1), (63, 512, 256, 1), (64, 512, 384, 3), (1000, 512, 384, 1),
                                     (4097, 1024, 128, 2), (15000, 512, 384, 1), (777, 64, 256, 5),
                                     (300, 512, 256, 64), (65, 48, 128, 33), (1, 16, 128, 1)])      # K up to 64; L % 32 == 16; one row
def test_gated_attention_pool_matches_restatement(gpu_device, N, L, D, K):
    from moc_amd import engine
    from oracle import baselines_oracle as BO
    g = lambda s, *shape: HB.randn(s, *shape)
    h = torch.relu(g(1, N, L))                                    # CLAM feeds ReLU features
    Wa, Wb = g(2, D, L) * (2.0 / (L + D)) ** 0.5, g(3, D, L) * (2.0 / (L + D)) ** 0.5
    ba, bb = g(4, D) * 0.1, g(5, D) * 0.1
    Wc, bc = g(6, K, D) * (2.0 / (D + K)) ** 0.5 * 3.0, g(7, K) * 0.1
    A_ref, M_ref = BO.gated_attention_pool(h.double(), Wa.double(), ba.double(), Wb.double(), bb.double(), Wc.double(), bc.double())
    dev = torch.device("cuda:0")
    A, M = engine.gated_attention_pool(*[t.to(dev) for t in (h, Wa, ba, Wb, bb, Wc, bc)])
    np.testing.assert_allclose(A.cpu().numpy(), A_ref.float().numpy(), atol=1e-4, rtol=0)
    np.testing.assert_allclose(M.cpu().numpy(), M_ref.float().numpy(), atol=1e-4, rtol=0)


def test_gated_attention_pool_survives_large_scores(gpu_device):
    """Online softmax: scores of +-80 must neither overflow nor lose the dominant rows."""
    from moc_amd import engine
    from oracle import baselines_oracle as BO
    N, L, D = 500, 512, 256
    h = torch.relu(HB.randn(11, N, L))
    Wa, Wb = HB.randn(12, D, L) * 0.05, HB.randn(13, D, L) * 0.05
    ba, bb, bc = torch.zeros(D), torch.zeros(D), torch.zeros(1)
    Wc = HB.randn(14, 1, D) * 20.0
    A_ref, M_ref = BO.gated_attention_pool(h.double(), Wa.double(), ba.double(), Wb.double(), bb.double(), Wc.double(), bc.double())
    assert float(A_ref.abs().max()) > 60
    dev = torch.device("cuda:0")
    A, M = engine.gated_attention_pool(*[t.to(dev) for t in (h, Wa, ba, Wb, bb, Wc, bc)])
    assert torch.isfinite(M).all()
    np.testing.assert_allclose(A.cpu().numpy(), A_ref.float().numpy(), atol=2e-3, rtol=0)
    np.testing.assert_allclose(M.cpu().numpy(), M_ref.float().numpy(), atol=5e-3, rtol=0)


@pytest.mark.parametrize("N,L,D,K,use", [(1, 512, 256, 1, "AM"), (63, 512, 256, 1, "AM"), (64, 512, 384, 3, "AM"),
                                         (1000, 512, 384, 1, "M"), (1000, 512, 384, 2, "A"), (4097, 1024, 128, 2, "AM"),
                                         (15000, 512, 384, 1, "AM"), (777, 64, 256, 5, "AM"), (65, 48, 128, 33, "AM"),
                                         (300, 512, 256, 64, "AM")])
def test_gated_attention_backward_matches_float64_autograd(gpu_device, N, L, D, K, use):
    """moc_gated_attention_backward (through the autograd Function the models use) against autograd on the float64
    restatement: every input's gradient, with the gradient arriving at A_raw, at M, or both.  Tolerance 1e-4 of each
    gradient's largest magnitude (fp32 sums over up to 15,000 rows; floor 1e-2: d_bc is exactly zero when only M is
    used -- the softmax does not see a shift -- and arrives as the rounding of a cancelling sum)."""
    from moc_amd.model_clam import gated_attention_pool
    from oracle import baselines_oracle as BO
    g = lambda s, *shape: HB.randn(s, *shape)
    h = torch.relu(g(1, N, L))
    ops = [h, g(2, D, L) * (2.0 / (L + D)) ** 0.5, g(4, D) * 0.1, g(3, D, L) * (2.0 / (L + D)) ** 0.5, g(5, D) * 0.1,
           g(6, K, D) * (2.0 / (D + K)) ** 0.5 * 3.0, g(7, K) * 0.1]
    uA, uM = g(8, K, N), g(9, K, L)
    ref_in = [t.double().requires_grad_(True) for t in ops]
    A_ref, M_ref = BO.gated_attention_pool(*ref_in)
    loss = (A_ref * uA.double()).sum() * ("A" in use) + (M_ref * uM.double()).sum() * ("M" in use)
    ref = torch.autograd.grad(loss, ref_in)
    dev = torch.device("cuda:0")
    got_in = [t.to(dev).requires_grad_(True) for t in ops]
    A, M = gated_attention_pool(*got_in)
    loss = (A * uA.to(dev)).sum() * ("A" in use) + (M * uM.to(dev)).sum() * ("M" in use)
    got = torch.autograd.grad(loss, got_in)
    for name, a, b in zip(["h", "Wa", "ba", "Wb", "bb", "Wc", "bc"], got, ref):
        scale = max(float(b.abs().max()), 1e-2)
        err = float((a.double().cpu() - b).abs().max()) / scale
        assert err < 1e-4, f"d{name}: {err:.3e} of its scale {scale:.3e}"


def test_gated_attention_backward_is_deterministic_and_skips_unneeded(gpu_device):
    """Fixed-order sums: two runs give the same bits; inputs that need no gradient get None (the un-gated network's
    constant gate operands)."""
    from moc_amd.model_clam import gated_attention_pool
    dev = torch.device("cuda:0")
    N, L, D, K = 5000, 512, 256, 2
    ops = [torch.relu(HB.randn(1, N, L)), HB.randn(2, D, L) * 0.05, HB.randn(3, D) * 0.1, torch.zeros(D, L), torch.full((D,), 40.0),
           HB.randn(4, K, D) * 0.2, HB.randn(5, K) * 0.1]
    runs = []
    for _ in range(2):
        ins = [t.to(dev).requires_grad_(i not in (3, 4)) for i, t in enumerate(ops)]
        A, M = gated_attention_pool(*ins)
        (A.sum() + (M * M).sum()).backward()
        assert ins[3].grad is None and ins[4].grad is None
        runs.append([t.grad.clone() for i, t in enumerate(ins) if i not in (3, 4)])
    for a, b in zip(*runs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("i", range(len(HB.CLAM_CASES)), ids=[c[0] for c in HB.CLAM_CASES])
def test_clam_models_on_the_hip_path(gpu_device, i):
    from test_baselines_cpu import check_clam, run_clam
    name, got = run_clam(i, torch.device("cuda:0"))
    if name in HB.CLAM_CPU_ONLY:
        # training-mode dropout: the masks are the GPU generator's, not the fixture's -- the forward must run (this is
        # the one mode that forms the scores with torch operations), give the right shapes and finite numbers
        exp = H.golden("clam")
        assert got["logits"].shape == exp[f"{name}:logits"].shape and got["A_raw"].shape == exp[f"{name}:A_raw"].shape
        assert all(np.isfinite(np.asarray(v, dtype=np.float64)).all() for k, v in got.items() if k != "psig")
        return
    check_clam(name, got, 1e-4)


@pytest.mark.parametrize("case", HB.CLAM_HOOK_CASES, ids=[c[0] for c in HB.CLAM_HOOK_CASES])
def test_clam_trainer_hooks_on_the_hip_path(gpu_device, case, tmp_path):
    """train_loop_clam / validate_clam / summary driving CLAM_SB / CLAM_MB on the GPU (attention through
    moc_gated_attention_pool / moc_gated_attention_backward) against the reference's own run."""
    import moc_amd.core_utils as core
    import moc_amd.model_clam as Mc
    from test_baselines_cpu import check_clam_hooks
    got = HB.run_clam_hooks(core, Mc, *case, torch.device("cuda:0"), tmp_path)
    check_clam_hooks(got, case[0], tol=2e-4)


def test_core_utils_train_drives_clam_end_to_end(gpu_device, tmp_path):
    """core_utils.train for model_type clam_sb: model construction, optimizer, cosine schedule, the CLAM loops, early
    stopping bookkeeping, checkpoint, summary -- one short fold on synthetic bags."""
    import types
    import pandas as pd
    import moc_amd.core_utils as core
    C, d = 2, 384
    loaders = []
    for s0 in (8100, 8200, 8300):
        ld = HB.Loader([(b.to("cuda:0"), y) for b, y in HB.hook_bags(s0, 6, d, C)])
        ld.dataset = types.SimpleNamespace(slide_data=pd.DataFrame({"slide_id": [f"s{k}" for k in range(len(ld))]}))
        loaders.append(ld)
    args = types.SimpleNamespace(model_type="clam_sb", n_classes=C, model_size="benchmark", drop_out=False, B=4, subtyping=False,
                                 bag_weight=0.7, opt="adam", lr=2e-4, reg=1e-5, max_epochs=3, early_stopping=True,
                                 results_dir=str(tmp_path), no_inst_cluster=False)
    torch.manual_seed(5)
    res, test_auc, val_auc, test_acc, val_acc = core.train(loaders, 0, args)
    assert (tmp_path / "s_0_checkpoint.pt").exists() and len(res) == 6
    assert 0.0 <= test_auc <= 1.0 and 0.0 <= val_auc <= 1.0 and 0.0 <= test_acc <= 1.0


def test_core_utils_train_drives_transmil(gpu_device, tmp_path):
    """model_type 'transmil' through core_utils.train (train_loop / validate / summary) on the GPU: the class keeps
    the reference's contract; its Nystrom attention is the restatement of moc_amd/nystrom.py (parity unpinned)."""
    import types
    import pandas as pd
    import moc_amd.core_utils as core
    C, d = 2, 512
    loaders = []
    for s0 in (8400, 8500, 8600):
        ld = HB.Loader([(b.to("cuda:0"), y) for b, y in HB.hook_bags(s0, 5, d, C)])
        ld.dataset = types.SimpleNamespace(slide_data=pd.DataFrame({"slide_id": [f"s{k}" for k in range(len(ld))]}))
        loaders.append(ld)
    args = types.SimpleNamespace(model_type="transmil", n_classes=C, model_size="conch", drop_out=False, opt="adam", lr=2e-4,
                                 reg=1e-5, max_epochs=2, early_stopping=False, results_dir=str(tmp_path))
    torch.manual_seed(6)
    res, test_auc, val_auc, test_acc, val_acc = core.train(loaders, 0, args)
    assert (tmp_path / "s_0_checkpoint.pt").exists() and len(res) == 5 and 0.0 <= test_auc <= 1.0
    state = torch.load(tmp_path / "s_0_checkpoint.pt", map_location="cpu")
    assert "layer1.attn.to_qkv.weight" in state and "pos_layer.proj.weight" in state
