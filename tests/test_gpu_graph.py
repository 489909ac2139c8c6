"""The pass of meta-steps as ONE hipGraph launch (moc_train_steps_graph) against the 2 n + 1 stream launches it
replaces (moc_train_steps): the same kernels with the same coefficient floats, so every parameter, Adam moment and
per-slide loss must be BIT-identical -- through table rebuilds, step counts changed behind the handle's back, odd pass
lengths (the W2 copy-back node), partial passes and the wide step."""
import numpy as np
import pytest
import torch

import helpers as H
from moc_amd import synth

pytestmark = pytest.mark.gpu


def _state(model, opt):
    ps = list(model.parameters())
    return ([p.detach().cpu().clone() for p in ps] + [opt.state[p]["exp_avg"].detach().cpu().clone() for p in ps] +
            [opt.state[p]["exp_avg_sq"].detach().cpu().clone() for p in ps] + [torch.tensor([float(opt.state[p]["step"]) for p in ps])])


def _run(dev, graph, C, sizes, dtype, lengths, j=100, K=10, table_steps=None, poke=None, seed=5):
    from moc_amd import engine as E, main_moc as M
    keep = (E.STEP_GRAPH, E.GRAPH_TABLE_STEPS)
    E.STEP_GRAPH = graph
    if table_steps:
        E.GRAPH_TABLE_STEPS = table_steps
    try:
        W, We = synth.make_bank(seed, 512, C)
        bags, labels = synth.make_slide_set(100 * seed, sizes, 512, We, C)
        M.set_classifier_bank(W.to(dev), We.to(dev))
        res = M.ResidentBags(bags, labels, dev, dtype=dtype)
        torch.manual_seed(seed)
        model = M.senet(512, 4).to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
        args = H.make_args(C, j, K)
        torch.manual_seed(seed + 1)
        losses = []
        for i, m in enumerate(lengths):
            res.repeat_num = m if m != len(sizes) else None
            res.next_pass_len = lengths[i + 1] if i + 1 < len(lengths) else 0
            if poke and i in poke:
                poke[i](model, opt)
            M.train(model, res, opt, dev, args)
            losses.append(M.train.last[0].meta_ws()[0]["loss"][:m].cpu().clone())
        torch.cuda.synchronize()
        meta = E.MetaState.cached(model, opt)
        return _state(model, opt), losses, meta.graph_stats()
    finally:
        E.STEP_GRAPH, E.GRAPH_TABLE_STEPS = keep


def _same(a, b):
    (sa, la, _), (sb, lb, _) = a, b
    for x, y in zip(sa, sb):
        assert torch.equal(x, y), "graph replay and stream launches ended with different bits"
    for x, y in zip(la, lb):
        assert torch.equal(x, y)


@pytest.mark.parametrize("C,dtype,sizes", [(2, torch.float32, [900, 1300, 700, 1100, 1000]),
                                           (2, torch.bfloat16, [900, 1300, 700, 1100]),
                                           (3, torch.float16, [800, 1200, 600]),
                                           (30, torch.bfloat16, [1200, 900, 1100]),      # wide step, pooling inside
                                           (30, torch.bfloat16, [4000, 3600])])          # wide step behind topk_mean_kernel
def test_graph_pass_is_bit_identical_to_stream_launches(gpu_device, C, dtype, sizes):
    n = len(sizes)
    lengths = [n, n, n - 1, n, 1, n]            # whole passes, partial passes (odd and even lengths), a one-step pass
    g = _run(gpu_device, True, C, sizes, dtype, lengths)
    e = _run(gpu_device, False, C, sizes, dtype, lengths)
    _same(g, e)
    captures, replays, eager = g[2]
    assert replays == len(lengths) and eager == 0, g[2]
    assert captures <= 2 * 3, g[2]               # one graph per (pass length, work-array set): three lengths, two sets
    assert e[2] == (0, 0, 0)


def test_graph_survives_table_rebuilds_and_foreign_steps(gpu_device):
    """A 64-step coefficient table (rebuilt every few passes), a learning-rate change between passes, and an
    optimizer whose step count was moved by someone else (a loaded state dict): the graph path must follow all of
    them exactly as the stream launches do."""
    sizes = [700, 900, 800, 1000, 600, 750, 850]
    n = len(sizes)

    def new_lr(model, opt):
        for gr in opt.param_groups:
            gr["lr"] = 3e-4

    def reload_state(model, opt):
        sd = opt.state_dict()
        for st in sd["state"].values():
            st["step"] = st["step"] + 5           # as if five more steps had been taken elsewhere
        opt.load_state_dict(sd)

    def bump_in_place(model, opt):
        for p in model.parameters():
            opt.state[p]["step"] += 3             # same tensors, same handle: only the count moved

    lengths = [n] * 24                            # 168 steps through a 64-entry table
    poke = {7: new_lr, 11: bump_in_place, 15: reload_state}
    g = _run(gpu_device, True, 2, sizes, torch.float32, lengths, table_steps=64, poke=poke)
    e = _run(gpu_device, False, 2, sizes, torch.float32, lengths, table_steps=64, poke=poke)
    _same(g, e)
    # (the reloaded state dict brought new moment tensors: a new MetaState and handle from pass 15 on)
    assert g[2][1] == len(lengths) - 15 and g[2][2] == 0, g[2]


def test_graph_matches_oracle_losses(gpu_device):
    """The graph path against the CPU oracle directly (not only against the stream launches)."""
    from moc_amd import engine as E, main_moc as M
    from oracle import moc_oracle as O
    dev = gpu_device
    keep, E.STEP_GRAPH = E.STEP_GRAPH, True             # (opt-in: stream launches are the default, MOC_STEP_GRAPH=1 switches)
    try:
        _graph_vs_oracle(dev, E, M, O)
    finally:
        E.STEP_GRAPH = keep


def _graph_vs_oracle(dev, E, M, O):
    C, j, K = 2, 100, 10
    W, We = synth.make_bank(9, 512, C)
    bags, labels = synth.make_slide_set(900, [1000, 1200, 900, 1100], 512, We, C)
    torch.manual_seed(3)
    ref_model = O.Senet(512, 4)
    ref_opt = O.make_optimizer(ref_model)
    torch.manual_seed(3)
    model = M.senet(512, 4).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    M.set_classifier_bank(W.to(dev), We.to(dev))
    res = M.ResidentBags(bags, labels, dev)
    args = H.make_args(C, j, K)
    for epoch in range(3):
        torch.manual_seed(50 + epoch)
        ref_losses = O.train_epoch(ref_model, ref_opt, bags, labels, W, We, C, j, K)
        torch.manual_seed(50 + epoch)
        res.next_pass_len = 0
        M.train(model, res, opt, dev, args)
        np.testing.assert_allclose(M.train.last[0].meta_ws()[0]["loss"].cpu().numpy(), np.asarray(ref_losses), atol=H.ATOL)
    assert E.MetaState.cached(model, opt).graph_stats()[1] == 3
    H.assert_adam_params_close(H.flat_params(model), H.flat_params(ref_model), H.flat_state(ref_opt, "exp_avg_sq"),
                               step=12, grad_noise=1e-6, what="graph path, 3 epochs")
