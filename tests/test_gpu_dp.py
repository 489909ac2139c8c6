"""The data-parallel trainer end to end on the GPU: 2 ranks (gloo, both on cuda:0 -- the box has one
GPU; the collective backend is not what is under test) against the oracle doing synchronous
minibatch steps of 2 slides.  Also the sharded evaluation gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch.distributed as dist
        import helpers as H
        from moc_amd import main_moc as M, synth, dist as mdist
        torch.set_num_threads(2)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda:0")
        C, j, K = 2, 100, 10
        W, We = synth.make_bank(77, 512, C)
        sizes = [900, 1100, 1000, 800]
        bags, labels = synth.make_slide_set(7700, sizes, 512, We, C)
        mine = [rank, rank + 2]                                   # step t uses slide t of every rank
        torch.manual_seed(5)
        model = M.senet(512, 4).to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
        M.set_classifier_bank(W.to(dev), We.to(dev))
        res = M.ResidentBags([bags[i] for i in mine], [labels[i] for i in mine], dev)
        torch.manual_seed(100 + rank)                             # each rank draws its own masks
        mdist.train_dp(model, res, opt, dev, H.make_args(C, j, K))
        torch.cuda.synchronize()
        losses = mdist.train_dp.last[0].meta_ws()[0]["loss"].cpu().numpy()
        ev = mdist.evaluation_dp(model, res, dev, H.make_args(C, j, K), labels, mine, [[0, 2], [1, 3]])
        q.put((rank, (H.flat_params(model), losses, ev, int(float(opt.state[next(model.parameters())]["step"])))))
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "ERR " + traceback.format_exc()))


def test_train_dp_two_ranks_matches_batch2_oracle(gpu_device):
    import helpers as H
    from moc_amd import synth
    from oracle import moc_oracle as O
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    for r, v in out.items():
        assert not isinstance(v, str), f"rank {r}: {v}"
    assert np.array_equal(out[0][0], out[1][0]), "ranks hold different parameters after the all-reduced steps"
    assert out[0][3] == 2 and out[1][3] == 2
    # oracle: two synchronous steps, each the mean gradient of one slide per rank
    C, j, K = 2, 100, 10
    W, We = synth.make_bank(77, 512, C)
    sizes = [900, 1100, 1000, 800]
    bags, labels = synth.make_slide_set(7700, sizes, 512, We, C)
    masks = {}
    for rank in range(2):
        torch.manual_seed(100 + rank)
        for i in (rank, rank + 2):
            masks[i] = O.draw_mask(sizes[i])
    torch.manual_seed(5)
    ref = O.Senet(512, 4)
    ropt = O.make_optimizer(ref)
    ref_losses = {}
    for t in range(2):
        grads = []
        for rank in range(2):
            i = rank + 2 * t
            sr = O.slide_process(bags[i], W, We, C, j, mask=masks[i])
            pooled = O.pool_top(O.mix_train(ref(sr["selected_feat"]), sr), [K])[1][K]
            loss = torch.nn.functional.cross_entropy(pooled, torch.tensor([labels[i]]))
            ref_losses[i] = float(loss)
            grads.append(torch.autograd.grad(loss, list(ref.parameters())))
        for p, g0, g1 in zip(ref.parameters(), *grads):
            p.grad = (g0 + g1) / 2
        ropt.step()
    for rank in range(2):
        np.testing.assert_allclose(out[rank][1], [ref_losses[rank], ref_losses[rank + 2]], atol=1e-4)
    H.assert_adam_params_close(out[0][0], H.flat_params(ref), H.flat_state(ropt, "exp_avg_sq"), step=2,
                               grad_noise=1e-6, what="dp2")
    ev_ref = O.evaluation(ref, bags, labels, W, We, C, j, K)
    for rank in range(2):
        ev = out[rank][2]
        assert abs(ev["loss"] - ev_ref["loss"]) < 1e-4 and ev["acc"] == ev_ref["acc"] and abs(ev["auc"] - ev_ref["auc"]) < 2e-3
