"""The data-parallel trainer end to end on the GPU: 2 ranks (gloo, both on cuda:0 -- the box has one
GPU; the collective backend is not what is under test) against the oracle doing synchronous
minibatch steps of 2 slides -- once with the gradient exchange inside the step kernel (peer-mapped
receive buffers, moc_p2p_*), once with the all-reduce between the launches.  Also the sharded
evaluation gather."""
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


SIZES = [900, 1100, 1000, 800, 950, 1050, 700, 1200, 850, 1150, 990, 760]      # world x T slides: step t uses slides [t*world, (t+1)*world)
# rank 0's bags all <= 4096 rows, rank 1's all above, with topj (2C + 2) = 7200 > 4096: taken per rank, the step-kernel
# choice would differ (narrow on rank 0, wide on rank 1) and the in-kernel exchange would pair up the wrong workgroups
SIZES_SPLIT = [3000, 5200, 3500, 4500, 2800, 6000]
SHAPES = {"std": (SIZES, 100, 10), "split": (SIZES_SPLIT, 400, 5)}


def _steps(world, sizes=SIZES):
    return len(sizes) // world


def _worker(rank, world, port, q, exchange, C, shape="std"):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          MOC_DP_EXCHANGE=exchange, MOC_P2P_TIMEOUT_MS="20000")
        import torch.distributed as dist
        import helpers as H
        from moc_amd import main_moc as M, synth, dist as mdist
        torch.set_num_threads(2)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda:0")
        sizes, j, K = SHAPES[shape]
        W, We = synth.make_bank(77, 512, C)
        bags, labels = synth.make_slide_set(7700, sizes, 512, We, C)
        T = _steps(world, sizes)
        mine = [rank + world * t for t in range(T)]               # step t uses slide t of every rank
        torch.manual_seed(5)
        model = M.senet(512, 4).to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
        M.set_classifier_bank(W.to(dev), We.to(dev))
        res = M.ResidentBags([bags[i] for i in mine], [labels[i] for i in mine], dev)
        torch.manual_seed(100 + rank)                             # each rank draws its own masks
        mdist.train_dp(model, res, opt, dev, H.make_args(C, j, K))
        torch.cuda.synchronize()
        assert mdist.exchange_error() == 0
        losses = mdist.train_dp.last[0].meta_ws()[0]["loss"].cpu().numpy()
        ev = None
        if C <= len(sizes):          # the AUC needs every class among the slides (sklearn's rule, the reference's too)
            ev = mdist.evaluation_dp(model, res, dev, H.make_args(C, j, K), labels, mine,
                                     [[r + world * t for t in range(T)] for r in range(world)])
        q.put((rank, (H.flat_params(model), losses, ev, int(float(opt.state[next(model.parameters())]["step"])),
                      mdist.train_dp.exchange)))
        mdist.shutdown()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "ERR " + traceback.format_exc()))


@pytest.mark.parametrize("exchange,world,C,shape", [("auto", 2, 2, "std"), ("rccl", 2, 2, "std"), ("auto", 4, 2, "std"),
                                                     ("auto", 2, 30, "std"), ("rccl", 2, 30, "std"), ("auto", 2, 8, "split")])
def test_train_dp_matches_minibatch_oracle(gpu_device, exchange, world, C, shape):
    import helpers as H
    from moc_amd import synth
    from oracle import moc_oracle as O
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    sizes, j, K = SHAPES[shape]
    T = _steps(world, sizes)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, exchange, C, shape)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for r, v in out.items():
        assert not isinstance(v, str), f"rank {r}: {v}"
    for r in range(1, world):
        assert np.array_equal(out[0][0], out[r][0]), "ranks hold different parameters after the summed-gradient steps"
    for r in range(world):
        assert out[r][3] == T
        # the path asked for is the path taken (auto = the in-kernel exchange on one node)
        assert out[r][4] == ("p2p" if exchange == "auto" else "collective")
    # oracle: T synchronous steps, each the mean gradient of one slide per rank
    W, We = synth.make_bank(77, 512, C)
    bags, labels = synth.make_slide_set(7700, sizes, 512, We, C)
    masks = {}
    for rank in range(world):
        torch.manual_seed(100 + rank)
        for i in [rank + world * t for t in range(T)]:
            masks[i] = O.draw_mask(sizes[i])
    torch.manual_seed(5)
    ref = O.Senet(512, 4)
    ropt = O.make_optimizer(ref)
    ref_losses = {}
    for t in range(T):
        grads = []
        for rank in range(world):
            i = rank + world * t
            sr = O.slide_process(bags[i], W, We, C, j, mask=masks[i])
            pooled = O.pool_top(O.mix_train(ref(sr["selected_feat"]), sr), [K])[1][K]
            loss = torch.nn.functional.cross_entropy(pooled, torch.tensor([labels[i]]))
            ref_losses[i] = float(loss.detach())
            grads.append(torch.autograd.grad(loss, list(ref.parameters())))
        for p, *gs in zip(ref.parameters(), *grads):
            p.grad = sum(gs) / world
        ropt.step()
    for rank in range(world):
        np.testing.assert_allclose(out[rank][1], [ref_losses[rank + world * t] for t in range(T)], atol=1e-4)
    H.assert_adam_params_close(out[0][0], H.flat_params(ref), H.flat_state(ropt, "exp_avg_sq"), step=T,
                               grad_noise=1e-6, what=f"dp{world}")
    if C > len(sizes):
        return
    ev_ref = O.evaluation(ref, bags, labels, W, We, C, j, K)
    for rank in range(world):
        ev = out[rank][2]
        assert abs(ev["loss"] - ev_ref["loss"]) < 1e-4 and ev["acc"] == ev_ref["acc"] and abs(ev["auc"] - ev_ref["auc"]) < 2e-3
