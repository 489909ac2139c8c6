"""Exact-sequential multi-GPU training (SURVEY.md section 8e mode 1, moc_amd.dist.train_seq): phase A sharded over the
ranks that hold the bags, compact results all-gathered, every rank runs the reference's one-step-per-slide recurrence.
The bar is not a tolerance: parameters, losses and the CPU generator must be BIT-IDENTICAL to the single-GPU
main_moc.train (which the parity tests pin to the reference), at world 1 in-process and at world 2 / 3 with the ranks
sharing cuda:0 (gloo; the box has one GPU -- the collective backend is not what is under test)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

SIZES = [900, 1100, 1000, 800, 950, 1050, 700]
PASSES = [None, None, 5, None]          # repeat_num per pass (None = all slides): a partial pass in the middle


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _task(C):
    from moc_amd import synth
    W, We = synth.make_bank(78, 512, C)
    bags, labels = synth.make_slide_set(7800, SIZES, 512, We, C)
    return W, We, bags, labels


def _single_gpu(dev, C, j, K, store):
    """The reference path: main_moc.train over a resident split, one GPU."""
    import helpers as H
    from moc_amd import main_moc as M
    W, We, bags, labels = _task(C)
    M.set_classifier_bank(W.to(dev), We.to(dev))
    torch.manual_seed(5)
    model = M.senet(512, 4).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    res = M.ResidentBags(bags, labels, dev, dtype=store)
    torch.manual_seed(100)
    losses = []
    for rn in PASSES:
        res.repeat_num = rn
        M.train(model, res, opt, dev, H.make_args(C, j, K))
        torch.cuda.synchronize()
        losses.append(M.train.last[0].meta_ws()[0]["loss"][:len(res)].cpu().numpy().copy())
    return H.flat_params(model), losses, torch.get_rng_state(), H.flat_state(opt, "exp_avg_sq")


def _seq_run(dev, rank, world, C, j, K, store, group=None, hints=True):
    import helpers as H
    from moc_amd import main_moc as M, dist as mdist
    W, We, bags, labels = _task(C)
    M.set_classifier_bank(W.to(dev), We.to(dev))
    torch.manual_seed(5)
    model = M.senet(512, 4).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    per = (len(SIZES) + world - 1) // world
    lo, hi = min(len(SIZES), rank * per), min(len(SIZES), (rank + 1) * per)
    sh = mdist.SeqShardedBags(bags[lo:hi], SIZES, labels, dev, rank, world, dtype=store)
    torch.manual_seed(100)
    losses = []
    for i, rn in enumerate(PASSES):
        sh.repeat_num = rn
        if hints:       # tell the pass what follows (bench.py does; without it the speculation for a different length is dropped)
            nxt = PASSES[i + 1] if i + 1 < len(PASSES) else 0
            sh.next_pass_len = (nxt or len(SIZES)) if nxt != 0 else 0
        mdist.train_seq(model, sh, opt, dev, H.make_args(C, j, K), group=group)
        torch.cuda.synchronize()
        losses.append(mdist.train_seq.last[0].meta_ws()[0]["loss"][:len(sh)].cpu().numpy().copy())
    return H.flat_params(model), losses, torch.get_rng_state(), H.flat_state(opt, "exp_avg_sq")


@pytest.mark.parametrize("C,j,K,store,hints", [(2, 100, 10, torch.bfloat16, True), (2, 100, 10, torch.float32, False),
                                               (30, 40, 5, torch.bfloat16, True)])
def test_world_1_is_bit_identical_to_train(gpu_device, C, j, K, store, hints):
    a = _single_gpu(gpu_device, C, j, K, store)
    b = _seq_run(gpu_device, 0, 1, C, j, K, store, hints=hints)
    np.testing.assert_array_equal(a[0], b[0])
    for x, y in zip(a[1], b[1]):
        np.testing.assert_array_equal(x, y)
    assert torch.equal(a[2], b[2]), "the CPU generator ends in a different state"
    np.testing.assert_array_equal(a[3], b[3])


def _worker(rank, world, port, q, C, j, K):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch.distributed as dist
        torch.set_num_threads(2)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda:0")
        out = _seq_run(dev, rank, world, C, j, K, torch.bfloat16)
        ref = _single_gpu(dev, C, j, K, torch.bfloat16) if rank == 0 else None
        q.put((rank, (out[0], out[1], out[2].numpy(), None if ref is None else (ref[0], ref[1], ref[2].numpy()))))
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "ERR " + traceback.format_exc()))


@pytest.mark.parametrize("world,C,j,K", [(2, 2, 100, 10), (3, 2, 100, 10), (2, 30, 40, 5)])
def test_ranks_sharing_one_device_are_bit_identical_to_one_gpu(gpu_device, world, C, j, K):
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, C, j, K)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for r, v in out.items():
        assert not isinstance(v, str), f"rank {r}: {v}"
    ref = out[0][3]
    for r in range(world):
        np.testing.assert_array_equal(out[r][0], ref[0], err_msg=f"rank {r}: parameters differ from the single-GPU run")
        for x, y in zip(out[r][1], ref[1]):
            np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(out[r][2], ref[2])
