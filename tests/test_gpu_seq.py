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
PASSES = [None, None, 5, None, 10]      # repeat_num per pass (None = all 7 slides): a partial pass in the middle, a wrapping one (7 + 3) last


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
        m_last = len(res) % len(SIZES) or len(SIZES)
        losses.append(M.train.last[0].meta_ws()[0]["loss"][len(res) - m_last:len(res)].cpu().numpy().copy())
    return H.flat_params(model), losses, torch.get_rng_state(), H.flat_state(opt, "exp_avg_sq")


def _seq_run(dev, rank, world, C, j, K, store, group=None, hints=True):
    import helpers as H
    from moc_amd import main_moc as M, dist as mdist
    W, We, bags, labels = _task(C)
    M.set_classifier_bank(W.to(dev), We.to(dev))
    torch.manual_seed(5)
    model = M.senet(512, 4).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    blk = mdist.block_lists(len(SIZES), world)[rank]
    sh = mdist.SeqShardedBags([bags[i] for i in blk], SIZES, labels, dev, rank, world, dtype=store)
    torch.manual_seed(100)
    losses = []
    for i, rn in enumerate(PASSES):
        sh.repeat_num = rn
        if hints:       # tell the pass what follows (bench.py does; without it the speculation for a different length is dropped)
            nxt = PASSES[i + 1] if i + 1 < len(PASSES) else 0
            sh.next_pass_len = (nxt or len(SIZES)) if nxt != 0 else 0
        mdist.train_seq(model, sh, opt, dev, H.make_args(C, j, K), group=group)
        torch.cuda.synchronize()
        m_last = len(sh) % len(SIZES) or len(SIZES)               # (a wrapping pass runs as rounds: the last round's losses)
        losses.append(mdist.train_seq.last[0].meta_ws()[0]["loss"][:m_last].cpu().numpy().copy())
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


@pytest.mark.parametrize("world,C,j,K", [(2, 2, 100, 10), (3, 2, 100, 10), (2, 30, 40, 5), (5, 2, 100, 10)])
def test_ranks_sharing_one_device_are_bit_identical_to_one_gpu(gpu_device, world, C, j, K):
    """(world 5 over 7 slides: blocks of 2, 2, 1, 1, 1 -- and the 5-slide partial pass leaves ranks 3 and 4 nothing)"""
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


# ---------------------------------------------------------------- the whole driver, two ranks (rows f1 x e)
def _driver_worker(rank, world, port, q, root, cid):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch.distributed as dist
        import helpers as H
        from test_gpu_driver import _task_on_disk
        from moc_amd import datasets as DS, main_moc as M, run_moc, dist as mdist
        torch.set_num_threads(2)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda:0")
        g = H.golden("driver")
        _, ntr, nva, nte, C, j, K, rep, seed = [int(v) for v in g["cases"][cid]]
        W, We = synth_bank(seed, C)
        M.set_classifier_bank(W.to(dev), We.to(dev))
        names = {f"CLS{c}": c for c in range(C)}
        ds = DS.Generic_MIL_Dataset(csv_path=os.path.join(root, "dataset_csv", "t.csv"),
                                    data_dir=os.path.join(root, "data", "t", "merge_features_conch"), print_info=False, label_dict=names)
        tr, va, te = ds.return_splits(from_id=False, csv_path=os.path.join(root, "splits", "splits_0.csv"), repeat_num=rep)
        loaders = [DS.to_sharded(sp, dev, rank, world, train=(i == 0)) for i, sp in enumerate((tr, va, te))]
        assert isinstance(loaders[0], mdist.SeqShardedBags) and loaders[0].local is not None
        args = run_moc.get_args(["--topj", str(j), "--topk", str(K), "--shot", "4", "--fold", "0", "--disable_tqdm",
                                 "--result_dir", os.path.join(root, "res")])
        args.n_classes = C
        torch.manual_seed(seed)
        model = M.senet(512, 4).to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
        val_aucs, orig = [], mdist.evaluation

        def logged(model_, split, device, a, group=None):
            r = orig(model_, split, device, a, group)
            if split is loaders[1]:
                val_aucs.append(r["auc"])
            return r
        mdist.evaluation = logged
        torch.manual_seed(seed + 1)
        res = run_moc.main(args, model, opt, *loaders, dev)
        q.put((rank, (val_aucs, res, H.flat_params(model))))
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "ERR " + traceback.format_exc()))


def synth_bank(seed, C):
    from moc_amd import synth
    return synth.make_bank(seed, 512, C)


@pytest.mark.parametrize("cid", [0, 1])
def test_two_rank_driver_reproduces_the_reference_main_fixture(gpu_device, tmp_path, cid):
    """run_moc.main() with every split spread over two ranks (train: exact-sequential; evaluations: sharded + gathered)
    against what the reference's own main() produced on the CPU for the same task (tests/golden/driver.npz): per-epoch
    validation AUC within +-0.002, same best epoch / test AUC / accuracy, zero-shot results, result files written by
    rank 0.  Case 1 visits its 6 train slides 9 times per epoch (repeat_num beyond the split: a round of 6 and one of 3)."""
    import helpers as H
    import json
    from test_gpu_driver import _task_on_disk
    g = H.golden("driver")
    _task_on_disk(str(tmp_path), cid, g)
    world = 2
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_driver_worker, args=(r, world, port, q, str(tmp_path), cid)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for r, v in out.items():
        assert not isinstance(v, str), f"rank {r}: {v}"
    exp = g[f"c{cid}_result"]
    for r in range(world):
        val_aucs, res, params = out[r]
        np.testing.assert_allclose(val_aucs, g[f"c{cid}_val_auc"], atol=2e-3)
        assert abs(res["best_val"] - exp[0]) < 2e-3 and abs(res["test_at_best_val"] - exp[1]) < 2e-3
        assert abs(res["test_acc_at_best_val"] - exp[2]) < 1e-9 and res["best_epoch"] == int(exp[3])
        np.testing.assert_allclose(params, g[f"c{cid}_final_params"], atol=5e-4)
    np.testing.assert_array_equal(out[0][2], out[1][2])
    on_disk = json.load(open(tmp_path / "res" / "best_results_shot_4_fold_0.json"))
    assert on_disk["best_epoch"] == int(exp[3])
    zs = json.load(open(tmp_path / "res" / "zs_results_shot_4_fold_0.json"))
    got_zs = np.array([[zs[k]["loss"], zs[k]["acc"], zs[k]["auc"]] for k in ("zs_train", "zs_val", "zs_test")])
    np.testing.assert_allclose(got_zs, g[f"c{cid}_zs"], atol=1e-4)
