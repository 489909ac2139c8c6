"""world_size-2 gloo tests of the multi-GPU host logic (moc_amd/dist.py): slide
sharding, the single flat-buffer meta-gradient all-reduce, and the ragged gather
used by sharded evaluation.  Gradients here come from the oracle (CPU) -- the
collective plumbing is what is under test; the kernels have their own GPU tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from moc_amd import dist as mdist
from moc_amd import synth
from oracle import moc_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(fn, world, *args):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_entry, args=(fn, r, world, port, q, args)) for r in range(world)]
    for p in procs:
        p.start()
    out = {}
    for _ in range(world):
        r, val = q.get(timeout=180)
        out[r] = val
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r, v in out.items():
        if isinstance(v, str) and v.startswith("ERR"):
            raise AssertionError(f"rank {r}: {v}")
    return out


def _entry(fn, rank, world, port, q, args):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        q.put((rank, fn(rank, world, *args)))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "ERR " + traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


# ------------------------------------------------------------------ pure functions
def test_shard_indices_partition_and_balance():
    sizes = [15000, 2000, 60000, 9000, 9000, 31000, 4000, 12000, 800]
    for world in (1, 2, 3, 8):
        parts = [mdist.shard_indices(len(sizes), r, world, sizes) for r in range(world)]
        assert sorted(i for p in parts for i in p) == list(range(len(sizes)))
        loads = [sum(sizes[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(sizes)          # greedy longest-first bound
        rr = [mdist.shard_indices(10, r, world) for r in range(world)]
        assert sorted(i for p in rr for i in p) == list(range(10))


def test_flat_grads_views_alias_one_buffer():
    params = [torch.zeros(64, 512), torch.zeros(64), torch.zeros(4, 64), torch.zeros(4)]
    fg = mdist.FlatGrads(params, "cpu")
    assert fg.flat.numel() == 33092
    fg.views[2].fill_(3.0)
    assert float(fg.flat[64 * 512 + 64]) == 3.0 and float(fg.flat.sum()) == 3.0 * 256
    assert mdist.allreduce_mean_(fg.flat) == 1.0             # no process group: identity, scale 1


def test_unshard_restores_item_order():
    lists = [[0, 3, 4], [1, 2]]
    rows = torch.tensor([[0.], [3.], [4.], [1.], [2.]])
    assert mdist.unshard(rows, lists, 5).flatten().tolist() == [0., 1., 2., 3., 4.]


# ------------------------------------------------------------------ world_size 2
def _dp_step_worker(rank, world, seed):
    """Each rank differentiates ITS slide; after the fused all-reduce every rank holds the mean
    gradient and takes the same Adam step == one process doing a batch of `world` slides."""
    C, j, K = 2, 100, 10
    W, We = synth.make_bank(seed, 512, C)
    bags, labels = synth.make_slide_set(seed + 1, [700, 900], 512, We, C)
    masks = []
    torch.manual_seed(seed)
    for b in bags:
        masks.append(O.draw_mask(b.size(0)))
    torch.manual_seed(seed + 7)
    model = O.Senet(512, 4)
    opt = O.make_optimizer(model)
    params = list(model.parameters())
    fg = mdist.FlatGrads(params, "cpu")

    def grad_of(m, i):
        sr = O.slide_process(bags[i], W, We, C, j, mask=masks[i])
        pooled = O.pool_top(O.mix_train(m(sr["selected_feat"]), sr), [K])[1][K]
        loss = torch.nn.functional.cross_entropy(pooled, torch.tensor([labels[i]]))
        return torch.autograd.grad(loss, list(m.parameters()))

    for v, g in zip(fg.views, grad_of(model, rank)):
        v.copy_(g)
    scale = mdist.allreduce_mean_(fg.flat)
    assert scale == 1.0 / world
    for p, v in zip(params, fg.views):
        p.grad = v * scale
    opt.step()
    got = torch.cat([p.detach().reshape(-1) for p in params])
    # single-process reference: mean of the two slides' gradients, same Adam
    torch.manual_seed(seed + 7)
    ref = O.Senet(512, 4)
    ropt = O.make_optimizer(ref)
    g0, g1 = grad_of(ref, 0), grad_of(ref, 1)
    for p, a, b in zip(ref.parameters(), g0, g1):
        p.grad = (a + b) / 2
    ropt.step()
    exp = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
    return float((got - exp).abs().max()), got.numpy()


def test_dp_meta_gradient_allreduce_world2():
    out = _run(_dp_step_worker, 2, 321)
    assert out[0][0] < 1e-6 and out[1][0] < 1e-6
    assert np.array_equal(out[0][1], out[1][1]), "ranks diverged after the all-reduced step"


def _gather_worker(rank, world):
    n_items, C = 7, 3
    sizes = [5000, 100, 3000, 2500, 2500, 400, 9000]
    lists = [mdist.shard_indices(n_items, r, world, sizes) for r in range(world)]
    mine = lists[rank]
    local = torch.tensor([[10.0 * i + c for c in range(C)] + [float(i)] for i in mine])     # [n_local, C+1]
    allv = mdist.unshard(mdist.gather_rows(local, [len(l) for l in lists]), lists, n_items)
    return allv.numpy()


def test_sharded_eval_gather_world2():
    out = _run(_gather_worker, 2)
    exp = np.array([[10.0 * i + c for c in range(3)] + [float(i)] for i in range(7)], dtype=np.float32)
    assert np.array_equal(out[0], exp) and np.array_equal(out[1], exp)


# ------------------------------------------------------------------ splits spread over the ranks (round 2)
def test_block_lists_are_contiguous_and_partition():
    for n, w in [(7, 3), (32, 8), (5, 8), (48, 4), (1, 2), (12, 8), (2, 2), (4, 3)]:
        blocks = mdist.block_lists(n, w)
        assert len(blocks) == w and sum(blocks, []) == list(range(n))
        per = (n + w - 1) // w
        assert all(len(b) <= per for b in blocks)
        # balanced: sizes differ by at most one, the longer blocks first -- so no rank is empty unless w > n
        lens = [len(b) for b in blocks]
        assert max(lens) - min(lens) <= 1 and lens == sorted(lens, reverse=True)
        assert (min(lens) >= 1) == (w <= n)
    assert [len(b) for b in mdist.block_lists(12, 8)] == [2, 2, 2, 2, 1, 1, 1, 1]      # (ceil-sized blocks: 2 x 6, then two empty ranks)
    with pytest.raises(AssertionError, match="partition"):
        mdist.ShardedSplit([], [], [0, 1, 0], [[0, 1], [1, 2]], "cpu")


def _sharded_driver_inputs(rank, world, root):
    """Every rank reads only its block of each split (datasets.to_sharded); the train split's row counts are
    exchanged through the process group; labels and the partition are the same everywhere."""
    import pandas as pd
    from moc_amd import datasets as DS
    ds = DS.Generic_MIL_Dataset(csv_path=os.path.join(root, "t.csv"), data_dir=os.path.join(root, "bags"), print_info=False,
                                label_dict={"A": 0, "B": 1})
    tr, va, te = ds.return_splits(from_id=False, csv_path=os.path.join(root, "splits_0.csv"), repeat_num=7)
    train = DS.to_sharded(tr, "cpu", rank, world, train=True)
    val = DS.to_sharded(va, "cpu", rank, world)
    return dict(lo=train.lo, hi=train.hi, per=train.per, all_sizes=train.all_sizes, my_ids=train.my_ids, labels=train.all_labels,
                n_local=0 if train.local is None else len(train.local.sizes), local_sizes=[] if train.local is None else train.local.sizes,
                val_ids=val.my_ids, val_lists=val.index_lists, val_local=None if val.local is None else val.local.sizes,
                repeat=train.repeat_num, length=len(train))


def test_to_sharded_reads_blocks_and_agrees_on_the_layout(tmp_path):
    import pandas as pd
    from moc_amd import datasets as DS
    sizes = [30, 45, 12, 60, 25, 33, 51]
    rows = []
    for i, n in enumerate(sizes):
        sid = f"s{i}"
        DS.write_bag(str(tmp_path / "bags"), sid, torch.randn(n, 256), fmt="pt" if i % 2 else "npy")
        rows.append((f"p{i}", sid, "A" if i % 2 else "B"))
    for i in range(3):
        DS.write_bag(str(tmp_path / "bags"), f"v{i}", torch.randn(10 + i, 256), fmt="npy")
        rows.append((f"pv{i}", f"v{i}", "A"))
    pd.DataFrame(rows, columns=["case_id", "slide_id", "label"]).to_csv(tmp_path / "t.csv", index=False)
    pd.DataFrame({"train": pd.Series([f"s{i}" for i in range(7)]), "val": pd.Series([f"v{i}" for i in range(3)]),
                  "test": pd.Series([f"v{i}" for i in range(3)])}).to_csv(tmp_path / "splits_0.csv")
    out = _run(_sharded_driver_inputs, 2, str(tmp_path))
    for r in (0, 1):
        assert out[r]["all_sizes"] == sizes and out[r]["labels"] == [1, 0, 1, 0, 1, 0, 1] and out[r]["per"] == 4
        assert out[r]["repeat"] is None and out[r]["length"] == 7          # repeat_num == number of slides: plain epochs
        assert out[r]["val_lists"] == [[0, 1], [2]]
    assert (out[0]["lo"], out[0]["hi"], out[1]["lo"], out[1]["hi"]) == (0, 4, 4, 7)
    assert out[0]["local_sizes"] == sizes[:4] and out[1]["local_sizes"] == sizes[4:]
    assert out[0]["val_local"] == [10, 11] and out[1]["val_local"] == [12]


# ------------------------------------------------------------------ exact-sequential hand-over (round 3): layout + unpadded gather
def test_more_ranks_than_train_slides_is_refused_on_every_rank_before_any_collective():
    """world 3, two train slides (NSCLC 1-shot has 2): decided from (n, world) alone, so every rank raises the same
    ValueError in the constructor -- no process group exists here at all, so nothing collective can have been entered."""
    for rank in range(3):
        with pytest.raises(ValueError, match="has 2 slide.*3 ranks"):
            mdist.SeqShardedBags([], [900, 1100], [0, 1], "cpu", rank, 3)


def test_seq_layout_packs_every_rank_to_the_largest_sum():
    mx, row_off, n_sel = mdist.seq_layout([[5, 7], [4], [9, 1]])
    assert mx == 12 and n_sel == [5, 7, 4, 9, 1]
    assert row_off == [0, 5, 12, 24, 33, 34]                  # rank r's piece starts at r * mx, its slides back to back
    mx, row_off, n_sel = mdist.seq_layout([[3], [], [2, 2]])  # a rank with no slide in this (partial) pass
    assert mx == 4 and row_off == [0, 8, 10, 12] and n_sel == [3, 2, 2]
    with pytest.raises(AssertionError, match="selected no row"):
        mdist.seq_layout([[3, 0]])


def _count(g):
    return 3 + (7 * g) % 11


def _seq_exchange_worker(rank, world, n, m, cap, D, C):
    """What _seq_issue does after phase A, on CPU tensors over gloo: every rank 'selected' _count(g) rows of slide g,
    row t of slide g holding the value 1000 g + t; after the hand-over EVERY rank must hold every slide's rows, in loader
    order, at the offsets of seq_layout -- having sent only its own unpadded piece."""
    from moc_amd.engine import CompactBatch
    blocks = mdist.block_lists(n, world)
    per = max(len(b) for b in blocks)
    n_by_rank = [max(0, min(m, b[-1] + 1) - b[0]) for b in blocks]
    nk = 2 * C + 2
    st = {"send_feat": torch.zeros(per * cap, D), "send_cand": torch.zeros(per * cap, nk), "send_nsel": torch.zeros(per, dtype=torch.int32),
          "all_nsel": torch.zeros(world * per, dtype=torch.int32), "nsel_host": torch.zeros(world * per, dtype=torch.int32),
          "recv_cand": torch.zeros(world * per * cap, nk)}
    recv = st["send_feat"] if world == 1 else torch.zeros(world * per * cap, D)
    st["compact"] = CompactBatch(m, world * per * cap, cap, D, torch.float32, C, C + 4, 400, 10, "cpu", X=recv)
    o = 0
    for i in range(n_by_rank[rank]):
        g = blocks[rank][i]
        c = _count(g)
        st["send_nsel"][i] = c
        vals = 1000.0 * g + torch.arange(c, dtype=torch.float32)
        st["send_feat"][o:o + c] = vals[:, None]
        st["send_cand"][o:o + c] = vals[:, None] + 0.001 * torch.arange(nk)[None, :]
        o += c
    mx, row_off, n_sel = mdist.seq_exchange(st, n_by_rank, per, world)
    cb = st["compact"]
    assert n_sel == [_count(g) for g in range(m)]
    assert mx == max(sum(_count(g) for g in blocks[r][:n_by_rank[r]]) for r in range(world)) and o <= mx
    for g in range(m):
        rows = cb.X[row_off[g]:row_off[g] + n_sel[g]]
        exp = 1000.0 * g + torch.arange(n_sel[g], dtype=torch.float32)
        assert torch.equal(rows, exp[:, None].expand(-1, D)), f"slide {g}: rows"
        for k in range(nk):
            assert torch.allclose(cb.cand[k, row_off[g]:row_off[g] + n_sel[g]], exp + 0.001 * k), f"slide {g}: candidate column {k}"
    cb.set_layout(row_off)
    assert cb.row_off.tolist() == row_off and [cb._row_off_c[i] for i in range(m + 1)] == row_off
    return dict(mx=mx, sent=o, padded=per * cap, row_off=row_off)


@pytest.mark.parametrize("world,n,m", [(4, 11, 11), (4, 11, 9), (8, 12, 12), (8, 32, 32), (8, 9, 3), (2, 2, 2)])
def test_seq_exchange_gathers_unpadded_pieces_in_loader_order(world, n, m):
    """gloo, world 4 and 8, uneven splits (n % world != 0), partial passes that leave trailing ranks nothing to send."""
    out = _run(_seq_exchange_worker, world, n, m, 16, 8, 2)
    assert len({tuple(out[r]["row_off"]) for r in range(world)}) == 1, "ranks disagree on the layout"
    assert all(out[r]["mx"] == out[0]["mx"] and out[r]["sent"] <= out[r]["mx"] < out[r]["padded"] for r in range(world))


def _direct_rccl_worker(rank, world):
    """Rank 1 cannot load librccl: BOTH ranks must come out with ok == False (and neither may hang in a collective
    the other skipped).  gloo: the agreement words travel through torch.distributed."""
    if rank == 1:
        os.environ["MOC_RCCL_LIB"] = "/nonexistent/librccl.so"
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        d = mdist.DirectRccl("cpu")
    return (d.ok, [str(x.message) for x in w])


def test_direct_rccl_setup_failure_on_one_rank_is_agreed_by_all():
    out = _run(_direct_rccl_worker, 2)
    assert out[0][0] is False and out[1][0] is False
    assert any("direct RCCL unavailable" in m for m in out[0][1]) and any("librccl" in m for m in out[1][1])
