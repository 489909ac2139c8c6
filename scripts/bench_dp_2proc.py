"""Two data-parallel ranks on ONE GPU (the box has one): steady-state time of the synchronous step
with the exchange inside the step kernel vs the collective between the launches.  The "peers" share
a device, so this measures the protocol (fences, flags, launches) -- not xGMI.

    python scripts/bench_dp_2proc.py [--exchange auto|rccl] [--epochs 40]
"""
import argparse
import os
import socket
import sys
import time
import types

import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, a, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      MOC_DP_EXCHANGE=a.exchange)
    import torch.distributed as dist
    from moc_amd import main_moc as M, synth, dist as mdist
    torch.set_num_threads(4)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    C, D = 2, 512
    W, We = synth.make_bank(1234, D, C)
    M.set_classifier_bank(W.to(dev), We.to(dev))
    bags = [synth.make_bag_device(1234 + 1000 * rank + i, a.patches, D, We, C, i % C, dev, torch.bfloat16)
            for i in range(a.slides)]
    res = M.ResidentBags(bags, [i % C for i in range(a.slides)], dev)
    args = types.SimpleNamespace(disable_tqdm=True, n_classes=C, topj=400, topk=10, discard_classifiers=[],
                                 pretrain="conch", ablation_study="none")
    torch.manual_seed(0)
    model = M.senet(D, 4).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    for _ in range(5):
        mdist.train_dp(model, res, opt, dev, args)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.epochs):
        mdist.train_dp(model, res, opt, dev, args)
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    q.put((rank, dt / (a.epochs * a.slides) * 1e6, mdist.train_dp.exchange, mdist.exchange_error()))
    mdist.shutdown()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--exchange", default="auto")
    ap.add_argument("--epochs", type=int, default=40)
    ap.add_argument("--slides", type=int, default=32)
    ap.add_argument("--patches", type=int, default=15000)
    a = ap.parse_args()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, a, q)) for r in range(2)]
    for p in procs:
        p.start()
    for _ in range(2):
        rank, us, ex, err = q.get(timeout=600)
        print(f"rank {rank}: {us:.2f} us per synchronous step (phase A included), exchange={ex}, error={err}")
    for p in procs:
        p.join(timeout=60)


if __name__ == "__main__":
    main()
