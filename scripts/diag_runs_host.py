import os, sys, time, types
import torch
sys.path.insert(0, "/root/repo")
from moc_amd import engine, main_moc as M, synth, runs as RUNS
dev = torch.device("cuda:0")
Cc, D, j, K, n = 2, 512, 400, 10, 32
W, We = synth.make_bank(1234, D, Cc)
M.set_classifier_bank(W.to(dev), We.to(dev))
args = types.SimpleNamespace(disable_tqdm=True, n_classes=Cc, topj=j, topk=K, discard_classifiers=[], pretrain="conch", ablation_study="none")
R = int(sys.argv[1])
models, opts, splits = [], [], []
for r in range(R):
    bags = [synth.make_bag_device(1234 + 1000 * r + i, 15000, D, We, Cc, i % Cc, dev, torch.float32) for i in range(n)]
    splits.append(M.ResidentBags(bags, [i % Cc for i in range(n)], dev))
    torch.manual_seed(r)
    m = M.senet(D, 4).to(dev)
    models.append(m); opts.append(torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-4))
rs = RUNS.TrainRuns(models, opts, splits, dev, args)
T = {}
def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); T[name] = T.get(name, 0.0) + time.perf_counter() - t0; return r
    return w
rs._draw = timed("draw", rs._draw)
rs._phase_a = timed("phase_a(all)", rs._phase_a)
for b in rs.batches:
    b.use_host_mask = timed("use_host_mask", b.use_host_mask)
    b.phase_a = timed("batch.phase_a", b.phase_a)
for _ in range(5): rs.train_pass()
torch.cuda.synchronize(); T.clear()
t0 = time.perf_counter()
for _ in range(20): rs.train_pass()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"R={R}: host {1e6*(t1-t0)/20:.0f} us per pass, wall {1e6*(t2-t0)/20:.0f};", {k: round(v/20*1e6) for k, v in T.items()})
