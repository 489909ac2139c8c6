"""Host-side timing of train() epochs (diagnostic, not part of the product)."""
import os, sys, time, types, cProfile, pstats, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moc_amd import main_moc as M, synth, engine

dev = torch.device("cuda:0")
C, D, j, K = 2, 512, 400, 10
W, We = synth.make_bank(1234, D, C)
M.set_classifier_bank(W.to(dev), We.to(dev))
bags = [synth.make_bag_device(1234 + i, 15000, D, We, C, i % C, dev, torch.bfloat16) for i in range(32)]
res = M.ResidentBags(bags, [i % C for i in range(32)], dev)
args = types.SimpleNamespace(disable_tqdm=True, n_classes=C, topj=j, topk=K, discard_classifiers=[], pretrain="conch", ablation_study="none")
torch.manual_seed(0)
model = M.senet(D, 4).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
mode = sys.argv[1] if len(sys.argv) > 1 else "async"
if mode.startswith("threads"):
    torch.set_num_threads(int(mode[7:]))
try:
    print("affinity", len(os.sched_getaffinity(0)), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip(), "torch threads", torch.get_num_threads())
except Exception as ex:
    print("cgroup probe failed", ex)
import gc
_gc_t = {}
def _gc_cb(phase, info):
    if phase == "start": _gc_t["t"] = time.perf_counter()
    else:
        d = (time.perf_counter() - _gc_t["t"]) * 1e3
        if d > 1.0: print(f"  gc gen{info['generation']} took {d:.1f} ms, collected {info['collected']}")
gc.callbacks.append(_gc_cb)
if mode == "nogc": gc.disable()
if mode == "freeze": gc.collect(); gc.freeze()
for e in range(3):
    M.train(model, res, opt, dev, args)
torch.cuda.synchronize()
ts = []
for e in range(24):
    t0 = time.perf_counter()
    if mode == "profile" and e == 11:
        pr = cProfile.Profile(); pr.enable()
    M.train(model, res, opt, dev, args)
    if mode == "profile" and e == 11:
        pr.disable(); s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue())
    t1 = time.perf_counter()
    if mode == "sync":
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append((round((t1 - t0) * 1e3, 2), round((t2 - t1) * 1e3, 2)))
torch.cuda.synchronize()
print(mode, "host_ms/sync_ms per epoch:", ts)
print("mem allocated MB", torch.cuda.memory_allocated() / 1e6, "reserved", torch.cuda.memory_reserved() / 1e6)
