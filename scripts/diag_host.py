"""Where the host time of one resident train() pass goes in steady state (diagnostic)."""
import os, sys, time, types
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
acc = {}
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name
    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        acc[label] = acc.get(label, 0.0) + (time.perf_counter() - t0)
        return r
    setattr(obj, name, g)
wrap(M, "_resident_pass_setup"); wrap(M, "resident_pass_done"); wrap(M, "_issue_phase_a"); wrap(engine, "train_steps")
wrap(engine, "draw_row_masks_from"); wrap(M, "MetaState"); wrap(torch, "get_rng_state"); wrap(torch, "set_rng_state")
wrap(torch.cuda.Event, "synchronize", "Event.synchronize"); wrap(torch.cuda.Event, "record", "Event.record")
wrap(engine.SlideBatch, "phase_a", "SlideBatch.phase_a"); wrap(M, "_bank_for")
for _ in range(10):
    M.train(model, res, opt, dev, args)
torch.cuda.synchronize()
acc.clear()
n = 60
t0 = time.perf_counter()
for _ in range(n):
    M.train(model, res, opt, dev, args)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host per pass {1e3 * (t1 - t0) / n:.3f} ms (+ {1e3 * (t2 - t1):.2f} ms to drain at the end)")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:28s} {1e3 * v / n:.3f} ms")
