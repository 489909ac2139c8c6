"""Which call of train() holds the periodic multi-millisecond host stall (diagnostic)."""
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
log = []
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        log.append((name, (time.perf_counter() - t0) * 1e3))
        return r
    setattr(mod, name, g)
for mod, name in ((M, "_resident_pass_setup"), (M, "resident_pass_done"), (engine, "train_steps"), (engine, "draw_row_masks_from"),
                  (M, "MetaState")):
    wrap(mod, name)
_pa = engine.SlideBatch.phase_a
def pa(self, bank):
    t0 = time.perf_counter(); _pa(self, bank); log.append(("phase_a", (time.perf_counter() - t0) * 1e3))
engine.SlideBatch.phase_a = pa
_sm = engine.SlideBatch.set_mask
def sm(self, *a):
    t0 = time.perf_counter(); _sm(self, *a); log.append(("set_mask", (time.perf_counter() - t0) * 1e3))
engine.SlideBatch.set_mask = sm
for e in range(3):
    M.train(model, res, opt, dev, args)
torch.cuda.synchronize()
for e in range(60):
    log.clear()
    t0 = time.perf_counter()
    M.train(model, res, opt, dev, args)
    dt = (time.perf_counter() - t0) * 1e3
    if dt > 2.0:
        print(f"epoch {e}: {dt:.2f} ms:", [(n, round(v, 2)) for n, v in log])
torch.cuda.synchronize()
import statistics
print("typical:", [(n, round(v, 3)) for n, v in log])
