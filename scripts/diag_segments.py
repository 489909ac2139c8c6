"""Which part of train() stalls?  (diagnostic)"""
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
keep = None
variant = sys.argv[1] if len(sys.argv) > 1 else "base"
for e in range(30):
    T = [time.perf_counter()]
    def tick(): T.append(time.perf_counter())
    X, sizes, x_starts, labels = M._collect(res, dev, args); tick()
    masks = [torch.rand(n) > 0.5 for n in sizes]; tick()
    meta = engine.MetaState(model, opt); tick()
    bank = M._bank_for(X, dev); tick()
    m = torch.cat(masks); tick()
    if variant == "pinned":
        m = m.to(torch.uint8).pin_memory()
    batch = engine.SlideBatch(X, sizes, 2, 6, j, K, [], mask=m, x_starts=x_starts); tick()
    lab = torch.tensor(labels, dtype=torch.int64).to(dev, non_blocking=True); tick()
    batch.phase_a(bank); tick()
    batch.meta_ws(); tick()
    engine.train_steps(batch, meta, lab, 0, 32, 15); tick()
    if variant != "nokeep":
        keep = (batch, lab)
    else:
        del batch
    tick()
    d = [round((b - a) * 1e3, 2) for a, b in zip(T, T[1:])]
    if sum(d) > 20:
        print("epoch", e, "collect/masks/meta/bank/cat/batch/lab/phaseA/ws/steps/keep =", d)
torch.cuda.synchronize()
print("done", variant)
