"""Where an evaluation pass spends its time: GPU kernels vs the host tail (main_moc._metrics)."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.set_num_threads(8)
from moc_amd import main_moc as M, synth
dev = torch.device("cuda:0")
C, D = 2, 512
W, We = synth.make_bank(1234, D, C)
M.set_classifier_bank(W.to(dev), We.to(dev))
n = 202
bags = [synth.make_bag_device(777 + i, 15000, D, We, C, i % C, dev, torch.bfloat16) for i in range(n)]
res = M.ResidentBags(bags, [i % C for i in range(n)], dev)
del bags
args = types.SimpleNamespace(disable_tqdm=True, n_classes=C, topj=400, topk=10, discard_classifiers=[], pretrain="conch", ablation_study="none")
torch.manual_seed(0)
model = M.senet(D, 4).to(dev)
acc = {"metrics": 0.0, "pass": 0.0}
orig_metrics, orig_pass = M._metrics, M._eval_pass
def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); acc[name] += time.perf_counter() - t0; return r
    return w
M._metrics = timed("metrics", orig_metrics)
M._eval_pass = timed("pass", orig_pass)
for _ in range(3):
    M.evaluation(model, res, dev, args)
for k in acc: acc[k] = 0.0
torch.cuda.synchronize(); t0 = time.perf_counter()
R = 20
for _ in range(R):
    M.evaluation(model, res, dev, args)
torch.cuda.synchronize(); tot = time.perf_counter() - t0
print(f"evaluation: {tot / R * 1e6:.0f} us per pass of {n} slides = {n * R / tot:.0f} slides/s;  _eval_pass {acc['pass'] / R * 1e6:.0f} us (launches + GPU + D2H), _metrics {acc['metrics'] / R * 1e6:.0f} us (host)")
# GPU-only time of a pass: events around the launches
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
M._metrics = lambda *a, **k: {"loss": 0.0, "acc": 0.0, "auc": 0.0}
gp = []
for _ in range(10):
    torch.cuda.synchronize(); s.record(); M.evaluation(model, res, dev, args); e.record(); torch.cuda.synchronize(); gp.append(s.elapsed_time(e) * 1e3)
print(f"GPU span of a pass (first launch -> D2H done): median {sorted(gp)[5]:.0f} us")
# host time to queue a pass (no wait for the GPU): everything before the device->host copy
import torch as _t
orig_cat = _t.cat
marks = []
def cat_mark(ts, dim=0):
    r = orig_cat(ts, dim)
    if r.is_cuda and r.dim() == 2 and r.size(0) == n:
        marks.append(time.perf_counter())
    return r
_t.cat = cat_mark
M._eval_pass = orig_pass
q = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); M.evaluation(model, res, dev, args); q.append((marks[-1] - t0) * 1e6)
print(f"host time to queue a pass: median {sorted(q)[5]:.0f} us")
