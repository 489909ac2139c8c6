"""Where the host time of one evaluation() call over a resident split goes (cProfile over many calls):
python scripts/diag_eval_host.py [C] [slides] [rows]"""
import cProfile, pstats, sys, os, types, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(8)
from moc_amd import main_moc as M, synth
dev = torch.device("cuda:0")
C = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = int(sys.argv[2]) if len(sys.argv) > 2 else 202
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 15000
D, j, K = 512, 400, 10
W, We = synth.make_bank(1234, D, C)
M.set_classifier_bank(W.to(dev), We.to(dev))
bags = [synth.make_bag_device(1234 + i, rows, D, We, C, i % C, dev, torch.bfloat16) for i in range(n)]
res = M.ResidentBags(bags, [i % C for i in range(n)], dev)
args = types.SimpleNamespace(disable_tqdm=True, n_classes=C, topj=j, topk=K, discard_classifiers=[], pretrain="conch", ablation_study="none")
torch.manual_seed(0)
model = M.senet(D, 4).to(dev)
for _ in range(3):
    M.evaluation(model, res, dev, args)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    M.evaluation(model, res, dev, args)
torch.cuda.synchronize()
print(f"evaluation(): {(time.perf_counter() - t0) / 10 * 1e6:.0f} us per call = {n / ((time.perf_counter() - t0) / 10):.0f} slides/s")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    M.evaluation(model, res, dev, args)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print(s.getvalue()[:5000])
