"""cProfile of main_moc.evaluation over resident slides: where the host time of a pass goes."""
import cProfile, pstats, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.set_num_threads(8)
from moc_amd import main_moc as M, synth
dev = torch.device("cuda:0")
C, D, n = 2, 512, 202
W, We = synth.make_bank(1234, D, C)
M.set_classifier_bank(W.to(dev), We.to(dev))
bags = [synth.make_bag_device(777 + i, 15000, D, We, C, i % C, dev, torch.bfloat16) for i in range(n)]
res = M.ResidentBags(bags, [i % C for i in range(n)], dev)
del bags
args = types.SimpleNamespace(disable_tqdm=True, n_classes=C, topj=400, topk=10, discard_classifiers=[], pretrain="conch", ablation_study="none")
torch.manual_seed(0)
model = M.senet(D, 4).to(dev)
for _ in range(3):
    M.evaluation(model, res, dev, args)
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    M.evaluation(model, res, dev, args)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
