"""Where the host time of one train() call over a resident split goes (cProfile over many calls; the GPU work is tiny)."""
import cProfile, pstats, sys, os, types, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(8)
from moc_amd import main_moc as M, synth
dev = torch.device("cuda:0")
C, D, j, K, n = 2, 512, 400, 10, int(sys.argv[1]) if len(sys.argv) > 1 else 20
W, We = synth.make_bank(1234, D, C)
M.set_classifier_bank(W.to(dev), We.to(dev))
bags = [synth.make_bag_device(1234 + i, 3000, D, We, C, i % C, dev, torch.float32) for i in range(n)]
res = M.ResidentBags(bags, [i % C for i in range(n)], dev)
args = types.SimpleNamespace(disable_tqdm=True, n_classes=C, topj=j, topk=K, discard_classifiers=[], pretrain="conch", ablation_study="none")
torch.manual_seed(0)
model = M.senet(D, 4).to(dev); opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
for _ in range(20):
    M.train(model, res, opt, dev, args)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    M.train(model, res, opt, dev, args)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(28)
print(s.getvalue()[:6000])
