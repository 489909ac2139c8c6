"""Aggregate rate of R batched runs (moc_amd.runs) on the default workload shape: R x 32 slides x 15,000 x 512."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import types
from moc_amd import engine, main_moc as M, synth
dev = torch.device("cuda:0")
DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[sys.argv[2] if len(sys.argv) > 2 else "fp32"]
Cc, D, j, K, n = 2, 512, 400, 10, 32
W, We = synth.make_bank(1234, D, Cc)
M.set_classifier_bank(W.to(dev), We.to(dev))
args = types.SimpleNamespace(disable_tqdm=True, n_classes=Cc, topj=j, topk=K, discard_classifiers=[], pretrain="conch", ablation_study="none")
for R in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,8").split(",")]:
    models, opts, splits = [], [], []
    for r in range(R):
        bags = [synth.make_bag_device(1234 + 1000 * r + i, 15000, D, We, Cc, i % Cc, dev, DT) for i in range(n)]
        splits.append(M.ResidentBags(bags, [i % Cc for i in range(n)], dev, cache_scores=(os.environ.get("CACHE") == "1")))
        torch.manual_seed(r)
        m = M.senet(D, 4).to(dev)
        models.append(m); opts.append(torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-4))
    for _ in range(6):
        rs = M.train_runs(models, splits, opts, dev, args)
    torch.cuda.synchronize()
    E = 40
    engine.SCORE_EVENTS = []
    t0 = time.perf_counter()
    for _ in range(E):
        M.train_runs(models, splits, opts, dev, args)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ev, engine.SCORE_EVENTS = engine.SCORE_EVENTS, None
    ms = [a.elapsed_time(b) for a, b, _ in ev]; by = [c for _, _, c in ev]
    sc = sum(by) / sum(ms) / 1e6 if ms else 0.0
    print(f"R={R}: {R * n * E / dt:9.0f} meta-steps/s aggregate ({dt / E * 1e6:7.1f} us per pass of {R * n} steps, {dt / (E * n) * 1e6:6.2f} us per lockstep step); "
          f"score pass live {sc:6.0f} GB/s ({sum(ms) / len(ms) * 1e3 if ms else 0:7.1f} us per launch)", flush=True)
    del rs, models, opts, splits
    M._run_sets.clear()
    torch.cuda.empty_cache()
