"""The meta-steps with NOTHING beside them: phase A once, then passes of 32 steps back to back on the same work arrays
(events around 50 passes).  The gap between this and bench.py's steady state is what phase A costs the loop."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moc_amd import engine, main_moc as M, synth
dev = torch.device("cuda:0")
DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "fp32"]
Cc, D, j, K = 2, 512, 400, 10
W, We = synth.make_bank(1234, D, Cc)
M.set_classifier_bank(W.to(dev), We.to(dev))
bags = [synth.make_bag_device(1234 + i, 15000, D, We, Cc, i % Cc, dev, DT) for i in range(32)]
res = M.ResidentBags(bags, [i % Cc for i in range(32)], dev)
torch.manual_seed(0)
model = M.senet(D, 4).to(dev); opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
bank = M._bank_for(res.X, dev)
plan = res.train_plan(Cc, Cc + 4, j, K, [])
batch, lab = plan["batch"], plan["labels"]
m, kept = engine.draw_row_masks(batch.total); batch.set_mask(m, kept); batch.phase_a(bank)
meta = engine.MetaState(model, opt)
for _ in range(5):
    engine.train_steps(batch, meta, lab, 0, 32, 15)
torch.cuda.synchronize()
for reps in (50, 50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        engine.train_steps(batch, meta, lab, 0, 32, 15)
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ms = e0.elapsed_time(e1)
    print(f"alone: {ms * 1e3 / (reps * 32):.2f} us/step by events ({reps * 32 / (ms * 1e-3):.0f} steps/s); host issue {(t1 - t0) * 1e6 / (reps * 32):.2f} us/step, wall {(t2 - t0) * 1e6 / (reps * 32):.2f}")
