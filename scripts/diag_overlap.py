"""How much of phase A hides behind phase B when both are simply kept busy on two streams (no events between
them)?  Separates what the hardware can overlap from what the pass-ahead logic achieves (diagnostic)."""
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moc_amd import engine, main_moc as M, synth  # noqa: E402

dev = torch.device("cuda:0")
C, D, j, K = 2, 512, 400, 10
W, We = synth.make_bank(1234, D, C)
M.set_classifier_bank(W.to(dev), We.to(dev))
bags = [synth.make_bag_device(1234 + i, 15000, D, We, C, i % C, dev, torch.bfloat16) for i in range(32)]
res = M.ResidentBags(bags, [i % C for i in range(32)], dev)
args = types.SimpleNamespace(disable_tqdm=True, n_classes=C, topj=j, topk=K, discard_classifiers=[], pretrain="conch",
                             ablation_study="none")
torch.manual_seed(0)
model = M.senet(D, 4).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
for _ in range(3):
    M.train(model, res, opt, dev, args)
torch.cuda.synchronize()
bank = M._bank_for(res.X, dev)
plan = res.train_plan(bank.C, bank.Ce, j, K, [])
bB, bA = plan["batches"][0], plan["batches"][1]
lab = plan["labels"]
meta = engine.MetaState(model, opt)
use = engine.train_use_bits([])
side = torch.cuda.Stream()
bB.phase_a(bank); bA.phase_a(bank)
torch.cuda.synchronize()

def run(n, do_a, do_b):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if do_b:
            engine.train_steps(bB, meta, lab, 0, 32, use)
        if do_a:
            with torch.cuda.stream(side):
                bA.phase_a(bank)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

def run_events(n, side_waits_mark, main_waits_done, host_waits_mark=False):
    """The pass-ahead loop's event structure, minus the host-side mask draw: A(e+1) may start when B(e-1) is
    done; B(e+1) when A(e+1) is done."""
    torch.cuda.synchronize()
    marks = [None, None]
    done = None
    t0 = time.perf_counter()
    for e in range(n):
        main = torch.cuda.current_stream()
        if main_waits_done and done is not None:
            main.wait_event(done)
        engine.train_steps(bB, meta, lab, 0, 32, use)
        m = torch.cuda.Event(); m.record(main); marks[e & 1] = m
        if host_waits_mark and marks[(e + 1) & 1] is not None:
            marks[(e + 1) & 1].synchronize()                 # the HOST waits for B(e-1); no packet on the side stream
        with torch.cuda.stream(side):
            if side_waits_mark and marks[(e + 1) & 1] is not None:
                side.wait_event(marks[(e + 1) & 1])
            bA.phase_a(bank)
            done = torch.cuda.Event(); done.record(side)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


n = 40
run(5, True, True)
a, b, ab = run(n, True, False), run(n, False, True), run(n, True, True)
ev_all, ev_side, ev_main = run_events(n, True, True), run_events(n, True, False), run_events(n, False, True)
ev_host = run_events(n, False, True, host_waits_mark=True)
print(f"host waits for the mark instead of the side stream: {ev_host:.3f} ms")
print(f"with the loop's events: both waits {ev_all:.3f} ms, only side-waits-mark {ev_side:.3f}, only main-waits-done {ev_main:.3f}")
print(f"per epoch: phase A alone {a:.3f} ms, phase B alone {b:.3f} ms, both on two streams {ab:.3f} ms "
      f"(sum {a + b:.3f}; hidden {100 * (a + b - ab) / a:.0f} % of phase A)")
