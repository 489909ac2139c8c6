"""Score-pass micro-benchmark: GB/s of moc_scores alone (events on the launch stream)."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_num_threads(8)
from moc_amd import engine, synth
from moc_amd._lib import lib, ptr, check
dev = torch.device("cuda:0")
Cc, D = 2, 512
if len(sys.argv) > 2:                      # python scripts/bench_scores.py <classes> <dim> [slides patches dtype masked]
    Cc, D = int(sys.argv[1]), int(sys.argv[2])
W, We = synth.make_bank(1, D, Cc)
def run(n_slides, N, dtype, masked, reps=20):
    X = torch.randn(n_slides * N, D, device=dev).to(dtype)
    mask = (torch.rand(n_slides * N) > 0.5) if masked else None
    b = engine.SlideBatch(X, [N] * n_slides, Cc, Cc + 4, 400, 10, mask=mask)
    bank = engine.Bank.get(W, We, dtype, dev)
    check(lib().moc_mask_compact(C.byref(b.c), engine._stream()), "mc")
    for _ in range(3): check(lib().moc_scores(C.byref(b.c), ptr(bank.image), engine._stream()), "s")
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); check(lib().moc_scores(C.byref(b.c), ptr(bank.image), engine._stream()), "s"); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(c) * 1e3 for a, c in evs)
    by = b.kept_rows_host * D * X.element_size()
    med = ts[len(ts) // 2]
    print(f"slides={n_slides:4d} N={N:6d} {str(dtype)[6:]:8s} masked={int(masked)}  bytes={by/1e6:8.1f} MB  median={med:8.1f} us  min={ts[0]:8.1f} us  -> {by/med/1e6:6.2f} TB/s (median) {by/ts[0]/1e6:6.2f} (best)")
if len(sys.argv) > 6:
    run(int(sys.argv[3]), int(sys.argv[4]), getattr(torch, sys.argv[5]), bool(int(sys.argv[6])))
    sys.exit(0)
for cfg in [(32, 15000, torch.bfloat16, True), (32, 15000, torch.bfloat16, False), (202, 15000, torch.bfloat16, False),
            (1, 15000, torch.bfloat16, False), (8, 15000, torch.bfloat16, False),
            (32, 15000, torch.float32, True), (202, 15000, torch.float32, False)]:
    run(*cfg)
