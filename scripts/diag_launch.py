"""Per-launch host latency: looks for periodic stalls in the HIP launch path (diagnostic)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moc_amd import _lib, engine
dev = torch.device("cuda:0")
L = _lib.lib()
pooled = torch.randn(4, 2, device=dev); lab = torch.zeros(4, dtype=torch.int64, device=dev)
loss = torch.empty(4, device=dev); pred = torch.empty(4, dtype=torch.int32, device=dev)
x = torch.zeros(1024, device=dev)
def ours(): L.moc_ce_loss(pooled.data_ptr(), lab.data_ptr(), 4, 2, loss.data_ptr(), pred.data_ptr(), engine._stream())
def theirs(): x.add_(1.0)
for name, fn, sync_every in (("ours", ours, 0), ("torch", theirs, 0), ("ours+sync100", ours, 100), ("mixed", None, 0)):
    torch.cuda.synchronize()
    stalls, t_all = [], time.perf_counter()
    for i in range(6000):
        t0 = time.perf_counter()
        if fn is None:
            (ours if i % 2 else theirs)()
        else:
            fn()
        dt = time.perf_counter() - t0
        if dt > 1e-3: stalls.append((i, round(dt * 1e3, 1)))
        if sync_every and i % sync_every == 0: torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(name, "total ms", round((time.perf_counter() - t_all) * 1e3, 1), "stalls", stalls[:12])
