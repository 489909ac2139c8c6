"""Does a kernel of a given footprint get CUs while the persistent score kernel is running on another stream?
(diagnostic for DESIGN.md section 11).  Launches the score pass (32 slides x 15,000 x 512 bf16, ~50 us) on a side
stream and, a few microseconds later, a spin kernel of ~10 us on the main stream; reports the spin kernel's
start delay and duration for several footprints."""
import ctypes as C
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moc_amd import engine, synth  # noqa: E402

src = os.path.join(ROOT, "scripts", "native", "coresidency.hip")
so = "/tmp/libcoresidency.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", so, src])
lib = C.CDLL(so)
lib.coresidency_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_ulonglong, C.c_void_p, C.c_void_p, C.c_void_p]

dev = torch.device("cuda:0")
Cc, D = 2, 512
W, We = synth.make_bank(1, D, Cc)
bags = torch.cat([synth.make_bag_device(10 + i, 15000, D, We, Cc, i % Cc, dev, torch.bfloat16) for i in range(32)])
batch = engine.SlideBatch(bags, [15000] * 32, Cc, Cc + 4, 400, 10)
bank = engine.Bank.get(W.to(dev), We.to(dev), torch.bfloat16, dev)
side = torch.cuda.Stream()
out = torch.zeros(4, device=dev)
stamps = torch.zeros(2, dtype=torch.int64, device=dev)
for _ in range(3):
    batch.scores(bank)
torch.cuda.synchronize()

def trial(grid, threads, nv, lds, with_scores):
    res = []
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if with_scores:
            with torch.cuda.stream(side):
                batch.scores(bank)
                batch.scores(bank)
        time.sleep(20e-6)
        e0.record()
        lib.coresidency_launch(grid, threads, nv, lds, 1000, out.data_ptr(), stamps.data_ptr(), torch.cuda.current_stream().cuda_stream)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3)
    return sorted(res)[len(res) // 2]

for name, grid, threads, nv, lds in (("16 x 1024 thr, nv 80, 65 KB (today's step kernel)", 16, 1024, 80, 65 * 1024),
                                     ("32 x 512 thr, nv 80, 40 KB", 32, 512, 80, 40 * 1024),
                                     ("86 x 256 thr, nv 220, 22 KB (today's forward)", 86, 256, 220, 22 * 1024),
                                     ("86 x 256 thr, nv 100, 22 KB", 86, 256, 100, 22 * 1024),
                                     ("64 x 256 thr, nv 48, 40 KB", 64, 256, 48, 40 * 1024)):
    alone = trial(grid, threads, nv, lds, False)
    beside = trial(grid, threads, nv, lds, True)
    print(f"{name:58s}: alone {alone:6.1f} us, beside the score pass {beside:6.1f} us")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    batch.scores(bank)
e1.record()
torch.cuda.synchronize()
print(f"score pass alone: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch (mask_compact included)")
