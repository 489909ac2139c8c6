"""What a plain streaming kernel reaches on this box (calibration for roofline.frac)."""
import torch, time
dev = torch.device("cuda:0")
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e-3)
    ts.sort(); return ts[len(ts) // 2]
for mb in (256, 1024, 3072):
    n = mb * 1024 * 1024 // 2
    x = torch.randn(n, device=dev, dtype=torch.float32).to(torch.bfloat16) if mb <= 1024 else torch.zeros(n, device=dev, dtype=torch.bfloat16).normal_()
    y = torch.empty_like(x)
    t = timeit(lambda: y.copy_(x)); print(f"{mb:5d} MB bf16 copy (read+write): {2 * x.numel() * 2 / t / 1e12:5.2f} TB/s total, {x.numel() * 2 / t / 1e12:5.2f} TB/s read side")
    t = timeit(lambda: x.sum()); print(f"{mb:5d} MB bf16 sum  (read only) : {x.numel() * 2 / t / 1e12:5.2f} TB/s")
    xf = x.view(torch.float32)
    t = timeit(lambda: xf.sum()); print(f"{mb:5d} MB f32  sum  (read only) : {xf.numel() * 4 / t / 1e12:5.2f} TB/s")
    t = timeit(lambda: torch.amax(xf)); print(f"{mb:5d} MB f32  amax (read only) : {xf.numel() * 4 / t / 1e12:5.2f} TB/s")
