"""Do two HIP streams of one process actually overlap on this box?  (diagnostic)
A chain of small dependent kernels per stream; if the streams run concurrently, two chains take about as
long as one."""
import time
import torch
dev = torch.device("cuda:0")
a = torch.zeros(1 << 16, device=dev)
b = torch.zeros(1 << 16, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def chain(x, n):
    for _ in range(n):
        x.add_(1.0)

def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3

n = 4000
for _ in range(2):
    one = timed(lambda: chain(a, n))
    def two():
        for _ in range(n // 50):
            with torch.cuda.stream(s1):
                chain(a, 50)
            with torch.cuda.stream(s2):
                chain(b, 50)
    both = timed(two)
    # long kernels: one big reduction per stream
    big1 = torch.randn(1 << 28, device=dev)
    big2 = torch.randn(1 << 28, device=dev)
    def long_one():
        for _ in range(20):
            big1.sum()
    def long_two():
        for _ in range(20):
            with torch.cuda.stream(s1):
                big1.sum()
            with torch.cuda.stream(s2):
                big2.sum()
    l1, l2 = timed(long_one), timed(long_two)
print(f"{n} tiny kernels on one stream: {one:.2f} ms; {n} + {n} on two streams: {both:.2f} ms")
print(f"20 x 1 GiB reductions on one stream: {l1:.2f} ms; 20 + 20 on two streams: {l2:.2f} ms")
