"""Row f4: time the gated-attention pooling step (moc_gated_attention_pool) against its bound -- the bf16
matrix pipe at six bf16 products per fp32-exact product (4*N*L*D*6 flops at 2.5 PFLOP/s dense; the fp32
matrix pipe, 157 TFLOP/s, is the rate a plain fp32 kernel is held to) -- and against torch on the same
GPU and on the host cores.

    python scripts/bench_attention.py [--n 15000] [--l 512] [--d 384] [--k 1]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moc_amd import engine  # noqa: E402

F32_MFMA_PEAK = 157.3e12      # 256 CUs x 2.4 GHz x 256 flop/clk/CU (MI355X_MICROARCH.md: fp32 matrix)
BF16_MFMA_PEAK = 2.5e15       # dense bf16 (MI355X_MICROARCH.md); v_mfma_f32_16x16x32_bf16 = 16 cycles per SIMD


def torch_step(h, Wa, ba, Wb, bb, Wc, bc):
    a = torch.tanh(torch.nn.functional.linear(h, Wa, ba))
    b = torch.sigmoid(torch.nn.functional.linear(h, Wb, bb))
    A = torch.nn.functional.linear(a * b, Wc, bc).t()
    return A, torch.softmax(A, dim=1) @ h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=15000)
    ap.add_argument("--l", type=int, default=512)
    ap.add_argument("--d", type=int, default=384)
    ap.add_argument("--k", type=int, default=1)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    h = torch.relu(torch.randn(a.n, a.l, generator=g))
    Wa, Wb = torch.randn(a.d, a.l, generator=g) * 0.05, torch.randn(a.d, a.l, generator=g) * 0.05
    ba, bb = torch.zeros(a.d), torch.zeros(a.d)
    Wc, bc = torch.randn(a.k, a.d, generator=g) * 0.1, torch.zeros(a.k)
    host = (h, Wa, ba, Wb, bb, Wc, bc)
    gpu = [t.to(dev) for t in host]
    flops = 4.0 * a.n * a.l * a.d

    def timed(fn, iters):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / iters * 1e3      # us

    us = timed(lambda: engine.gated_attention_pool(*gpu), a.iters)
    us_t = timed(lambda: torch_step(*gpu), a.iters)
    # forward + backward (gradient arriving at A_raw and at M, every input's gradient asked for)
    from moc_amd.model_clam import gated_attention_pool
    uA, uM = torch.randn(a.k, a.n, device=dev), torch.randn(a.k, a.l, device=dev)

    def fb(fn):
        ins = [t.detach().requires_grad_(True) for t in gpu]
        A, M = fn(*ins)
        torch.autograd.grad([A, M], ins, [uA, uM])

    us_fb, us_fb_t = timed(lambda: fb(gated_attention_pool), a.iters), timed(lambda: fb(torch_step), a.iters)
    A_saved = engine.gated_attention_pool(*gpu)[0]
    us_b = timed(lambda: engine.gated_attention_backward(*gpu[:6], A_saved, uA, uM), a.iters)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        torch_step(*host)
    cpu_us = (time.perf_counter() - t0) / reps * 1e6
    print(f"N={a.n} L={a.l} D={a.d} K={a.k}: {flops / 1e9:.2f} GFLOP per bag")
    print(f"  moc_gated_attention_pool : {us:8.1f} us  {flops / us / 1e6:6.1f} TFLOP/s fp32-exact = {6 * flops / us / 1e6:6.1f} TFLOP/s of bf16 products "
          f"({6 * flops / us / 1e6 / (BF16_MFMA_PEAK / 1e12):.2f} of the bf16 matrix peak; {flops / us / 1e6 / (F32_MFMA_PEAK / 1e12):.2f} x the fp32 matrix peak)")
    print(f"  torch on the same GPU    : {us_t:8.1f} us  (5 library kernels, [N, D] activations through HBM)")
    print(f"  forward + backward       : {us_fb:8.1f} us  (moc_gated_attention_backward alone, with its two library GEMMs: {us_b:.1f} us); "
          f"torch autograd on the same GPU {us_fb_t:.1f} us")
    print(f"  torch on the host cores  : {cpu_us:8.1f} us  ({torch.get_num_threads()} threads)")


if __name__ == "__main__":
    main()
