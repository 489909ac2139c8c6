// What does HBM deliver when a read stream carries writes with it?  (the score pass reads 1 KiB per
// bag row and writes (2C + 3) floats of statistics -- 3 % of the bytes at C = 2, 25 % at C = 30 -- and its rate falls from
// 6.1 TB/s with nothing stored to 3.3-3.6 TB/s of reads at C = 30.)  The plainest kernel there is: every thread streams 16-byte loads
// over SRC, and for every EVERY-th load stores 16 bytes to DST -- fully coalesced both ways, nothing else.
//   hipcc -O3 --offload-arch=gfx950 scripts/native/mix_bench.hip -o /tmp/mix_bench && /tmp/mix_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int EVERY>   // one 16-byte store per EVERY 16-byte loads (0: none)
__global__ __launch_bounds__(256, 2) void mix(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n, unsigned* sink) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    unsigned acc = 0;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    constexpr int UN = 8;
    for (; i + (UN - 1) * stride < n; i += UN * stride) {
        uint4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
            if (EVERY > 0 && u % (EVERY > UN ? UN : EVERY) == 0 && (EVERY <= UN || (i / (UN * stride)) % (EVERY / UN) == 0))
                dst[(i + u * stride) / EVERY] = v[u];
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <int EVERY>
static void run(const uint4* src, uint4* dst, int64_t n, unsigned* sink) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) mix<EVERY><<<2048, 256>>>(src, dst, n, sink);
    float best = 1e30f, sum = 0.f;
    const int reps = 10;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(e0));
        mix<EVERY><<<2048, 256>>>(src, dst, n, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; sum += ms;
    }
    const double rd = (double)n * 16, wr = EVERY ? rd / EVERY : 0.0, t = sum / reps * 1e-3;
    printf("one 16-B store per %2d loads (writes = %4.1f %% of the bytes): %7.1f us  reads %5.2f TB/s  reads + writes %5.2f TB/s  (best %.1f us)\n",
           EVERY, 100.0 * wr / (rd + wr), t * 1e6, rd / t / 1e12, (rd + wr) / t / 1e12, best * 1e3);
}

int main() {
    const int64_t bytes = 3ll << 30;                 // 3 GiB read stream
    const int64_t n = bytes / 16;
    uint4 *src, *dst; unsigned* sink;
    CHECK(hipMalloc(&src, bytes)); CHECK(hipMalloc(&dst, bytes)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(src, 1, bytes)); CHECK(hipMemset(dst, 0, bytes));
    printf("scripts/native/mix_bench.hip on one MI355X: 3 GiB coalesced read stream, 16-byte stores mixed in\n");
    run<0>(src, dst, n, sink);
    run<32>(src, dst, n, sink);
    run<8>(src, dst, n, sink);
    run<4>(src, dst, n, sink);
    run<2>(src, dst, n, sink);
    run<1>(src, dst, n, sink);
    return 0;
}
