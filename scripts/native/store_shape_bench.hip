// Which property of the score pass's statistics stores costs what?    A persistent grid of 1,024 waves walks
// 16-KiB tiles in the score kernel's strided order (tile = wave + k * 1024), reads each with sixteen coalesced 16-byte loads
// per lane and then writes 4 KiB per tile (25 % of the read bytes, the ratio at 30 classes) in one of these shapes:
//   0  nothing
//   1  the kernel's: 16 store instructions of 4 bytes per lane, each to FOUR statistics rows x 16 slots (64-byte pieces, 64 rows)
//   2  256-byte pieces: 16 instructions of 4 bytes per lane, each ONE statistics row x 64 slots (a wave fills a row's 64 slots
//      from four tiles' worth of results: what buffering four tiles would give)
//   3  1-KiB pieces: 4 instructions of 16 bytes per lane, one statistics row x 256 slots each
//   4  tile-major: the tile's 4 KiB contiguous (16 instructions of 4 bytes per lane)
//   hipcc -O3 --offload-arch=gfx950 scripts/native/store_shape_bench.hip -o /tmp/ssb && /tmp/ssb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void walk(const uint4* __restrict__ src, float* __restrict__ dst, int64_t n_tiles, int64_t row_stride, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nw = (int64_t)gridDim.x * 4;
    unsigned acc = 0;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < n_tiles; t += nw) {
        const uint4* p = src + t * 1024 + lane;                 // 16 KiB per tile
        uint4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = p[u * 64];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        const float x = __uint_as_float(acc | 1u);
        if (SHAPE == 1) {
#pragma unroll
            for (int s = 0; s < 16; ++s) dst[(int64_t)(s * 4 + (lane >> 4)) * row_stride + t * 16 + (lane & 15)] = x;
        } else if (SHAPE == 2) {
#pragma unroll
            for (int s = 0; s < 16; ++s) dst[(int64_t)(s + 16 * (t & 3)) * row_stride + (t >> 2) * 64 + lane] = x;
        } else if (SHAPE == 3) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
                reinterpret_cast<float4*>(dst + (int64_t)(s + 4 * (t & 15)) * row_stride + (t >> 4) * 256)[lane] = float4{x, x, x, x};
        } else if (SHAPE == 4) {
#pragma unroll
            for (int s = 0; s < 16; ++s) dst[t * 1024 + s * 64 + lane] = x;
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <int SHAPE>
static void run(const char* what, const uint4* src, float* dst, int64_t n_tiles, int64_t row_stride, unsigned* sink) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) walk<SHAPE><<<256, 256>>>(src, dst, n_tiles, row_stride, sink);
    float sum = 0.f, best = 1e30f;
    const int reps = 10;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(e0));
        walk<SHAPE><<<256, 256>>>(src, dst, n_tiles, row_stride, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        sum += ms; best = ms < best ? ms : best;
    }
    const double rd = (double)n_tiles * 16384, t = sum / reps * 1e-3;
    printf("%-58s %7.1f us  reads %5.2f TB/s  (best %.1f us)\n", what, t * 1e6, rd / t / 1e12, best * 1e3);
}

int main() {
    const int64_t n_tiles = 112500;                  // 1.8 M rows x 1 KiB
    const int64_t row_stride = n_tiles * 16;         // slots per statistics row
    uint4* src; float* dst; unsigned* sink;
    CHECK(hipMalloc(&src, n_tiles * 16384)); CHECK(hipMalloc(&dst, (size_t)64 * row_stride * 4 + (1 << 20))); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(src, 1, n_tiles * 16384)); CHECK(hipMemset(dst, 0, (size_t)64 * row_stride * 4));
    printf("scripts/native/store_shape_bench.hip on one MI355X: 1,024 waves (one per SIMD), 1.84 GB read in 16-KiB tiles, 4 KiB written per tile\n");
    run<0>("nothing stored", src, dst, n_tiles, row_stride, sink);
    run<1>("64-byte pieces into 64 statistics rows (the kernel's)", src, dst, n_tiles, row_stride, sink);
    run<2>("256-byte pieces (one row x 64 slots per instruction)", src, dst, n_tiles, row_stride, sink);
    run<3>("1-KiB pieces (16 bytes per lane, one row x 256 slots)", src, dst, n_tiles, row_stride, sink);
    run<4>("tile-major (4 KiB contiguous per tile)", src, dst, n_tiles, row_stride, sink);
    return 0;
}
