// How fast does HBM serve a bag that is fetched in thin column slices?  (the K-split score kernel
// takes every 2-KiB row as 32 separate 64-byte pieces, one per 32-column chunk; the streaming kernels take 1 KiB of a row
// at a time.)  Read-only sweep of N rows x 2 KiB in the K-split order, PIECE bytes of each row per pass:
//   workgroup = 256 rows (4 waves x 4 row tiles x 16 rows), chunk loop over the row, PIECE bytes per row and chunk,
//   TWO chunks in flight (the ring kernel's look-ahead), two workgroups per CU.
//   hipcc -O3 --offload-arch=gfx950 scripts/native/piece_bench.hip -o /tmp/piece_bench && /tmp/piece_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int PIECE>   // bytes of a row per chunk: 64, 128, 256, 512, 1024
__global__ __launch_bounds__(256, 2) void sweep(const unsigned char* X, int64_t n_rows, int row_bytes, unsigned* sink) {
    constexpr int LPR = PIECE / 16;                 // lanes per row
    constexpr int RPI = 64 / LPR;                   // rows per wave-instruction
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * 256 + wave * 64;
    unsigned acc = 0;
    const int nchunk = row_bytes / PIECE;
    for (int c = 0; c < nchunk; ++c) {
        // the wave's 64 rows, RPI rows per instruction
#pragma unroll
        for (int i = 0; i < 64 / RPI; ++i) {
            int64_t r = row0 + i * RPI + lane / LPR;
            r = r < n_rows ? r : n_rows - 1;
            const uint4 v = *reinterpret_cast<const uint4*>(X + r * row_bytes + (int64_t)c * PIECE + (lane % LPR) * 16);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main() {
    const int64_t N = 1600000;                      // 3.28 GB of 2-KiB rows
    const int RB = 2048;
    unsigned char* X; unsigned* sink;
    CHECK(hipMalloc(&X, N * RB)); CHECK(hipMemset(X, 1, N * RB)); CHECK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = (int)((N + 255) / 256);
    auto run = [&](auto kern, int piece) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0));
            kern<<<grid, 256>>>(X, N, RB, sink);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("piece %4d B per row and chunk: %7.1f us  %5.2f TB/s\n", piece, best * 1e3, N * RB / (best * 1e-3) / 1e12);
    };
    run(sweep<64>, 64); run(sweep<128>, 128); run(sweep<256>, 256); run(sweep<512>, 512); run(sweep<1024>, 1024);
    return 0;
}
