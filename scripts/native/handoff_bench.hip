// Go / no-go measurement for "one launch per meta-step" (VERDICT round 1, item 6): is handing the forward's output
// to the step's 16 workgroups INSIDE one launch (arrival counter, write-through stores, one agent acquire --
// cdna_hip_programming.md Guideline 16) cheaper than the kernel boundary it replaces?
//
// The two shapes of one meta-step, with the real kernels' grids, dependent first touches and byte counts but no
// arithmetic (the question is about latency between the phases, which is what bounds the sequential step):
//   A  two launches:   forward-like  (96 WGs x 256: idx -> 16 gathered 1-KiB rows -> 128 B of scores + 4 KiB of
//                                      hidden rows each)  | boundary |
//                      step-like     (16 WGs x 1024: n_sel -> all 13 KiB of scores -> 20 gathered rows -> its 2-KiB
//                                      slice of the parameters, which the NEXT forward reads)  | boundary |
//   B  one launch:     the same 96 + 16 workgroups; the 16 wait on an arrival counter the 96 bump after their
//                      sc1 stores have drained; one agent-scope acquire, then plain loads.
// Prints microseconds per step for both over a chain of STEPS steps (each step depends on the previous one's
// parameter slice, as in training).
//
//   hipcc -O3 --offload-arch=gfx950 scripts/native/handoff_bench.hip -o /tmp/handoff_bench && /tmp/handoff_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <ctime>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int NP = 96, NC = 16, ROWS = 1600, D = 512, STEPS = 2000;

struct Buf {
    const unsigned short* X;     // [N][D] bf16
    const int* sel_row;          // [ROWS]
    const int* n_sel;            // [1]
    float* mixed;                // [2][ROWS]
    float* H1;                   // [ROWS][64]
    float* W;                    // [NC][512] "parameters"
    unsigned* arrive;            // [STEPS] arrival counters (B)
    unsigned* tmo;               // time-out word
    float* sink;
};

__device__ __forceinline__ float fwd_body(const Buf& b, int wg) {
    // dependent chain of the forward: n_sel -> sel_row -> bag rows; plus the parameters the previous step wrote
    const int n = *(volatile const int*)b.n_sel;
    const int r0 = wg * 16 + (threadIdx.x >> 4);
    float acc = 0.f;
    if (r0 < n) {
        const int row = b.sel_row[r0];
        const uint4* p = reinterpret_cast<const uint4*>(b.X + (size_t)row * D) + (threadIdx.x & 15) * 4;
        const uint4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
        acc = __uint_as_float(v0.x ^ v1.y ^ v2.z ^ v3.w) * 1e-30f;
    }
    acc += b.W[(wg % NC) * 512 + threadIdx.x] * 1e-3f;
    return acc;
}

__global__ __launch_bounds__(256) void forward_like(Buf b) {
    const float acc = fwd_body(b, blockIdx.x);
    const int r0 = blockIdx.x * 16 + (threadIdx.x >> 4);
    if ((threadIdx.x & 15) < 2) b.mixed[(threadIdx.x & 15) * ROWS + r0] = acc;
    for (int i = threadIdx.x; i < 16 * 64; i += 256) b.H1[(size_t)blockIdx.x * 16 * 64 + i] = acc;
}

__device__ __forceinline__ void step_body(const Buf& b, int wg, float* red) {
    const int n = *(volatile const int*)b.n_sel;
    float m = -1e30f;
    for (int i = threadIdx.x; i < 2 * ROWS; i += 1024) { const float v = b.mixed[i]; m = v > m ? v : m; }   // first touch of the scores
    for (int off = 32; off; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    float t = red[threadIdx.x & 15];
    // 20 "pairs": row id -> bag row + hidden row (two dependent hops)
    const int pr = ((int)(fabsf(t) * 1e3f) + threadIdx.x / 51) % (n > 0 ? n : 1);
    const int row = b.sel_row[pr];
    t += __uint_as_float(reinterpret_cast<const unsigned*>(b.X + (size_t)row * D)[threadIdx.x & 255]) * 1e-30f + b.H1[(size_t)pr * 64 + (threadIdx.x & 63)];
    if (threadIdx.x < 512) b.W[wg * 512 + threadIdx.x] = t * 1e-6f + 1.f;
}

__global__ __launch_bounds__(1024) void step_like(Buf b) {
    __shared__ float red[16];
    step_body(b, blockIdx.x, red);
}

typedef __attribute__((address_space(1))) unsigned gu32;
__global__ __launch_bounds__(1024) void fused_like(Buf b, int step) {
    __shared__ float red[16];
    if (blockIdx.x < NP) {
        if (threadIdx.x >= 256) return;
        const float acc = fwd_body(b, blockIdx.x);
        const int r0 = blockIdx.x * 16 + (threadIdx.x >> 4);
        // write-through (sc1) payload stores, every storing wave drains, ONE lane bumps the arrival counter
        if ((threadIdx.x & 15) < 2) __hip_atomic_store(b.mixed + (threadIdx.x & 15) * ROWS + r0, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = threadIdx.x; i < 16 * 64; i += 256)
            __hip_atomic_store(b.H1 + (size_t)blockIdx.x * 16 * 64 + i, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains, then signals for itself
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(b.arrive + step, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int wg = blockIdx.x - NP;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(b.arrive + step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(NP * 4)) {
            if (wall_clock64() - t0 > 200000000ull) { __hip_atomic_store(b.tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // 2 s: never hangs
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    step_body(b, wg, red);
}

int main() {
    const size_t N = 15000;
    unsigned short* X; int *sel, *nsel; float *mixed, *H1, *W, *sink; unsigned *arrive, *tmo;
    CHECK(hipMalloc(&X, N * D * 2)); CHECK(hipMemset(X, 0x3c, N * D * 2));
    CHECK(hipMalloc(&sel, ROWS * 4)); CHECK(hipMalloc(&nsel, 4));
    CHECK(hipMalloc(&mixed, 2 * ROWS * 4)); CHECK(hipMalloc(&H1, (size_t)ROWS * 64 * 4)); CHECK(hipMalloc(&W, NC * 512 * 4));
    CHECK(hipMalloc(&arrive, STEPS * 4)); CHECK(hipMalloc(&tmo, 16)); CHECK(hipMalloc(&sink, 16));
    std::vector<int> h(ROWS);
    for (int i = 0; i < ROWS; ++i) h[i] = (int)((i * 9301ull + 49297ull) % N);
    CHECK(hipMemcpy(sel, h.data(), ROWS * 4, hipMemcpyHostToDevice));
    int n = 1536;                                  // 96 row tiles
    CHECK(hipMemcpy(nsel, &n, 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(mixed, 0, 2 * ROWS * 4)); CHECK(hipMemset(H1, 0, (size_t)ROWS * 64 * 4)); CHECK(hipMemset(W, 0, NC * 512 * 4));
    CHECK(hipMemset(tmo, 0, 16));
    Buf b{X, sel, nsel, mixed, H1, W, arrive, tmo, sink};
    hipStream_t s; CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        float msA = 0, msB = 0;
        CHECK(hipEventRecord(e0, s));
        for (int t = 0; t < STEPS; ++t) { forward_like<<<NP, 256, 0, s>>>(b); step_like<<<NC, 1024, 0, s>>>(b); }
        CHECK(hipEventRecord(e1, s)); CHECK(hipStreamSynchronize(s)); CHECK(hipEventElapsedTime(&msA, e0, e1));
        CHECK(hipMemsetAsync(arrive, 0, STEPS * 4, s));
        CHECK(hipEventRecord(e0, s));
        for (int t = 0; t < STEPS; ++t) fused_like<<<NP + NC, 1024, 0, s>>>(b, t);
        CHECK(hipEventRecord(e1, s)); CHECK(hipStreamSynchronize(s)); CHECK(hipEventElapsedTime(&msB, e0, e1));
        unsigned to = 0; CHECK(hipMemcpy(&to, tmo, 4, hipMemcpyDeviceToHost));
        printf("rep %d: two launches %.2f us/step   one launch + arrival counter %.2f us/step   (time-outs: %u)\n", rep,
               msA * 1e3 / STEPS, msB * 1e3 / STEPS, to);
    }
    // C  the two launches of a 32-step pass captured ONCE in a hipGraph and replayed (the playbook's answer to a
    //    launch-bound inner loop): the same 64 dispatches per pass, one host call.  GPU time per step and the host
    //    time of issuing a pass both ways.
    {
        constexpr int PASS = 32;
        hipGraph_t graph; hipGraphExec_t exec;
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int t = 0; t < PASS; ++t) { forward_like<<<NP, 256, 0, s>>>(b); step_like<<<NC, 1024, 0, s>>>(b); }
        CHECK(hipStreamEndCapture(s, &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int w = 0; w < 5; ++w) CHECK(hipGraphLaunch(exec, s));
        CHECK(hipStreamSynchronize(s));
        const int passes = STEPS / PASS;
        for (int rep = 0; rep < 3; ++rep) {
            float msG = 0, msS = 0;
            timespec t0, t1;
            CHECK(hipEventRecord(e0, s));
            clock_gettime(CLOCK_MONOTONIC, &t0);
            for (int p = 0; p < passes; ++p) CHECK(hipGraphLaunch(exec, s));
            clock_gettime(CLOCK_MONOTONIC, &t1);
            CHECK(hipEventRecord(e1, s)); CHECK(hipStreamSynchronize(s)); CHECK(hipEventElapsedTime(&msG, e0, e1));
            const double hostG = ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / 1e3 / passes;
            CHECK(hipEventRecord(e0, s));
            clock_gettime(CLOCK_MONOTONIC, &t0);
            for (int p = 0; p < passes; ++p)
                for (int t = 0; t < PASS; ++t) { forward_like<<<NP, 256, 0, s>>>(b); step_like<<<NC, 1024, 0, s>>>(b); }
            clock_gettime(CLOCK_MONOTONIC, &t1);
            CHECK(hipEventRecord(e1, s)); CHECK(hipStreamSynchronize(s)); CHECK(hipEventElapsedTime(&msS, e0, e1));
            const double hostS = ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / 1e3 / passes;
            printf("rep %d: hipGraph of a 32-step pass %.2f us/step (host %.1f us per pass)   stream launches %.2f us/step (host %.1f us per pass)\n",
                   rep, msG * 1e3 / (passes * PASS), hostG, msS * 1e3 / (passes * PASS), hostS);
        }
    }
    return 0;
}
