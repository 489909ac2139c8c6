// Diagnostic: can a small kernel (G workgroups x T threads, ~V VGPRs, L bytes of LDS) start while the
// persistent score kernel holds every CU?  Built by scripts/diag_coresidency.py with hipcc, loaded with ctypes.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int NV, int THREADS>
__global__ __launch_bounds__(THREADS) void spin_kernel(unsigned long long ticks, float* out, unsigned long long* stamps) {
    extern __shared__ float lds[];
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = threadIdx.x * 0.001f + i;
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = t0;
    lds[threadIdx.x] = v[0];
    __syncthreads();
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = v[i] * 1.0001f + lds[(threadIdx.x + i) & 255];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i];
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[1] = wall_clock64();
}

extern "C" int coresidency_launch(int grid, int threads, int nv, int lds_bytes, unsigned long long ticks, float* out,
                                  unsigned long long* stamps, void* stream) {
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(NV, TH) do { hipFuncSetAttribute((const void*)spin_kernel<NV, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        spin_kernel<NV, TH><<<grid, threads, lds_bytes, s>>>(ticks, out, stamps); } while (0)
    if (threads == 1024) { if (nv <= 24) LAUNCH(24, 1024); else LAUNCH(80, 1024); }          // 80 -> ~112 VGPRs, no scratch
    else if (threads == 512) { if (nv <= 48) LAUNCH(48, 512); else LAUNCH(80, 512); }
    else { if (nv <= 48) LAUNCH(48, 256); else if (nv <= 100) LAUNCH(100, 256); else LAUNCH(220, 256); }
    return (int)hipGetLastError();
}
