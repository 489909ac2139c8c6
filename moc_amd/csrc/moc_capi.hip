// libmoc_hip.so: error plumbing, version and the composite entry points.
#include <stdarg.h>
#include <stdio.h>
#include "moc_common.h"

static thread_local char g_err[512] = "";

void moc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int moc_version(void) { return MOC_ABI_VERSION; }
extern "C" const char* moc_last_error(void) { return g_err; }

extern "C" int moc_phase_a(const moc_batch_t* B, const void* bank, moc_stream_t stream) {
    if (int rc = moc_mask_compact(B, stream)) return rc;
    if (int rc = moc_scores(B, bank, stream)) return rc;
    if (int rc = moc_select(B, stream)) return rc;
    return moc_gather_candidates(B, nullptr, stream);
}

#ifdef MOC_STAMPS
__device__ unsigned long long g_moc_stamps[128];
extern "C" int moc_debug_stamps(unsigned long long* host_out, int n) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_moc_stamps), sizeof(unsigned long long) * n);
}
extern "C" int moc_debug_stamps_set(const unsigned long long* host_in, int n) {      // (MOC_STAMP_MAX / _MIN slots start from here)
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_moc_stamps), host_in, sizeof(unsigned long long) * n);
}
#endif

// ---- which compute units are there (include/moc_hip.h: compute units kept free of the score pass) ---------------
// one wave per workgroup: where am I (XCC_ID, HW_ID's CU / SH / SE fields), then hold the slot for a while so that the
// launch spreads over every CU the queue may use
__global__ void __launch_bounds__(64) cu_census_kernel(int32_t* hist, long long hold_ticks) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) atomicAdd(&hist[(xcc & 15u) * 256u + ((hw >> 8) & 255u)], 1);
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(8);
}

extern "C" int moc_cu_census(int32_t* hist, int n_wg, int hold_us, moc_stream_t stream) {
    MOC_REQUIRE(hist != nullptr && n_wg >= 1 && n_wg <= (1 << 20) && hold_us >= 0 && hold_us <= 1000, "moc_cu_census: bad arguments");
    // wall_clock64 ticks at 100 MHz on this part
    hipLaunchKernelGGL(cu_census_kernel, dim3(n_wg), dim3(64), 0, (hipStream_t)stream, hist, (long long)hold_us * 100);
    MOC_CHECK_LAUNCH("cu_census_kernel");
    return MOC_OK;
}
