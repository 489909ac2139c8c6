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
#endif
