// Peer-to-peer gradient exchange: communicator life cycle and the stand-alone all-reduce
// (protocol and device functions: moc_p2p.h).
#include <stdlib.h>
#include <string.h>
#include "moc_p2p.h"

struct moc_p2p {
    int world, rank;
    int64_t n_par;
    size_t bytes;
    void* local;                               // flags | receive buffer (fine-grained)
    void* peer[MOC_P2P_MAX_WORLD];             // mapped peers ([rank] = local)
    bool opened[MOC_P2P_MAX_WORLD];
    int32_t* error;                            // host-pinned, device-visible
    uint32_t seq;
    unsigned long long timeout_ticks;
};

namespace {

constexpr size_t FLAG_BYTES = 4096;            // world * channels * 4 = 512 B, padded to a page
constexpr int STICKY_WORD = 1000;              // flags[1000]: "an exchange timed out" (never written by peers)

size_t comm_bytes(int world, int64_t n_par) { return FLAG_BYTES + sizeof(float) * 2 * (size_t)world * (size_t)n_par; }

// buf[0..n) <- sum over ranks, channel = workgroup: contiguous chunk per workgroup
__global__ __launch_bounds__(1024) void p2p_allreduce_kernel(P2pArgs x, float* buf, int64_t n) {
    __shared__ int ok_lds;
    const int64_t per = (n + MOC_P2P_CHANNELS - 1) / MOC_P2P_CHANNELS;
    const int64_t lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
    for (int64_t e = lo + threadIdx.x; e < hi; e += 1024) p2p_push(x, e, buf[e]);
    if (!p2p_signal_wait(x, blockIdx.x, &ok_lds)) return;
    for (int64_t e = lo + threadIdx.x; e < hi; e += 1024) buf[e] = p2p_sum(x, e, buf[e]);
}

}  // namespace

int moc_p2p_next_args(moc_p2p* c, P2pArgs* x) {
    MOC_REQUIRE(c && c->local, "moc_p2p: null communicator");
    for (int q = 0; q < c->world; ++q)
        MOC_REQUIRE(c->peer[q], "moc_p2p: rank %d is not connected (moc_p2p_connect)", q);
    memset(x, 0, sizeof(*x));
    x->world = c->world; x->rank = c->rank; x->n_par = c->n_par;
    x->seq = ++c->seq;                         // starts at 1 (the flags start at 0 = "nothing pushed yet"); wraps through 0
                                               // after 2^32 exchanges, which the signed-difference test in the kernel absorbs
                                               // and which keeps the buffer parity alternating
    x->flags = (uint32_t*)c->local;
    x->sticky = x->flags + STICKY_WORD;
    x->recv = (float*)((char*)c->local + FLAG_BYTES);
    for (int q = 0; q < c->world; ++q) {
        x->peer_flags[q] = (uint32_t*)c->peer[q];
        x->peer_recv[q] = (float*)((char*)c->peer[q] + FLAG_BYTES);
    }
    x->error = c->error;
    x->timeout_ticks = c->timeout_ticks;
    return MOC_OK;
}

extern "C" int moc_p2p_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

extern "C" int moc_p2p_create(int world, int rank, int64_t n_par, moc_p2p_t** out) {
    MOC_REQUIRE(out, "moc_p2p_create: null out");
    *out = nullptr;
    MOC_REQUIRE(world >= 1 && world <= MOC_P2P_MAX_WORLD && rank >= 0 && rank < world && n_par >= 1,
                "moc_p2p_create: bad world=%d rank=%d n_par=%lld (world <= %d)", world, rank, (long long)n_par,
                MOC_P2P_MAX_WORLD);
    moc_p2p* c = (moc_p2p*)calloc(1, sizeof(moc_p2p));
    MOC_REQUIRE(c, "moc_p2p_create: out of host memory");
    c->world = world; c->rank = rank; c->n_par = n_par; c->bytes = comm_bytes(world, n_par);
    c->timeout_ticks = 500000000ull;           // 5 s of the 100 MHz clock
    if (const char* ms = getenv("MOC_P2P_TIMEOUT_MS")) {
        const long v = atol(ms);
        if (v > 0) c->timeout_ticks = (unsigned long long)v * 100000ull;
    }
    hipError_t e = hipExtMallocWithFlags(&c->local, c->bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        const size_t want = c->bytes;
        free(c);
        MOC_FAIL(MOC_ELAUNCH, "moc_p2p_create: fine-grained allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
    }
    e = hipMemset(c->local, 0, c->bytes);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->error, sizeof(int32_t), hipHostMallocMapped);
    if (e != hipSuccess) {
        (void)hipFree(c->local);
        free(c);
        MOC_FAIL(MOC_ELAUNCH, "moc_p2p_create: %s", hipGetErrorString(e));
    }
    *c->error = 0;
    c->peer[rank] = c->local;
    (void)hipDeviceSynchronize();
    *out = c;
    return MOC_OK;
}

extern "C" int moc_p2p_export(moc_p2p_t* c, void* blob) {
    MOC_REQUIRE(c && blob, "moc_p2p_export: null argument");
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, c->local);
    if (e != hipSuccess) MOC_FAIL(MOC_ELAUNCH, "moc_p2p_export: hipIpcGetMemHandle: %s", hipGetErrorString(e));
    memcpy(blob, &h, sizeof(h));
    return MOC_OK;
}

extern "C" int moc_p2p_connect(moc_p2p_t* c, const void* blobs) {
    MOC_REQUIRE(c && blobs, "moc_p2p_connect: null argument");
    int dev = 0, ndev = 0;
    (void)hipGetDevice(&dev);
    (void)hipGetDeviceCount(&ndev);
    for (int d = 0; d < ndev; ++d) {           // best effort: the mapping below is what matters
        int can = 0;
        if (d != dev && hipDeviceCanAccessPeer(&can, dev, d) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(d, 0);
    }
    (void)hipGetLastError();
    for (int q = 0; q < c->world; ++q) {
        if (q == c->rank || c->peer[q]) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char*)blobs + (size_t)q * sizeof(h), sizeof(h));
        void* p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) MOC_FAIL(MOC_ELAUNCH, "moc_p2p_connect: hipIpcOpenMemHandle(rank %d): %s", q, hipGetErrorString(e));
        c->peer[q] = p;
        c->opened[q] = true;
    }
    return MOC_OK;
}

extern "C" int moc_p2p_error(moc_p2p_t* c) {
    if (!c || !c->error) return -1;
    return *(volatile int32_t*)c->error;
}

extern "C" int moc_p2p_destroy(moc_p2p_t* c) {
    if (!c) return MOC_OK;
    (void)hipDeviceSynchronize();
    for (int q = 0; q < c->world; ++q)
        if (c->opened[q] && c->peer[q]) (void)hipIpcCloseMemHandle(c->peer[q]);
    if (c->local) (void)hipFree(c->local);
    if (c->error) (void)hipHostFree(c->error);
    free(c);
    return MOC_OK;
}

extern "C" int moc_p2p_allreduce(moc_p2p_t* c, float* buf, int64_t n, moc_stream_t stream) {
    MOC_REQUIRE(c && buf && n >= 1 && n <= c->n_par, "moc_p2p_allreduce: bad buffer (n <= %lld)", c ? (long long)c->n_par : 0LL);
    if (c->world == 1) return MOC_OK;
    P2pArgs x;
    if (int rc = moc_p2p_next_args(c, &x)) return rc;
    p2p_allreduce_kernel<<<MOC_P2P_CHANNELS, 1024, 0, (hipStream_t)stream>>>(x, buf, n);
    MOC_CHECK_LAUNCH("moc_p2p_allreduce");
    return MOC_OK;
}
