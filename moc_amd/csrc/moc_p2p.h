// One-shot gradient exchange between the GPUs of one node (SURVEY.md section 8, row e).
//
// The data-parallel meta-step is latency bound: 132 KB of gradient per rank, needed by the very next
// kernel.  A ring collective pays a launch plus 2(G-1) hops; here every rank WRITES its gradient
// slice straight into each peer's receive buffer over xGMI (posted stores, one hop), raises a flag
// per peer and workgroup, waits for its own flags in LOCAL memory and sums the G slices in rank
// order -- inside the step kernel, so the reduced gradient never takes a launch of its own and every
// rank computes bit-identical parameters.
//
//   receive buffer  recv[parity][src rank][n_par]   parity = seq & 1 (a rank can be at most one step
//                                                   ahead of a peer: its step t+1 cannot finish
//                                                   without that peer's step-t+1 push)
//   flags           flag[src rank][channel]         last sequence number pushed; never reset
//
// Both live in ONE fine-grained (system-scope coherent) allocation per rank, exported with
// hipIpcGetMemHandle and mapped by the peers.  Every access to it is a system-scope atomic
// (sc0 sc1: no stale line in an XCD's L2), ordered by a system-scope release fence before the flag.
#pragma once
#include "moc_common.h"

#define MOC_P2P_MAX_WORLD 8
#define MOC_P2P_CHANNELS 16          // = workgroups of the step kernel (H / 4)

struct P2pArgs {
    int world, rank;                 // world <= 1: no exchange
    uint32_t seq;                    // sequence number of this exchange (>= 1), same on every rank
    int64_t n_par;                   // floats per slot
    float* recv;                     // local receive buffer
    uint32_t* flags;                 // local flags [world][MOC_P2P_CHANNELS]
    float* peer_recv[MOC_P2P_MAX_WORLD];      // [q]: rank q's receive buffer as mapped here ([rank] unused)
    uint32_t* peer_flags[MOC_P2P_MAX_WORLD];
    uint32_t* sticky;                // local word: non-zero once any exchange timed out (later waits bail out at once)
    int32_t* error;                  // host-pinned word: set to 1 + the silent rank on a time-out
    unsigned long long timeout_ticks;   // of the 100 MHz constant clock
};

__device__ __forceinline__ void p2p_push(const P2pArgs& x, int64_t e, float v) {
    const int64_t slot = ((int64_t)(x.seq & 1u) * x.world + x.rank) * x.n_par + e;
    for (int q = 0; q < x.world; ++q)
        if (q != x.rank) __hip_atomic_store(x.peer_recv[q] + slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// All threads of the workgroup call it after their pushes.  Returns false on a time-out (the same
// value in every thread).  `ok_lds` is one int of LDS.
__device__ __forceinline__ bool p2p_signal_wait(const P2pArgs& x, int channel, int* ok_lds) {
    __threadfence_system();          // this thread's pushes are visible system-wide ...
    if (threadIdx.x == 0) *ok_lds = 1;
    __syncthreads();                 // ... and so are everybody else's before any flag goes up
    const int q = threadIdx.x;
    if (q < x.world && q != x.rank) {
        __hip_atomic_store(x.peer_flags[q] + x.rank * MOC_P2P_CHANNELS + channel, x.seq, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
        const uint32_t* f = x.flags + q * MOC_P2P_CHANNELS + channel;
        const unsigned long long t0 = wall_clock64();
        while ((int32_t)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - x.seq) < 0) {
            // the exit every wave reaches: the time-out, or an earlier exchange's time-out
            if (wall_clock64() - t0 > x.timeout_ticks ||
                __hip_atomic_load(x.sticky, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                *ok_lds = 0;
                __hip_atomic_store(x.sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(x.error, 1 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    const bool ok = *ok_lds != 0;
    __threadfence_system();          // acquire side for the threads that did not poll
    return ok;
}

// sum over the ranks in rank order (the same order, hence the same bits, on every rank)
__device__ __forceinline__ float p2p_sum(const P2pArgs& x, int64_t e, float own) {
    const float* base = x.recv + (int64_t)(x.seq & 1u) * x.world * x.n_par + e;
    float v[MOC_P2P_MAX_WORLD];
#pragma unroll
    for (int q = 0; q < MOC_P2P_MAX_WORLD; ++q)
        v[q] = (q < x.world && q != x.rank)
                   ? __hip_atomic_load(base + (int64_t)q * x.n_par, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.f;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < MOC_P2P_MAX_WORLD; ++q)
        if (q < x.world) s += (q == x.rank) ? own : v[q];
    return s;
}

// host side (moc_p2p.hip)
struct moc_p2p;
int moc_p2p_next_args(struct moc_p2p* comm, P2pArgs* out);     // fills the kernel argument, advances seq
