// The exchange protocol of moc_p2p.h, free of any HIP name: the includer supplies the memory and execution
// primitives as macros, so that the SAME code is compiled into the step kernels (moc_p2p.h: HIP atomics, fences,
// __syncthreads) and into a host harness (tests/native/p2p_protocol_host.cpp: GCC atomics, one thread per lane,
// std::barrier) that runs ranks with random delays, a deliberately late rank and a silent one on the CPU.
//
//   P2P_FN                       function qualifiers
//   P2P_ST_F_RELAXED_SYS(p, v)   float store, system scope        P2P_LD_F_RELAXED_SYS(p)
//   P2P_ST_U_RELEASE_SYS(p, v)   uint32 flag store (release)      P2P_LD_U_ACQUIRE_SYS(p)
//   P2P_ST_U_RELAXED_DEV(p, v)   uint32, device scope (sticky)    P2P_LD_U_RELAXED_DEV(p)
//   P2P_ST_I_RELAXED_SYS(p, v)   int32 (host-visible error word)
//   P2P_FENCE_SYS()  P2P_BARRIER()  P2P_LANE()  P2P_CLOCK()  P2P_PAUSE()
#pragma once
#include <stdint.h>

#define MOC_P2P_MAX_WORLD 8
#ifndef MOC_P2P_CHANNELS
#define MOC_P2P_CHANNELS 16          // = workgroups of the step kernel (H / 4)
#endif

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

P2P_FN void p2p_push(const P2pArgs& x, int64_t e, float v) {
    const int64_t slot = ((int64_t)(x.seq & 1u) * x.world + x.rank) * x.n_par + e;
    for (int q = 0; q < x.world; ++q)
        if (q != x.rank) P2P_ST_F_RELAXED_SYS(x.peer_recv[q] + slot, v);
}

// All threads of the workgroup call it after their pushes.  Returns false on a time-out (the same
// value in every thread).  `ok_lds` is one int of workgroup-shared memory.
P2P_FN bool p2p_signal_wait(const P2pArgs& x, int channel, int* ok_lds) {
    P2P_FENCE_SYS();                 // this thread's pushes are visible system-wide ...
    if (P2P_LANE() == 0) *ok_lds = 1;
    P2P_BARRIER();                   // ... and so are everybody else's before any flag goes up
    const int q = P2P_LANE();
    if (q < x.world && q != x.rank) {
        P2P_ST_U_RELEASE_SYS(x.peer_flags[q] + x.rank * MOC_P2P_CHANNELS + channel, x.seq);
        const uint32_t* f = x.flags + q * MOC_P2P_CHANNELS + channel;
        const unsigned long long t0 = P2P_CLOCK();
        while ((int32_t)(P2P_LD_U_ACQUIRE_SYS(f) - x.seq) < 0) {
            // the exit every wave reaches: the time-out, or an earlier exchange's time-out
            const bool timed_out = P2P_CLOCK() - t0 > x.timeout_ticks;
            if (timed_out || P2P_LD_U_RELAXED_DEV(x.sticky) != 0u) {
                P2P_LDS_FLAG_CLEAR(ok_lds);     // (several pollers may clear it at once: the same value)
                if (timed_out) {         // (a bail-out on the sticky word keeps the error word of the exchange that timed out:
                    P2P_ST_U_RELAXED_DEV(x.sticky, 1u);             // the peer polled NOW may merely be late itself)
                    P2P_ST_I_RELAXED_SYS(x.error, 1 + q);
                }
                break;
            }
            P2P_PAUSE();
        }
    }
    P2P_BARRIER();
    const bool ok = *ok_lds != 0;
    P2P_FENCE_SYS();                 // acquire side for the threads that did not poll
    return ok;
}

// sum over the ranks in rank order (the same order, hence the same bits, on every rank)
P2P_FN float p2p_sum(const P2pArgs& x, int64_t e, float own) {
    const float* base = x.recv + (int64_t)(x.seq & 1u) * x.world * x.n_par + e;
    float v[MOC_P2P_MAX_WORLD];
#pragma unroll
    for (int q = 0; q < MOC_P2P_MAX_WORLD; ++q)
        v[q] = (q < x.world && q != x.rank) ? P2P_LD_F_RELAXED_SYS(base + (int64_t)q * x.n_par) : 0.f;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < MOC_P2P_MAX_WORLD; ++q)
        if (q < x.world) s += (q == x.rank) ? own : v[q];
    return s;
}
