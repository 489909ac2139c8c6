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

// the protocol itself (P2pArgs, p2p_push, p2p_signal_wait, p2p_sum) is in moc_p2p_proto.h, written against these
// primitives so that tests/native/p2p_protocol_host.cpp can run the very same code on the CPU
#define P2P_FN __device__ __forceinline__
#define P2P_ST_F_RELAXED_SYS(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#define P2P_LD_F_RELAXED_SYS(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#define P2P_ST_U_RELEASE_SYS(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM)
#define P2P_LD_U_ACQUIRE_SYS(p) __hip_atomic_load((p), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)
#define P2P_ST_U_RELAXED_DEV(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define P2P_LD_U_RELAXED_DEV(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define P2P_ST_I_RELAXED_SYS(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#define P2P_FENCE_SYS() __threadfence_system()
#define P2P_BARRIER() __syncthreads()
#define P2P_LANE() ((int)threadIdx.x)
#define P2P_CLOCK() wall_clock64()
#define P2P_PAUSE() __builtin_amdgcn_s_sleep(2)
#define P2P_LDS_FLAG_CLEAR(p) (*(p) = 0)
#include "moc_p2p_proto.h"

// host side (moc_p2p.hip)
struct moc_p2p;
int moc_p2p_next_args(struct moc_p2p* comm, P2pArgs* out);     // fills the kernel argument, advances seq
