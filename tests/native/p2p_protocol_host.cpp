// Host harness for the node-local exchange protocol: compiles moc_amd/csrc/moc_p2p_proto.h -- the code the step
// kernels run -- against GCC atomics and threads.  A "rank" is a set of channels (workgroups); a workgroup is `world`
// threads (the lanes that raise and poll flags; the other lanes of the real workgroup only pass the barriers).
//
//   p2p_protocol_host <world> <steps> <mode>
//     mode "late":   every rank takes random pauses; rank world-1 is late by 30 ms at every third step.  Every step,
//                    every rank, every element: the sum must be the sum of that step's pushes of all ranks (two
//                    parities alternate; a fast rank must never overwrite a slot a slow one still reads).
//     mode "silent": at step 3 the last rank stops for good.  The others must time out (bounded wait), report
//                    1 + that rank in their error word, skip the update, and return at once from every later wait.
#include <atomic>
#include <barrier>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#define MOC_P2P_CHANNELS 2
static thread_local int tl_lane;
static thread_local std::barrier<>* tl_barrier;
static inline unsigned long long now_ticks() {      // 100 MHz, like the device's constant clock
    return (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(
               std::chrono::steady_clock::now().time_since_epoch()).count() / 10ull;
}
#define P2P_FN static inline
#define P2P_ST_F_RELAXED_SYS(p, v) do { float v__ = (v); __atomic_store((p), &v__, __ATOMIC_RELAXED); } while (0)
static inline float ld_f(const float* p) { float v; __atomic_load(p, &v, __ATOMIC_RELAXED); return v; }
#define P2P_LD_F_RELAXED_SYS(p) ld_f(p)
#define P2P_ST_U_RELEASE_SYS(p, v) __atomic_store_n((p), (v), __ATOMIC_RELEASE)
#define P2P_LD_U_ACQUIRE_SYS(p) __atomic_load_n((p), __ATOMIC_ACQUIRE)
#define P2P_ST_U_RELAXED_DEV(p, v) __atomic_store_n((p), (v), __ATOMIC_RELAXED)
#define P2P_LD_U_RELAXED_DEV(p) __atomic_load_n((p), __ATOMIC_RELAXED)
#define P2P_ST_I_RELAXED_SYS(p, v) __atomic_store_n((p), (v), __ATOMIC_RELAXED)
#define P2P_FENCE_SYS() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define P2P_BARRIER() tl_barrier->arrive_and_wait()
#define P2P_LANE() tl_lane
#define P2P_CLOCK() now_ticks()
#define P2P_PAUSE() std::this_thread::yield()
#define P2P_LDS_FLAG_CLEAR(p) __atomic_store_n((p), 0, __ATOMIC_RELAXED)   // same-value stores from several threads
#include "moc_p2p_proto.h"

static const int64_t N_PAR = 96;                    // elements per rank and step (split over the channels)
static float value(int rank, int step, int64_t e) { return (float)((rank + 1) * 1000 + step * 7 + (int)(e % 13)); }

struct Rank {
    std::vector<float> recv;                        // [2][world][N_PAR]
    std::vector<uint32_t> flags;                    // [world][CHANNELS] + sticky word
    int32_t error = 0;
};

int main(int argc, char** argv) {
    const int world = argc > 1 ? atoi(argv[1]) : 4, steps = argc > 2 ? atoi(argv[2]) : 40;
    const bool silent = argc > 3 && !strcmp(argv[3], "silent");
    std::vector<Rank> R(world);
    for (auto& r : R) { r.recv.assign(2 * (size_t)world * N_PAR, -1.f); r.flags.assign((size_t)world * MOC_P2P_CHANNELS + 1, 0u); }
    std::atomic<long> bad{0}, timeouts{0}, fast_returns{0}, updates{0};
    std::vector<std::thread> th;
    for (int rank = 0; rank < world; ++rank)
        for (int ch = 0; ch < MOC_P2P_CHANNELS; ++ch) {
            auto* bar = new std::barrier<>(world);
            auto* ok_lds = new int(1);
            for (int lane = 0; lane < world; ++lane)
                th.emplace_back([=, &R, &bad, &timeouts, &fast_returns, &updates] {
                    tl_lane = lane; tl_barrier = bar;
                    std::mt19937 rng(rank * 131 + ch * 17 + lane);
                    const int64_t per = N_PAR / MOC_P2P_CHANNELS, lo = per * ch, hi = lo + per;
                    bool timed_out_before = false;
                    for (int step = 1; step <= steps; ++step) {      // (the host side hands out seq = 1, 2, ...)
                        if (silent && rank == world - 1 && step >= 3) return;     // (all lanes of the rank: no barrier is left waiting)
                        if (!silent) {
                            if (rng() % 4 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 300));
                            if (rank == world - 1 && step % 3 == 0) std::this_thread::sleep_for(std::chrono::milliseconds(30));
                        }
                        P2pArgs x{};
                        x.world = world; x.rank = rank; x.seq = (uint32_t)step; x.n_par = N_PAR;
                        x.recv = R[rank].recv.data(); x.flags = R[rank].flags.data();
                        x.sticky = R[rank].flags.data() + (size_t)world * MOC_P2P_CHANNELS;
                        x.error = &R[rank].error;
                        x.timeout_ticks = silent ? 20000000ull : 2000000000ull;      // 0.2 s / 20 s
                        for (int q = 0; q < world; ++q) { x.peer_recv[q] = R[q].recv.data(); x.peer_flags[q] = R[q].flags.data(); }
                        for (int64_t e = lo + lane; e < hi; e += world) p2p_push(x, e, value(rank, step, e));
                        const unsigned long long t0 = now_ticks();
                        const bool ok = p2p_signal_wait(x, ch, ok_lds);
                        if (!ok) {
                            if (lane == 0) { ++timeouts; if (timed_out_before && now_ticks() - t0 < 1000000ull) ++fast_returns; }
                            timed_out_before = true;
                        } else {
                            for (int64_t e = lo + lane; e < hi; e += world) {
                                float want = 0.f;
                                for (int q = 0; q < world; ++q) want += value(q, step, e);
                                if (p2p_sum(x, e, value(rank, step, e)) != want) ++bad;
                            }
                            if (lane == 0) ++updates;
                        }
                        bar->arrive_and_wait();       // the kernel boundary: one exchange per launch (ok_lds is per launch)
                    }
                });
        }
    for (auto& t : th) t.join();
    long err_words = 0;
    for (int r = 0; r < world - 1; ++r) err_words += (R[r].error == world) ? 1 : 0;   // 1 + silent rank (= world - 1)
    printf("world=%d steps=%d mode=%s bad=%ld updates=%ld timeouts=%ld fast_returns=%ld error_words_naming_silent_rank=%ld\n", world, steps,
           silent ? "silent" : "late", bad.load(), updates.load(), timeouts.load(), fast_returns.load(), err_words);
    return bad.load() ? 1 : 0;
}
