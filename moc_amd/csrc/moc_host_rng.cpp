// Host-only part of libmoc_hip.so (plain C++, built by g++ so the two hot loops can be
// multi-versioned for AVX2).
#include <stdint.h>
#include "../../include/moc_hip.h"

void moc_set_error(const char* fmt, ...);

// ---- host-side row masks: the reference's own random stream, faster -----------------------
// main_moc.py:330 draws `torch.rand(N) > 0.5` on the CPU default generator: mt19937, one 32-bit
// output per float32 sample, sample = (y & 0xFFFFFF) * 2^-24 (ATen uniform_real_distribution),
// so the mask bit is (y & 0xFFFFFF) > 2^23.  This regenerates exactly those bits from the
// generator state torch hands out (torch.get_rng_state(): legacy pod {u64 seed; i32 left;
// i32 seeded; u64 next; u64 state[624]; ...}) and leaves the state where torch.rand would,
// so any later consumer of the generator sees the same stream.  Block regeneration is written
// so the compiler can vectorise it (the three sub-loops have dependence distance >= 227).
namespace {
constexpr int MT_N = 624, MT_M = 397;

inline uint32_t mt_twist(uint32_t u, uint32_t v) {
    return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}

__attribute__((target_clones("avx2", "default"))) void mt_next_state(uint32_t* __restrict__ p) {
    uint32_t q[MT_N + 1];
    for (int i = 0; i < MT_N; ++i) q[i] = p[i];
    q[MT_N] = 0;
    // out-of-place first pass keeps every read on OLD values, exactly as the in-place recurrence
    for (int i = 0; i < MT_N - MT_M; ++i) p[i] = q[i + MT_M] ^ mt_twist(q[i], q[i + 1]);
    // p[i] = NEW p[i - (N-M)] ^ twist(old p[i], old p[i+1]); distance 227 -> chunks are independent
    for (int i = MT_N - MT_M; i < MT_N - 1; ++i) p[i] = p[i - (MT_N - MT_M)] ^ mt_twist(q[i], q[i + 1]);
    p[MT_N - 1] = p[MT_M - 1] ^ mt_twist(q[MT_N - 1], p[0]);
}

__attribute__((target_clones("avx2", "default"))) int64_t mt_emit(const uint32_t* __restrict__ st, int n,
                                                                   uint8_t* __restrict__ out) {
    int64_t kept = 0;
    for (int i = 0; i < n; ++i) {
        uint32_t y = st[i];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        const uint8_t m = (y & 0xFFFFFFu) > 0x800000u;
        out[i] = m;
        kept += m;
    }
    return kept;
}
}  // namespace

extern "C" int64_t moc_host_draw_masks(uint8_t* rng_state, int64_t state_bytes, int64_t n, uint8_t* out) {
    if (!rng_state || !out || n < 0 || state_bytes < (int64_t)(24 + MT_N * 8)) {
        moc_set_error("moc_host_draw_masks: bad arguments (state_bytes=%lld, n=%lld)", (long long)state_bytes, (long long)n);
        return -1;
    }
    int32_t* left = reinterpret_cast<int32_t*>(rng_state + 8);
    const int32_t seeded = *reinterpret_cast<int32_t*>(rng_state + 12);
    uint64_t* next = reinterpret_cast<uint64_t*>(rng_state + 16);
    uint64_t* st64 = reinterpret_cast<uint64_t*>(rng_state + 24);
    if (!seeded || *left < 1 || *left > MT_N || *next > (uint64_t)MT_N) {
        moc_set_error("moc_host_draw_masks: unexpected generator state (left=%d next=%llu seeded=%d)", *left,
                      (unsigned long long)*next, seeded);
        return -1;
    }
    uint32_t st[MT_N];
    for (int i = 0; i < MT_N; ++i) st[i] = (uint32_t)st64[i];
    int l = *left;
    int nx = (int)*next;
    int64_t kept = 0, done = 0;
    // at::mt19937::operator(): if (--left == 0) next_state(); y = state[next++]
    while (done < n) {
        if (l == 1) {          // the next draw would hit --left == 0
            mt_next_state(st);
            l = MT_N + 1;      // next_state sets left = N; the pending --left brings it to N for this draw
            nx = 0;
        }
        // draws available before the next regeneration: each consumes one unit of left until left == 1
        int avail = l - 1;
        if (avail > n - done) avail = (int)(n - done);
        kept += mt_emit(st + nx, avail, out + done);
        nx += avail;
        l -= avail;
        done += avail;
    }
    for (int i = 0; i < MT_N; ++i) st64[i] = st[i];
    *left = l;
    *next = (uint64_t)nx;
    return kept;
}

// The largest number of kept rows of any slide, from the keep flags the GPU is about to read: `max_rows` of the batch
// (moc_hip.h) only has to bound the rows a slide brings to the selectors, and with a row mask that is the kept rows --
// about half.  The bound decides kernel shapes (a 30-way union of ~7,500 kept rows pools inside the step kernel,
// <= 8,192; bounded by the 15,000 rows of the bag it would take the separate top-K launch every step).
extern "C" int64_t moc_host_max_kept(const uint8_t* mask, const int64_t* row_off_host, int n_slides) {
    if (!mask || !row_off_host || n_slides < 1) {
        moc_set_error("moc_host_max_kept: bad arguments");
        return -1;
    }
    int64_t best = 0;
    for (int b = 0; b < n_slides; ++b) {
        const uint8_t* m = mask + row_off_host[b];
        const int64_t n = row_off_host[b + 1] - row_off_host[b];
        int64_t k = 0;
        for (int64_t i = 0; i < n; ++i) k += m[i] != 0;
        best = k > best ? k : best;
    }
    return best;
}
