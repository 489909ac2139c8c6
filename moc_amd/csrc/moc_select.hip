// Phase A, part 2: the four top-j patch selectors as column-wise radix selects,
// their union, the ascending compaction of the union with the candidate scores,
// and the generic top-K-mean used for pooling.
//
// Reference semantics:
//   psi_p / psi_sigma / psi_delta / psi_beta ... utils/patch_selection_classifier_index.py:17-87
//   union, sorted ............................. main_moc.py:341-354
//   candidate scores .......................... main_moc.py:359-366
//   top-K mean pooling ........................ utils/patch_selection_classifier.py:18-32
//
// As *sets* the selectors are 2C+2 independent "j best rows of one key column"
// problems (psi_delta's C columns are identical, psi_beta is the j smallest
// background sums: its second topk only permutes them).  Each is one workgroup:
// a 4-pass 8-bit radix select over order-preserving uint32 keys finds the j-th
// key exactly; rows above it are marked, ties at it are taken in ascending row
// order.  Marks from all columns land in one byte-per-row flag array = the union.
#include "moc_common.h"

int moc_check_batch(const moc_batch_t* B, const char* who);

namespace {

// ---- block-wide radix select -------------------------------------------------
// Finds T = the `want`-th largest key among n keys produced by keyfn(i), i in [0,n).
// Returns T and *n_tie_take = how many keys equal to T belong to the top `want`,
// *n_tie_all = how many keys equal T.  1 <= want <= n.  All threads must call.
// hist: LDS int[256 + 4].
template <typename KeyFn>
__device__ __forceinline__ uint32_t block_radix_select(KeyFn keyfn, int n, int want, int* hist,
                                                       int* n_tie_take, int* n_tie_all) {
    uint32_t prefix = 0, pmask = 0;
    int remaining = want;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const uint32_t u = keyfn(i);
            if ((u & pmask) == prefix) atomicAdd(&hist[(u >> shift) & 255], 1);
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            // lane L owns digits 4L..4L+3; inclusive suffix sums over lanes via shuffles
            const int L = threadIdx.x;
            const int h0 = hist[4 * L], h1 = hist[4 * L + 1], h2 = hist[4 * L + 2], h3 = hist[4 * L + 3];
            int suf = h0 + h1 + h2 + h3;   // becomes sum over lanes >= L
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_down(suf, off, 64);
                if (L + off < 64) suf += o;
            }
            const int above = suf - (h0 + h1 + h2 + h3);   // keys in digits > 4L+3
            // the digit d holding the `remaining`-th key: above(d) < remaining <= above(d) + hist[d]
            int a3 = above, a2 = above + h3, a1 = a2 + h2, a0 = a1 + h1;
            int d = -1, ab = 0, hd = 0;
            if (a3 < remaining && remaining <= a3 + h3) { d = 4 * L + 3; ab = a3; hd = h3; }
            else if (a2 < remaining && remaining <= a2 + h2) { d = 4 * L + 2; ab = a2; hd = h2; }
            else if (a1 < remaining && remaining <= a1 + h1) { d = 4 * L + 1; ab = a1; hd = h1; }
            else if (a0 < remaining && remaining <= a0 + h0) { d = 4 * L; ab = a0; hd = h0; }
            if (d >= 0) { hist[256] = d; hist[257] = ab; hist[258] = hd; }
        }
        __syncthreads();
        const int d = hist[256];
        remaining -= hist[257];
        *n_tie_all = hist[258];
        prefix |= (uint32_t)d << shift;
        pmask |= 255u << shift;
        __syncthreads();
    }
    *n_tie_take = remaining;
    return prefix;
}

constexpr int SEL_CAND = 4096;     // candidate list of select_kernel's two-sweep path

struct SelectArgs {
    const float* stats;
    const int64_t* row_off;
    const int32_t* n_kept;   // nullable (no mask)
    uint8_t* sel_flag;
    int64_t stride;
    int C, topj;
    uint32_t discard_bits;
    int compact;             // MOC_STATS_COMPACT layout: logits[C] | m1 | 1/den | gap | bg_sum | bg_max
};


// order-preserving key of selector column kc (0..C-1 psi_p, C..2C-1 psi_sigma, 2C psi_delta, 2C+1 psi_beta: smallest
// background sum first) at slot i of a slide; `s` = stats + first slot of the slide
__device__ __forceinline__ uint32_t sel_key(const SelectArgs& a, const float* s, int kc, int i) {
    const int C = a.C;
    const int64_t st = a.stride;
    if (!a.compact) {
        const uint32_t u = moc_key_desc(s[(int64_t)kc * st + i]);
        return kc == 2 * C + 1 ? ~u : u;
    }
    if (kc < C) return moc_key_desc(s[(int64_t)kc * st + i]);
    if (kc < 2 * C) return moc_key_desc(moc_softmax_from(s[(int64_t)(kc - C) * st + i], s[(int64_t)C * st + i], s[(int64_t)(C + 1) * st + i]));
    if (kc == 2 * C) return moc_key_desc(s[(int64_t)(C + 2) * st + i]);
    return ~moc_key_desc(s[(int64_t)(C + 3) * st + i]);
}

// grid (2C+2, n_slides)
__global__ __launch_bounds__(1024) void select_kernel(SelectArgs a) {
    __shared__ int hist[260];
    __shared__ int wave_tot[17];
    const int kc = blockIdx.x, b = blockIdx.y, C = a.C;
    const int sel = kc < C ? 0 : kc < 2 * C ? 1 : kc == 2 * C ? 2 : 3;
    if (a.discard_bits >> sel & 1u) return;
    const int64_t base = a.row_off[b];
    const int nk = a.n_kept ? a.n_kept[b] : (int)(a.row_off[b + 1] - base);
    if (nk <= 0) return;
    uint8_t* flag = a.sel_flag + base;
    if (a.topj >= nk) {   // maxj = min(topj, N'): every kept row is selected
        for (int i = threadIdx.x; i < nk; i += blockDim.x) flag[i] = 1;
        return;
    }
    // (psi_beta ranks by the SMALLEST background sum: its key order is inverted)
    const float* srow = a.stats + base;
    const bool direct = !a.compact || sel != 1;            // one stored column is the key (everything but compact psi_sigma)
    const float* col = srow + (int64_t)(!a.compact ? kc : sel == 0 ? kc : sel == 2 ? C + 2 : sel == 3 ? C + 3 : kc - C) * a.stride;
    const float* cm1 = srow + (int64_t)C * a.stride;
    const float* crd = srow + (int64_t)(C + 1) * a.stride;
    const uint32_t flip = sel == 3 ? 0xFFFFFFFFu : 0u;
    auto keyfn = [&](int i) { return direct ? (moc_key_desc(col[i]) ^ flip) : moc_key_desc(moc_softmax_from(col[i], cm1[i], crd[i])); };
    int take, ties;
    // ---- topj <= 1024: the topj-th largest of the 1024 per-thread maxima is a lower bound T0 of the topj-th largest
    // key (topj threads hold a key >= T0), so only keys >= T0 can be selected: a few hundred of 15,000.  They go to
    // an LDS list and the exact select runs on the list (and on the 1024 maxima) instead of on the column: two
    // sweeps over the keys instead of five.  Overflowing lists and ties that straddle the boundary (ascending-row
    // order needs the column order) take the general path below.
    if (a.topj <= 1024 && nk > 2048) {
        __shared__ uint32_t tmax[1024];
        __shared__ unsigned long long cand[SEL_CAND];
        __shared__ int n_cand;
        uint32_t mx = 0;
        for (int i = threadIdx.x; i < nk; i += 1024) { const uint32_t u = keyfn(i); mx = u > mx ? u : mx; }
        tmax[threadIdx.x] = mx;
        if (threadIdx.x == 0) n_cand = 0;
        __syncthreads();
        int t_take, t_ties;
        const uint32_t T0 = block_radix_select([&](int i) { return tmax[i]; }, 1024, a.topj, hist, &t_take, &t_ties);
        for (int i = threadIdx.x; i < nk; i += 1024) {
            const uint32_t u = keyfn(i);
            if (u >= T0) {
                const int pos = atomicAdd(&n_cand, 1);
                if (pos < SEL_CAND) cand[pos] = ((unsigned long long)u << 32) | (uint32_t)i;
            }
        }
        __syncthreads();
        const int nc = n_cand;
        if (nc <= SEL_CAND) {                                           // block-uniform; nc >= topj
            const uint32_t T = block_radix_select([&](int i) { return (uint32_t)(cand[i] >> 32); }, nc, a.topj, hist, &take, &ties);
            if (take == ties) {
                for (int i = threadIdx.x; i < nc; i += 1024)
                    if ((uint32_t)(cand[i] >> 32) >= T) flag[(uint32_t)cand[i]] = 1;
                return;
            }
        }
        __syncthreads();
    }
    const uint32_t T = block_radix_select(keyfn, nk, a.topj, hist, &take, &ties);
    if (take == ties) {
        for (int i = threadIdx.x; i < nk; i += blockDim.x)
            if (keyfn(i) >= T) flag[i] = 1;
        return;
    }
    // more keys tie at the boundary than fit: take the lowest rows first
    int running = 0;
    for (int c0 = 0; c0 < nk; c0 += blockDim.x) {
        const int i = c0 + threadIdx.x;
        const uint32_t u = i < nk ? keyfn(i) : 0u;
        const bool tie = i < nk && u == T;
        int tot;
        const int pos = moc_block_flag_scan(tie, wave_tot, &tot);
        if (i < nk && (u > T || (tie && running + pos < take))) flag[i] = 1;
        running += tot;
    }
}


// ---- wide banks: one workgroup per (slide, group of columns) -------------------------------------------------------
// select_kernel gives every one of the 2C+2 columns of a slide a workgroup of its own: at thirty classes 62 workgroups
// per slide, each a chain of ~40 barriers (two four-pass radix selects) around two sweeps of 15 keys per thread -- 541 us
// for 202 slides, the latency of 12,500 short workgroups, not bytes.  Here a workgroup takes SG_COLS columns of a slide
// through the SAME barriers: per-thread maxima of all its columns in one sweep, ONE four-pass radix select over the
// SG_COLS x 1024 maxima (a histogram per column, a wave per column for the scans), one sweep that lists the keys above
// each column's bound, one four-pass select over the lists, marks.  (Third form: the candidates of all the group's columns
// share ONE pool of tagged entries -- 64 KiB instead of eight 16-KiB lists, so two workgroups fit a CU and one's barriers
// and load latencies hide behind the other's work.)  (Round 3, second form: the bound comes from a SAMPLE of
// 1024 slots instead of a full sweep of per-thread maxima -- the slide's statistics are read once, not twice.)  Columns are ordered (psi_p[c], psi_sigma[c]) pairs,
// then psi_delta, psi_beta: with the compact statistics a pair shares its loads (v[c]; m1 and 1/den once per row).
// Workgroups of one slide are dealt to one XCD (dispatch order: consecutive multiples of 8), so the row-wide columns
// and the second sweep are L2 hits.  Rare cases -- a list that overflows, ties across the boundary that need the
// column order -- take the per-column path (select_column_general) inside the same workgroup.
constexpr int SG_COLS = 8;          // columns per workgroup
constexpr int SG_POOL = 8192;       // tagged candidate entries of all the group's columns together (64 KiB; = 8 x 1024 samples)
constexpr int SG_LDS_BYTES = SG_POOL * 8 + SG_COLS * 256 * 4 + 2048;      // 75,776: two workgroups per CU

// general exact top-j of ONE column (the tail of select_kernel): radix select over the column, ties lowest row first
template <typename KeyFn>
__device__ __forceinline__ void select_column_general(KeyFn keyfn, int nk, int topj, uint8_t* flag, int* hist, int* wave_tot) {
    int take, ties;
    const uint32_t T = block_radix_select(keyfn, nk, topj, hist, &take, &ties);
    if (take == ties) {
        for (int i = threadIdx.x; i < nk; i += blockDim.x)
            if (keyfn(i) >= T) flag[i] = 1;
        return;
    }
    int running = 0;
    for (int c0 = 0; c0 < nk; c0 += blockDim.x) {
        const int i = c0 + threadIdx.x;
        const uint32_t u = i < nk ? keyfn(i) : 0u;
        const bool tie = i < nk && u == T;
        int tot;
        const int pos = moc_block_flag_scan(tie, wave_tot, &tot);
        if (i < nk && (u > T || (tie && running + pos < take))) flag[i] = 1;
        running += tot;
    }
}

// The want-th largest key of each of up to SG_COLS key sets at once.  The sets live in ONE pool of tagged entries
// (key << 32 | set << 29 | payload): an entry is counted into its own set's histogram, so a pass costs one LDS read and one
// LDS atomic per entry whatever the number of sets.  Per-set state in LDS (st: [SG_COLS][8] ints: 0 prefix, 1 remaining,
// 2 keys tied at the boundary digit, 3 total keys of the set, 4 ok).  hist: [SG_COLS][256].  Ends in a barrier.
// Afterwards st[q][0] = the want-th largest key T, st[q][1] = how many keys equal to T belong to the top `want`,
// st[q][2] = how many keys equal T, st[q][4] = 0 when the set holds fewer than `want` keys (then the rest is undefined).
__device__ __forceinline__ void pool_radix_select(const unsigned long long* pool, int n, unsigned active, int want, int* hist, int* st) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < SG_COLS) {
        int* sq = st + threadIdx.x * 8;
        sq[0] = 0; sq[1] = want; sq[2] = 0; sq[3] = 0; sq[4] = (active >> threadIdx.x) & 1;
    }
    uint32_t pmask = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < SG_COLS * 256; i += 1024) hist[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 1024) {
            const unsigned long long e = pool[i];
            const int q = (int)(e >> 29) & 7;
            const uint32_t u = (uint32_t)(e >> 32);
            if ((u & pmask) == (uint32_t)st[q * 8]) atomicAdd(&hist[q * 256 + ((u >> shift) & 255)], 1);
        }
        __syncthreads();
        if (wave < SG_COLS && st[wave * 8 + 4]) {                             // wave q scans set q's histogram
            const int q = wave, L = lane;
            const int* hq = hist + q * 256;
            const int h0 = hq[4 * L], h1 = hq[4 * L + 1], h2 = hq[4 * L + 2], h3 = hq[4 * L + 3];
            int suf = h0 + h1 + h2 + h3;                                      // becomes the sum over lanes >= L
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_down(suf, off, 64);
                if (L + off < 64) suf += o;
            }
            const int total = __shfl(suf, 0, 64);
            const int rem = st[q * 8 + 1];
            if (rem > total) {                                                // fewer keys than wanted: the caller's other path
                if (L == 0) st[q * 8 + 4] = 0;
            } else {
                const int above = suf - (h0 + h1 + h2 + h3);                  // keys in digits > 4L + 3
                const int a3 = above, a2 = above + h3, a1 = a2 + h2, a0 = a1 + h1;
                int d = -1, ab = 0, hd = 0;
                if (a3 < rem && rem <= a3 + h3) { d = 4 * L + 3; ab = a3; hd = h3; }
                else if (a2 < rem && rem <= a2 + h2) { d = 4 * L + 2; ab = a2; hd = h2; }
                else if (a1 < rem && rem <= a1 + h1) { d = 4 * L + 1; ab = a1; hd = h1; }
                else if (a0 < rem && rem <= a0 + h0) { d = 4 * L; ab = a0; hd = h0; }
                if (d >= 0) {
                    st[q * 8] |= d << shift;
                    st[q * 8 + 1] = rem - ab;
                    st[q * 8 + 2] = hd;
                    if (shift == 24) st[q * 8 + 3] = total;
                }
            }
        }
        __syncthreads();
        pmask |= 255u << shift;
    }
}

// grid (groups * slides rounded up to 8), 1024 threads, SG_LDS_BYTES of dynamic LDS (two workgroups per CU)
template <int SG_ROWS, int MINW>
__global__ __launch_bounds__(1024, MINW) void select_group_kernel(SelectArgs a, int groups, int n_slides) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* pool = reinterpret_cast<unsigned long long*>(smem);             // [SG_POOL] tagged entries
    int* hist = reinterpret_cast<int*>(smem + (size_t)SG_POOL * 8);                     // [SG_COLS][256]
    int* st = hist + SG_COLS * 256;                                                     // [SG_COLS][8]
    int* n_pool = st + SG_COLS * 8;                                                     // [1] (+3 pad)
    int* wave_tot = n_pool + 4;                                                         // [20]
    int* hist1 = wave_tot + 20;                                                         // [260] for the per-column path
    const int lane = threadIdx.x & 63;
    // the workgroups of one slide on one XCD: linear ids L, L + 8, L + 16, ... share an XCD under round-robin dispatch
    // (speed only).  L = xcd + 8 * s: s-th workgroup of that XCD -> slide (s / groups) * 8 + xcd, group s % groups
    const int L = blockIdx.x, xcd = L & 7, sq = L >> 3;
    const int b = (sq / groups) * 8 + xcd, grp = sq % groups;
    if (b >= n_slides) return;                             // (the grid covers the slides rounded up to a multiple of 8)
    const int C = a.C, ncol = 2 * C + 2;
    const int p0 = grp * SG_COLS;
    const int nq = min(SG_COLS, ncol - p0);
    // ---- what this group reads and how a key is formed from it.  Position p of the pairing order is selector column
    // kc = c (p = 2c: psi_p), C + c (p = 2c + 1: psi_sigma), then 2C (psi_delta), 2C + 1 (psi_beta).  `src` = the rows of
    // `stats` this group loads per slot (at most 8), column q's key = form[q] of its source:
    //   0: key(v)    1: key(exp2((v - m1) log2 e) / den)  [compact layout]    2: ~key(v)  (psi_beta: smallest first)
    // Slots are fixed by position (no searching, no indexed registers): full layout -- slot q holds column q's own row;
    // compact layout -- slots 0, 1 = m1, 1/den, slot 2 + (q >> 1) = v[c] of the pair at columns q, q + 1 (or psi_delta's
    // gap), one slot further psi_beta's background sum (it is always the last column, right behind psi_delta).
    int src[SG_COLS], form[SG_COLS], kcs[SG_COLS];
    unsigned active = 0;
#pragma unroll
    for (int q = 0; q < SG_COLS; ++q) { src[q] = -1; form[q] = 0; kcs[q] = 0; }
#pragma unroll
    for (int q = 0; q < SG_COLS; ++q) {
        if (q >= nq) continue;
        const int p = p0 + q;
        const int kc = p < 2 * C ? ((p & 1) ? C + (p >> 1) : (p >> 1)) : p;
        kcs[q] = kc;
        const int sel = kc < C ? 0 : kc < 2 * C ? 1 : kc == 2 * C ? 2 : 3;
        if (a.discard_bits >> sel & 1u) continue;
        active |= 1u << q;
        form[q] = (a.compact && sel == 1) ? 1 : sel == 3 ? 2 : 0;
    }
    if (!active) return;
    if (!a.compact) {
#pragma unroll
        for (int q = 0; q < SG_COLS; ++q) if (active >> q & 1u) src[q] = kcs[q];
    } else {
        bool any_sigma = false;
#pragma unroll
        for (int q = 0; q < SG_COLS; ++q) any_sigma = any_sigma || ((active >> q & 1u) && form[q] == 1);
        if (any_sigma) { src[0] = C; src[1] = C + 1; }
#pragma unroll
        for (int q = 0; q < SG_COLS; ++q) {
            if (!(active >> q & 1u)) continue;
            const int kc = kcs[q];
            const int row = kc < C ? kc : kc < 2 * C ? kc - C : kc == 2 * C ? C + 2 : C + 3;
            const int slot = 2 + (q >> 1) + (form[q] == 2 ? 1 : 0);           // (q compile-time: a fixed register per case)
#pragma unroll
            for (int t = 2; t < SG_COLS; ++t) if (slot == t) src[t] = row;    // (a pair writes the same row twice)
        }
    }
    const int64_t base = a.row_off[b];
    const int nk = a.n_kept ? a.n_kept[b] : (int)(a.row_off[b + 1] - base);
    if (nk <= 0) return;
    uint8_t* flag = a.sel_flag + base;
    const float* srow = a.stats + base;
    if (a.topj >= nk) {                                    // maxj = min(topj, N'): every kept row is selected
        for (int i = threadIdx.x; i < nk; i += 1024) flag[i] = 1;
        return;
    }
    MOC_STAMP(20);
    auto load_row = [&](int i, float (&r)[SG_COLS]) {      // the group's sources at slot i: independent loads
#pragma unroll
        for (int t = 0; t < SG_COLS; ++t) r[t] = src[t] >= 0 ? srow[(int64_t)src[t] * a.stride + i] : 0.f;
    };
    auto keys_from = [&](const float (&r)[SG_COLS], uint32_t (&k)[SG_COLS]) {
#pragma unroll
        for (int q = 0; q < SG_COLS; ++q) {
            k[q] = 0;
            if (!(active >> q & 1u)) continue;
            constexpr int LAST = SG_COLS - 1;
            const int sa = 2 + (q >> 1), sb = sa + 1 < LAST ? sa + 1 : LAST;
            float v = a.compact ? (form[q] == 2 ? r[sb] : r[sa < LAST ? sa : LAST]) : r[q];
            if (form[q] == 1) v = moc_softmax_from(v, r[0], r[1]);            // (slots 0, 1: m1 and 1/den)
            const uint32_t u = moc_key_desc(v);
            k[q] = form[q] == 2 ? ~u : u;
        }
    };
    unsigned todo = 0;                                     // columns left to the per-column path
    if (nk > 2048) {
        // ---- a bound T0[q] per column from a SAMPLE of 1024 evenly spaced slots: the ks-th largest sampled key with
        // ks = twice the share the wanted topj have in the slide, so that about 2 topj keys lie at or above it (for
        // topj = 400 of 15,000: 810 +- 110).  The bound only decides how many candidates there are; the selection
        // itself is exact: a column must end up with at least topj candidates in the pool (else, ~1e-4 of the columns,
        // the per-column path).
        int ks = (int)((2048ll * a.topj + nk - 1) / nk) + 4;
        ks = ks > 1024 ? 1024 : ks;
        {
            // 64 evenly spaced runs of 16 consecutive slots (64 bytes of a column each), not 1024 single slots: single
            // slots 15 apart touch EVERY 128-byte line of the eight columns -- the sample then moves as many bytes as the
            // sweep itself (the kernel is HBM-bound: stamps: sample 11.8 + sweep 36 us of a workgroup's 75).  A run of 16
            // neighbours says less than 16 scattered slots where neighbouring patches look alike; the bound only sizes
            // the candidate lists (checked below), it does not decide anything
            const int i = (int)(((int64_t)(threadIdx.x >> 4) * nk) >> 6) + (int)(threadIdx.x & 15);
            float r[SG_COLS];
            uint32_t k[SG_COLS];
            load_row(i, r);
            keys_from(r, k);
#pragma unroll
            for (int q = 0; q < SG_COLS; ++q) pool[q * 1024 + threadIdx.x] = ((unsigned long long)k[q] << 32) | ((unsigned long long)q << 29);
        }
        __syncthreads();
        MOC_STAMP(21);
        pool_radix_select(pool, SG_COLS * 1024, active, ks, hist, st);
        MOC_STAMP(22);
        uint32_t T0[SG_COLS];
#pragma unroll
        for (int q = 0; q < SG_COLS; ++q) T0[q] = (uint32_t)st[q * 8];
        if (threadIdx.x == 0) n_pool[0] = 0;
        __syncthreads();                                   // everyone has its bounds; the pool is free again
        // ---- ONE sweep over the slide: the keys at or above their column's bound go to the pool, tagged with column
        // and slot.  SG_ROWS slots per thread in flight (SG_ROWS x up to 8 independent loads) before the first key is formed;
        // one LDS atomic per wave and slot batch (the wave's hits are placed by a prefix sum over its lanes).
        for (int ib = 0; ib < nk; ib += SG_ROWS * 1024) {          // (uniform trip count: the body shuffles across the wave)
            const int i0 = ib + (int)threadIdx.x;
            float r[SG_ROWS][SG_COLS];
#pragma unroll
            for (int u = 0; u < SG_ROWS; ++u) {
                const int i = i0 + u * 1024;
                load_row(i < nk ? i : nk - 1, r[u]);
            }
#pragma unroll
            for (int u = 0; u < SG_ROWS; ++u) {
                const int i = i0 + u * 1024;
                uint32_t k[SG_COLS];
                keys_from(r[u], k);
                unsigned hits = 0;
                if (i < nk) {
#pragma unroll
                    for (int q = 0; q < SG_COLS; ++q) hits |= ((active >> q & 1u) && k[q] >= T0[q]) ? 1u << q : 0u;
                }
                const int cnt = __popc(hits);
                int inc = cnt;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int o = __shfl_up(inc, off, 64);
                    if (lane >= off) inc += o;
                }
                const int wtot = __shfl(inc, 63, 64);
                if (wtot == 0) continue;                   // (uniform)
                int wbase = 0;
                if (lane == 0) wbase = atomicAdd(n_pool, wtot);
                wbase = __builtin_amdgcn_readfirstlane(wbase);
                int pos = wbase + inc - cnt;
#pragma unroll
                for (int q = 0; q < SG_COLS; ++q) {
                    if (hits >> q & 1u) {
                        if (pos < SG_POOL) pool[pos] = ((unsigned long long)k[q] << 32) | ((unsigned long long)q << 29) | (unsigned)i;
                        ++pos;
                    }
                }
            }
        }
        __syncthreads();
        MOC_STAMP(23);
        const int np = n_pool[0];
        if (np <= SG_POOL) {                               // (uniform)
            pool_radix_select(pool, np, active, a.topj, hist, st);
            MOC_STAMP(24);
            // a column is settled when it had at least topj candidates and no tie straddles the boundary
            for (int i = threadIdx.x; i < np; i += 1024) {
                const unsigned long long e = pool[i];
                const int q = (int)(e >> 29) & 7;
                const int* sq = st + q * 8;
                if (sq[4] && sq[1] == sq[2] && (uint32_t)(e >> 32) >= (uint32_t)sq[0]) flag[(uint32_t)e & 0x1FFFFFFFu] = 1;
            }
            MOC_STAMP(25);
#pragma unroll
            for (int q = 0; q < SG_COLS; ++q)
                if ((active >> q & 1u) && !(st[q * 8 + 4] && st[q * 8 + 1] == st[q * 8 + 2])) todo |= 1u << q;
        } else {
            todo = active;                                 // the pool overflowed: which column lost entries is not known
        }
        if (!todo) return;
        __syncthreads();
    } else {
        todo = active;                                     // short slides: the per-column path (a sample says little)
    }
    // ---- per-column path for what is left (short slides, too few candidates, boundary ties that need the row order)
#pragma unroll
    for (int q = 0; q < SG_COLS; ++q) {
        if (!(todo >> q & 1u)) continue;                   // uniform
        const int kc = kcs[q];
        select_column_general([&](int i) { return sel_key(a, srow, kc, i); }, nk, a.topj, flag, hist1, wave_tot);
        __syncthreads();
    }
}

struct CompactArgs {
    const float* stats;
    const int64_t* row_off;
    const int64_t* x_off;    // nullable
    const int32_t* kept;     // nullable
    const int32_t* n_kept;   // nullable
    const uint8_t* sel_flag;
    int32_t* sel_idx;
    int64_t* sel_row;
    int32_t* n_sel;
    float* cand;
    const unsigned char* X;
    unsigned char* selected_feat;   // nullable
    int cand_inline;                // compact_kernel copies the candidate columns itself
    int64_t stride;
    int C, row_bytes;
    int compact;                    // MOC_STATS_COMPACT layout of `stats`
};

// candidate score k (0..C-1 s_p, C..2C-1 s_sigma, 2C s_delta, 2C+1 s_beta = MAX background logit: main_moc.py:359-366)
// of the row at `s` (= stats + slot), either layout
__device__ __forceinline__ float cand_value(const float* s, int64_t stride, int C, int k, int compact) {
    if (!compact) return s[(int64_t)(k == 2 * C + 1 ? 2 * C + 2 : k) * stride];
    if (k < C) return s[(int64_t)k * stride];
    if (k < 2 * C) return moc_softmax_from(s[(int64_t)(k - C) * stride], s[(int64_t)C * stride], s[(int64_t)(C + 1) * stride]);
    return s[(int64_t)(k == 2 * C ? C + 2 : C + 4) * stride];
}

// grid (n_slides)
__global__ __launch_bounds__(1024) void compact_kernel(CompactArgs a) {
    __shared__ int wave_tot[17];
    const int b = blockIdx.x, C = a.C;
    const int64_t base = a.row_off[b];
    const int64_t xbase = a.x_off ? a.x_off[b] : base;
    const int nk = a.n_kept ? a.n_kept[b] : (int)(a.row_off[b + 1] - base);
    int running = 0;
    // a thread owns 16 consecutive slots: its flags come in one batch of loads, one block scan places them
    for (int c0 = 0; c0 < nk; c0 += 16 * 1024) {
        const int i0 = c0 + (int)threadIdx.x * 16;
        unsigned bits = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = i0 + q;
            const uint8_t f = a.sel_flag[base + (i < nk ? i : nk - 1)];
            bits |= (i < nk && f != 0 ? 1u : 0u) << q;
        }
        int tot;
        const int pos = moc_block_count_scan(__popc(bits), wave_tot, &tot);
        int64_t o = base + running + pos;
        while (bits) {                                   // ascending: lowest set bit first
            const int i = i0 + __ffs((int)bits) - 1;
            bits &= bits - 1;
            a.sel_idx[o] = i;
            a.sel_row[o] = xbase + (a.kept ? a.kept[base + i] : i);
            ++o;
        }
        running += tot;
    }
    if (threadIdx.x == 0) a.n_sel[b] = running;
    if (a.cand_inline) {
        // Few columns (C <= 4): the candidate columns are copied here, no second launch -- as a phase of its own,
        // one selected slot per thread: every load independent of the others and the stores coalesced (inside the
        // loop above a thread with five selected rows walked five dependent rounds of 2C + 2 scattered loads).
        __threadfence_block();
        __syncthreads();                                   // the slide's sel_idx, written by this workgroup, is visible
        for (int o = threadIdx.x; o < running; o += 1024) {
            const int i = a.sel_idx[base + o];
            const float* s = a.stats + base + i;
            float* c = a.cand + base + o;
            float v[10];
#pragma unroll
            for (int k = 0; k < 10; ++k)
                if (k < 2 * C + 2) v[k] = cand_value(s, a.stride, C, k, a.compact);   // the last one is s_beta = max background
#pragma unroll
            for (int k = 0; k < 10; ++k)
                if (k < 2 * C + 2) c[(int64_t)k * a.stride] = v[k];
        }
    }
}

// grid (ceil(max_rows / 256), ceil((2C+2) / CAND_COLS), n_slides): candidate columns of the selected rows,
// one selected slot per thread, CAND_COLS columns per workgroup.  (With the copy inside compact_kernel -- one
// workgroup per slide, 2C+2 strided columns per thread -- that kernel took 1.6 ms at 64 classes, longer than
// the score pass.)
constexpr int CAND_COLS = 8;
__global__ __launch_bounds__(256) void cand_gather_kernel(CompactArgs a) {
    const int b = blockIdx.z, C = a.C;
    const int64_t base = a.row_off[b];
    const int S = a.n_sel[b];
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= S) return;
    const int i = a.sel_idx[base + o];
    const int k_lo = blockIdx.y * CAND_COLS, k_hi = min(k_lo + CAND_COLS, 2 * C + 2);
    const float* s = a.stats + base + i;
    float* c = a.cand + base + o;
    // (every statistic of the thread's columns requested before the first candidate is stored: load -> store per column
    // in one loop is a round trip per column -- the compiler cannot tell that `cand` and `stats` do not overlap)
    float v[CAND_COLS];
    if (a.compact) {                                   // s_sigma re-formed from (v, m1, 1/den); m1 and 1/den once per thread
        const float m1 = s[(int64_t)C * a.stride], rden = s[(int64_t)(C + 1) * a.stride];
#pragma unroll
        for (int q = 0; q < CAND_COLS; ++q) {
            const int k = k_lo + q < k_hi ? k_lo + q : k_hi - 1;
            const int row = k < C ? k : k < 2 * C ? k - C : (k == 2 * C ? C + 2 : C + 4);
            v[q] = s[(int64_t)row * a.stride];
        }
#pragma unroll
        for (int q = 0; q < CAND_COLS; ++q) {
            const int k = k_lo + q;
            if (k < k_hi) c[(int64_t)k * a.stride] = (k >= C && k < 2 * C) ? moc_softmax_from(v[q], m1, rden) : v[q];
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < CAND_COLS; ++q) {              // candidate k <- statistic k; the last one is s_beta = max background
        const int k = k_lo + q < k_hi ? k_lo + q : k_hi - 1;
        v[q] = s[(int64_t)(k == 2 * C + 1 ? 2 * C + 2 : k) * a.stride];
    }
#pragma unroll
    for (int q = 0; q < CAND_COLS; ++q)
        if (k_lo + q < k_hi) c[(int64_t)(k_lo + q) * a.stride] = v[q];
}

// grid (ceil(max_rows/16), n_slides): copies selected rows (16 B per thread per step)
__global__ __launch_bounds__(256) void gather_rows_kernel(CompactArgs a) {
    const int b = blockIdx.y;
    const int64_t base = a.row_off[b];
    const int S = a.n_sel[b];
    const int vec_per_row = a.row_bytes / 16;
    for (int s = blockIdx.x * 16 + (threadIdx.x >> 4); s < S; s += gridDim.x * 16) {
        const uint4* src = reinterpret_cast<const uint4*>(a.X + a.sel_row[base + s] * (int64_t)a.row_bytes);
        uint4* dst = reinterpret_cast<uint4*>(a.selected_feat + (base + s) * (int64_t)a.row_bytes);
        for (int v0 = threadIdx.x & 15; v0 < vec_per_row; v0 += 64) {     // four pieces requested, then stored (not load -> store four times)
            uint4 piece[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) piece[u] = src[v0 + 16 * u < vec_per_row ? v0 + 16 * u : v0];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (v0 + 16 * u < vec_per_row) dst[v0 + 16 * u] = piece[u];
        }
    }
}

// ---- compact hand-over of phase A's result (exact-sequential multi-GPU, SURVEY.md section 8e mode 1) ----------
// grid (ceil(cap/16), n): slide slide0+y's selected rows -> feat_out[y][cap][D] (16 B per thread per step) and its
// candidate scores -> cand_out[y][2C+2][cap]; entries at and beyond n_sel are left as they are.
struct PackArgs {
    const unsigned char* X;
    const int64_t* row_off;
    const int64_t* sel_row;
    const int32_t* n_sel;
    const float* cand;
    unsigned char* feat_out;
    float* cand_out;
    int64_t stride;
    int64_t out_rows;       // packed form: rows the outputs hold
    int row_bytes, C, cap, slide0, packed;
};

// packed != 0: no padding -- slide slide0+y's rows start at row sum_{q<y} min(n_sel[slide0+q], cap) of feat_out
// [sum S][D], and its candidate scores are the ROWS cand_out[that row + t][2C+2] (row-major: what an all-gather of
// unequal pieces can carry; the receiver transposes).
__global__ __launch_bounds__(256) void pack_selected_kernel(PackArgs a) {
    const int y = blockIdx.y, b = a.slide0 + y;
    const int64_t base = a.row_off[b];
    const int S = min(a.n_sel[b], a.cap);
    if (blockIdx.x * 16 >= S) return;
    __shared__ int64_t pref_s;
    int64_t pref = (int64_t)y * a.cap;
    if (a.packed) {
        if (threadIdx.x < 64) {                           // one wave sums the y counts in front of this slide
            int64_t part = 0;
            for (int q = threadIdx.x; q < y; q += 64) part += min(a.n_sel[a.slide0 + q], a.cap);
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
            if (threadIdx.x == 0) pref_s = part;
        }
        __syncthreads();
        pref = pref_s;
    }
    const int vec_per_row = a.row_bytes / 16;
    const int s = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (s < S && (!a.packed || pref + s < a.out_rows)) {
        const uint4* src = reinterpret_cast<const uint4*>(a.X + a.sel_row[base + s] * (int64_t)a.row_bytes);
        uint4* dst = reinterpret_cast<uint4*>(a.feat_out + (pref + s) * a.row_bytes);
        for (int v0 = threadIdx.x & 15; v0 < vec_per_row; v0 += 64) {     // four pieces requested, then stored (not load -> store four times)
            uint4 piece[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) piece[u] = src[v0 + 16 * u < vec_per_row ? v0 + 16 * u : v0];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (v0 + 16 * u < vec_per_row) dst[v0 + 16 * u] = piece[u];
        }
    }
    const int nk = 2 * a.C + 2;
    if (a.packed) {
        for (int e = threadIdx.x; e < nk * 16; e += 256) {
            const int t = blockIdx.x * 16 + e / nk, k = e - (e / nk) * nk;      // consecutive threads: consecutive floats of a row
            if (t < S && pref + t < a.out_rows) a.cand_out[(pref + t) * nk + k] = a.cand[(int64_t)k * a.stride + base + t];
        }
        return;
    }
    for (int e = threadIdx.x; e < nk * 16; e += 256) {
        const int k = e >> 4, t = blockIdx.x * 16 + (e & 15);
        if (t < S) a.cand_out[((int64_t)y * nk + k) * a.cap + t] = a.cand[(int64_t)k * a.stride + base + t];
    }
}

// ---- generic top-K mean ------------------------------------------------------
constexpr int TK_CAND = 1024;      // candidate list of topk_mean_kernel's K <= 16 path

struct TopkArgs {
    const float* keys;
    const float* vals;
    const int64_t* seg_off;
    const int32_t* seg_len;   // nullable
    float* pooled;
    int32_t* idx_out;         // nullable
    int32_t* cnt_out;         // nullable
    int64_t key_stride, val_stride;
    int C, K, smallest, seg0;
};

// grid (C, n_seg).  All LDS is dynamic (16-B aligned base): [P x u64 list][P x f32 vals][scratch ints],
// P = K rounded up to a power of two.
__global__ __launch_bounds__(1024) void topk_mean_kernel(TopkArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int Pmax = 1;
    while (Pmax < a.K) Pmax <<= 1;
    float* lvals = reinterpret_cast<float*>(smem + (size_t)Pmax * 8);
    int* hist = reinterpret_cast<int*>(smem + (size_t)Pmax * 12);   // [260]
    int* wave_tot = hist + 260;                                     // [17]
    int& n_list = hist[280];
    const int c = blockIdx.x, seg = a.seg0 + blockIdx.y;
    const int64_t base = a.seg_off[seg];
    const int n = a.seg_len ? a.seg_len[seg] : (int)(a.seg_off[seg + 1] - base);
    const int out = seg * a.C + c;
    if (n <= 0) {   // mean over an empty set: NaN, like torch
        if (threadIdx.x == 0) {
            a.pooled[out] = __uint_as_float(0x7FC00000u);
            if (a.cnt_out) a.cnt_out[out] = 0;
        }
        return;
    }
    const float* kcol = a.keys + (int64_t)c * a.key_stride + base;
    const float* vcol = a.vals + (int64_t)c * a.val_stride + base;
    const uint32_t flip = a.smallest ? 0xFFFFFFFFu : 0u;
    auto keyfn = [&](int i) { return moc_key_desc(kcol[i]) ^ flip; };
    const int k = a.K < n ? a.K : n;
    int P = 1;
    while (P < k) P <<= 1;
    unsigned long long* list = reinterpret_cast<unsigned long long*>(smem);   // (key << 32) | ~row : descending
    if (threadIdx.x == 0) n_list = 0;
    for (int i = threadIdx.x; i < P; i += blockDim.x) list[i] = 0ull;
    // ---- K <= 16: no radix passes.  The k-th largest of the 16 per-wave maxima is a lower bound T0 of the k-th
    // largest key (k waves hold a key >= T0), so only keys >= T0 can be in the top k -- typically a few dozen of
    // thousands.  They go to an LDS list (TK_CAND entries) and one wave extracts the k largest in order, ties by
    // ascending row: the same list the radix path ends with, after two sweeps over the keys instead of five.
    // Too many candidates (flat key distributions, fewer than k waves with a key): the radix path below.
    bool done = false;                                                  // block-uniform
    if (a.K <= 16 && k < n) {
        unsigned long long* cand = reinterpret_cast<unsigned long long*>(((uintptr_t)(hist + 288) + 7) & ~(uintptr_t)7);   // [TK_CAND]
        int* wmax_s = wave_tot;                                         // [16] per-wave maxima, then [16] = T0
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        // up to 16 keys per thread stay in registers between the two sweeps (n <= 16,384: every union of the reference's
        // configurations); longer segments read the keys again (L2)
        constexpr int RK = 16;
        const bool in_regs = n <= RK * 1024;
        uint32_t kr[RK];
        uint32_t mx = 0;
        if (in_regs) {
#pragma unroll
            for (int q = 0; q < RK; ++q) {
                const int i = (int)threadIdx.x + q * 1024;
                { const uint32_t uk = keyfn(i < n ? i : n - 1); kr[q] = i < n ? uk : 0u; }   // (clamped, not branched: the loads of a batch go out together)
                mx = kr[q] > mx ? kr[q] : mx;
            }
        } else {
            for (int i = threadIdx.x; i < n; i += blockDim.x) { const uint32_t u = keyfn(i); mx = u > mx ? u : mx; }
        }
        mx = (uint32_t)(wave_max_u64((unsigned long long)mx));
        if (lane == 0) wmax_s[wave] = (int)mx;
        __syncthreads();
        if (threadIdx.x < 16) {
            const uint32_t v = (uint32_t)wmax_s[threadIdx.x];
            int rank = 0;
            for (int j = 0; j < 16; ++j) {
                const uint32_t o = (uint32_t)wmax_s[j];
                rank += (o > v || (o == v && j < (int)threadIdx.x)) ? 1 : 0;
            }
            if (rank == k - 1) wmax_s[16] = (int)v;
        }
        __syncthreads();
        const uint32_t T0 = (uint32_t)wmax_s[16];
        auto offer = [&](uint32_t u, int i) {
            if (u >= T0) {
                const int pos = atomicAdd(&n_list, 1);
                if (pos < TK_CAND) cand[pos] = ((unsigned long long)u << 32) | (uint32_t)(~(uint32_t)i);
            }
        };
        if (in_regs) {
#pragma unroll
            for (int q = 0; q < RK; ++q) {
                const int i = (int)threadIdx.x + q * 1024;
                if (i < n) offer(kr[q], i);
            }
        } else {
            for (int i = threadIdx.x; i < n; i += blockDim.x) offer(keyfn(i), i);
        }
        __syncthreads();
        const int nc = n_list;
        __syncthreads();
        if (nc <= 64) {                                                 // the usual case: one candidate per lane
            if (wave == 0) {
                unsigned long long mine = lane < nc ? cand[lane] : 0ull;
                for (int r = 0; r < k; ++r) {
                    const unsigned long long best = wave_max_u64(mine);   // keys are distinct (row in the low word)
                    mine = mine == best ? 0ull : mine;
                    if (lane == 0) list[r] = best;
                }
            }
            done = true;
        } else if (nc <= TK_CAND) {                                     // (nc >= k: k waves contributed)
            if (wave == 0) {
                unsigned long long mine[TK_CAND / 64];
#pragma unroll
                for (int q = 0; q < TK_CAND / 64; ++q) mine[q] = q * 64 + lane < nc ? cand[q * 64 + lane] : 0ull;
                for (int r = 0; r < k; ++r) {
                    unsigned long long best = 0ull;
#pragma unroll
                    for (int q = 0; q < TK_CAND / 64; ++q) best = mine[q] > best ? mine[q] : best;
                    best = wave_max_u64(best);                          // keys are distinct (row in the low word)
#pragma unroll
                    for (int q = 0; q < TK_CAND / 64; ++q) mine[q] = mine[q] == best ? 0ull : mine[q];
                    if (lane == 0) list[r] = best;
                }
            }
            done = true;
        } else if (threadIdx.x == 0) n_list = 0;
        __syncthreads();
    }
    int take = 0, ties = 0;
    uint32_t T = 0;
    if (done) {
    } else if (k < n) T = block_radix_select(keyfn, n, k, hist, &take, &ties);
    else __syncthreads();
    if (done) {
    } else if (k == n || take == ties) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const uint32_t u = keyfn(i);
            if (k == n || u >= T) list[atomicAdd(&n_list, 1)] = ((unsigned long long)u << 32) | (uint32_t)(~(uint32_t)i);
        }
    } else {
        int running = 0;
        for (int c0 = 0; c0 < n; c0 += blockDim.x) {
            const int i = c0 + threadIdx.x;
            const uint32_t u = i < n ? keyfn(i) : 0u;
            const bool tie = i < n && u == T;
            int tot;
            const int pos = moc_block_flag_scan(tie, wave_tot, &tot);
            if (i < n && (u > T || (tie && running + pos < take)))
                list[atomicAdd(&n_list, 1)] = ((unsigned long long)u << 32) | (uint32_t)(~(uint32_t)i);
            running += tot;
        }
    }
    __syncthreads();
    // bitonic sort, descending (key desc, then row asc); padding zeros sink to the end
    for (int size = 2; size <= P && !done; size <<= 1) {
        for (int st = size >> 1; st > 0; st >>= 1) {
            for (int i = threadIdx.x; i < P; i += blockDim.x) {
                const int jx = i ^ st;
                if (jx > i) {
                    const unsigned long long x = list[i], y = list[jx];
                    const bool desc = (i & size) == 0;
                    if (desc ? x < y : x > y) { list[i] = y; list[jx] = x; }
                }
            }
            __syncthreads();
        }
    }
    if (a.idx_out)
        for (int i = threadIdx.x; i < a.K; i += blockDim.x)
            a.idx_out[(int64_t)out * a.K + i] = i < k ? (int32_t)(~(uint32_t)(list[i] & 0xFFFFFFFFull)) : -1;
    // mean of the k values: gather in rank order, fixed-shape pairwise tree (deterministic)
    // (keys and values in one array -- pooling the mixed scores: the value is the key read backwards, no second gather;
    // a canonical zero does not say which zero it was)
    const bool same = a.keys == a.vals && a.key_stride == a.val_stride;
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        float v = 0.f;
        if (i < k) {
            const uint32_t u = (uint32_t)(list[i] >> 32) ^ flip;
            if (same && u != 0x80000000u) v = __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
            else v = vcol[(int)(~(uint32_t)(list[i] & 0xFFFFFFFFull))];
        }
        lvals[i] = v;
    }
    __syncthreads();
    if (k <= 64) {
        if (threadIdx.x == 0) {   // short lists: sequential, largest first (as a [k]-row mean would)
            float s = 0.f;
            for (int i = 0; i < k; ++i) s += lvals[i];
            a.pooled[out] = s / (float)k;
        }
    } else {
        for (int st = P >> 1; st > 0; st >>= 1) {
            for (int i = threadIdx.x; i < st; i += blockDim.x) lvals[i] += lvals[i + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) a.pooled[out] = lvals[0] / (float)k;
    }
    if (threadIdx.x == 0 && a.cnt_out) a.cnt_out[out] = k;
}


// ---- top-K mean, ONE WAVE per (segment, class): for launches of thousands of tasks (an evaluation pass of a wide bank:
// 202 slides x 30 classes) -- the 1,024-thread kernel above runs them twelve rounds of 512 workgroups with six barriers
// each (240 us per 202 x 15,000 x 30); here every task is in flight at once and no wave waits for another.
// K <= 16.  A lower bound T0 of the k-th largest key from the lane maxima of a sample (long segments) or of all keys; one
// sweep: keys >= T0 into the wave's LDS list (WT_CAP entries, positions by ballot prefix); then k rounds of wave maximum
// over (key << 32 | ~row): the order of the kernel above -- key descending, ties by ascending row -- and the same
// sequential sum.  A list that overflows even with the bound from all keys (flat key distributions): k rounds over ALL
// keys, each finding the largest entry below the previous one (slow, exact, rare).
constexpr int WT_CAP = 256;          // candidate entries per wave
constexpr int WT_WAVES = 4;          // tasks per workgroup

__global__ __launch_bounds__(64 * WT_WAVES) void topk_mean_wave_kernel(TopkArgs a, int n_tasks) {
    __shared__ unsigned long long cand_s[WT_WAVES][WT_CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int task = blockIdx.x * WT_WAVES + wave;
    if (task >= n_tasks) return;                                   // (no barrier below: waves are on their own)
    const int seg = a.seg0 + task / a.C, c = task - (task / a.C) * a.C;
    const int64_t base = a.seg_off[seg];
    const int n = a.seg_len ? a.seg_len[seg] : (int)(a.seg_off[seg + 1] - base);
    const int out = seg * a.C + c;
    if (n <= 0) {   // mean over an empty set: NaN, like torch
        if (lane == 0) {
            a.pooled[out] = __uint_as_float(0x7FC00000u);
            if (a.cnt_out) a.cnt_out[out] = 0;
        }
        return;
    }
    const float* kcol = a.keys + (int64_t)c * a.key_stride + base;
    const float* vcol = a.vals + (int64_t)c * a.val_stride + base;
    const uint32_t flip = a.smallest ? 0xFFFFFFFFu : 0u;
    auto keyfn = [&](int i) { return moc_key_desc(kcol[i]) ^ flip; };
    const int k = a.K < n ? a.K : n;
    unsigned long long* cand = cand_s[wave];
    unsigned long long mine[WT_CAP / 64];
#pragma unroll
    for (int q = 0; q < WT_CAP / 64; ++q) mine[q] = 0ull;
    bool listed = false;                                           // wave-uniform: `mine` holds every candidate
    if (n <= WT_CAP) {
        // short segments: every key is a candidate
#pragma unroll
        for (int q = 0; q < WT_CAP / 64; ++q) {
            const int i = q * 64 + lane;
            if (i < n) mine[q] = ((unsigned long long)keyfn(i) << 32) | (uint32_t)(~(uint32_t)i);
        }
        listed = true;
    } else {
        // ---- the bound: the k-th largest of the 64 lane maxima over (a) eight runs of 256 consecutive keys spread over the
        // segment -- long segments: one pass over the keys instead of two, the kernel is bound by those bytes -- or (b) all
        // keys.  Either is a lower bound of the k-th largest key (k lanes hold a key >= it; every lane holds a key: n > 256).
        auto lane_bound = [&](bool sample) -> uint32_t {
            uint32_t mx = 0;
            if (sample) {
                for (int r = 0; r < 8; ++r) {
                    const int start = (int)(((int64_t)r * n) >> 3);
                    uint32_t u[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int i = start + q * 64 + lane;
                        { const uint32_t uk = keyfn(i < n ? i : n - 1); u[q] = i < n ? uk : 0u; }    // (clamped, not branched: the loads of a batch go out together)
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) mx = u[q] > mx ? u[q] : mx;
                }
            } else {
                for (int i0 = 0; i0 < n; i0 += 512) {
                    uint32_t u[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int i = i0 + q * 64 + lane;
                        { const uint32_t uk = keyfn(i < n ? i : n - 1); u[q] = i < n ? uk : 0u; }    // (clamped, not branched: the loads of a batch go out together)
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) mx = u[q] > mx ? u[q] : mx;
                }
            }
            uint32_t T = 0;
            bool alive = true;
            for (int r = 0; r < k; ++r) {                           // k rounds, one lane knocked out per round
                const uint32_t best = (uint32_t)wave_max_u64(alive ? (unsigned long long)mx : 0ull);
                const unsigned long long m = __ballot(alive && mx == best);
                const int first = __ffsll((long long)m) - 1;
                if (lane == first) alive = false;
                T = best;
            }
            return T;
        };
        // ---- the sweep: candidates >= T0, positions by ballot prefix
        auto collect = [&](uint32_t T0) -> int {
            int cnt = 0;
            for (int i0 = 0; i0 < n; i0 += 512) {
                uint32_t u[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int i = i0 + q * 64 + lane;
                    { const uint32_t uk = keyfn(i < n ? i : n - 1); u[q] = i < n ? uk : 0u; }    // (clamped, not branched: the loads of a batch go out together)
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int i = i0 + q * 64 + lane;
                    const bool hit = i < n && u[q] >= T0;
                    const unsigned long long m = __ballot(hit);
                    if (m == 0ull) continue;                        // (uniform)
                    const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
                    if (hit && pos < WT_CAP) cand[pos] = ((unsigned long long)u[q] << 32) | (uint32_t)(~(uint32_t)i);
                    cnt += __popcll(m);
                }
            }
            return cnt;
        };
        const bool sampled = n > 4096;
        int cnt = collect(lane_bound(sampled));
        if (cnt > WT_CAP && sampled) cnt = collect(lane_bound(false));   // (the runs were not typical of the segment: the bound from all keys)
        if (cnt <= WT_CAP) {
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own LDS writes, before it reads them back
#pragma unroll
            for (int q = 0; q < WT_CAP / 64; ++q) mine[q] = q * 64 + lane < cnt ? cand[q * 64 + lane] : 0ull;
            listed = true;
        }
    }
    // ---- the k largest in order; lane r keeps entry r
    unsigned long long mylist = 0ull;
    if (listed) {
        for (int r = 0; r < k; ++r) {
            unsigned long long best = 0ull;
#pragma unroll
            for (int q = 0; q < WT_CAP / 64; ++q) best = mine[q] > best ? mine[q] : best;
            best = wave_max_u64(best);                              // entries are distinct (row in the low word)
#pragma unroll
            for (int q = 0; q < WT_CAP / 64; ++q) mine[q] = mine[q] == best ? 0ull : mine[q];
            if (lane == r) mylist = best;
        }
    } else {
        unsigned long long prev = 0ull;
        for (int r = 0; r < k; ++r) {
            unsigned long long best = 0ull;
            for (int i = lane; i < n; i += 64) {
                const unsigned long long e = ((unsigned long long)keyfn(i) << 32) | (uint32_t)(~(uint32_t)i);
                best = ((r == 0 || e < prev) && e > best) ? e : best;
            }
            best = wave_max_u64(best);
            if (lane == r) mylist = best;
            prev = best;
        }
    }
    if (a.idx_out && lane < a.K)
        a.idx_out[(int64_t)out * a.K + lane] = lane < k ? (int32_t)(~(uint32_t)(mylist & 0xFFFFFFFFull)) : -1;
    // mean of the k values, summed in rank order (as the kernel above does for k <= 64)
    const bool same = a.keys == a.vals && a.key_stride == a.val_stride;
    float v = 0.f;
    if (lane < k) {
        const uint32_t u = (uint32_t)(mylist >> 32) ^ flip;
        if (same && u != 0x80000000u) v = __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
        else v = vcol[(int)(~(uint32_t)(mylist & 0xFFFFFFFFull))];
    }
    float sum = 0.f;
    for (int r = 0; r < k; ++r) sum += __shfl(v, r, 64);
    if (lane == 0) {
        a.pooled[out] = sum / (float)k;
        if (a.cnt_out) a.cnt_out[out] = k;
    }
}


}  // namespace

extern "C" int moc_select(const moc_batch_t* B, moc_stream_t stream) {
    if (int rc = moc_check_batch(B, "moc_select")) return rc;
    MOC_REQUIRE(B->stats && B->sel_flag, "moc_select: null stats/sel_flag");
    SelectArgs a;
    a.stats = B->stats; a.row_off = B->row_off; a.n_kept = B->mask ? B->n_kept : nullptr;
    a.sel_flag = B->sel_flag; a.stride = B->total_rows; a.C = B->C; a.topj = B->topj;
    a.discard_bits = B->discard_bits;
    a.compact = (B->flags & MOC_STATS_COMPACT) ? 1 : 0;
    const int ncol = 2 * B->C + 2;
    static const int force = getenv("MOC_SELECT_KERNEL") ? atoi(getenv("MOC_SELECT_KERNEL")) : 0;   // diagnostic: 1 per column, 2 grouped
    if ((ncol > SG_COLS || B->n_slides >= 128 || force == 2) && force != 1 && !(B->flags & MOC_SELECT_PER_COLUMN) && B->topj <= 1024) {
        // wide banks: one workgroup per slide and group of SG_COLS columns (select_group_kernel).  Narrow banks take it for
        // launches over many slides (evaluation: 202 x 15,000 x 2 classes 73 against 92 us, 3 classes 116 against 145); for a
        // 32-slide train pass one workgroup per column is the faster one (15 against 47 us: too few workgroups otherwise)
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)select_group_kernel<2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, SG_LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)select_group_kernel<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, SG_LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)select_group_kernel<1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, SG_LDS_BYTES);
            attr_set = true;
        }
        const int groups = moc_cdiv(ncol, SG_COLS);
        const int n8 = (B->n_slides + 7) & ~7;             // a slide's workgroups share an XCD: see the kernel's id mapping
        // <slots in flight per thread, waves per SIMD>: measured at 202 x 15,000 x 30 classes (scripts/bench_select.py):
        // <1, 8> 394 us (59 registers, two workgroups per CU), <2, 8> 402 (64 registers, 6 spilled), <4, 4> 459 (one
        // workgroup per CU); one workgroup per column (select_kernel): 965
        static const int variant = getenv("MOC_SELECT_VARIANT") ? atoi(getenv("MOC_SELECT_VARIANT")) : 2;
        if (variant == 1) select_group_kernel<4, 4><<<dim3(groups * n8), 1024, SG_LDS_BYTES, (hipStream_t)stream>>>(a, groups, B->n_slides);
        else if (variant == 0) select_group_kernel<2, 8><<<dim3(groups * n8), 1024, SG_LDS_BYTES, (hipStream_t)stream>>>(a, groups, B->n_slides);
        else select_group_kernel<1, 8><<<dim3(groups * n8), 1024, SG_LDS_BYTES, (hipStream_t)stream>>>(a, groups, B->n_slides);
        MOC_CHECK_LAUNCH("moc_select(group)");
        return MOC_OK;
    }
    dim3 grid(ncol, B->n_slides);
    select_kernel<<<grid, 1024, 0, (hipStream_t)stream>>>(a);
    MOC_CHECK_LAUNCH("moc_select");
    return MOC_OK;
}

extern "C" int moc_gather_candidates(const moc_batch_t* B, void* selected_feat, moc_stream_t stream) {
    if (int rc = moc_check_batch(B, "moc_gather_candidates")) return rc;
    MOC_REQUIRE(B->stats && B->sel_flag && B->sel_idx && B->sel_row && B->n_sel && B->cand,
                "moc_gather_candidates: null work array");
    CompactArgs a;
    a.stats = B->stats; a.row_off = B->row_off; a.x_off = B->x_off;
    a.kept = B->mask ? B->kept : nullptr; a.n_kept = B->mask ? B->n_kept : nullptr;
    a.sel_flag = B->sel_flag; a.sel_idx = B->sel_idx; a.sel_row = B->sel_row; a.n_sel = B->n_sel;
    a.cand = B->cand; a.X = (const unsigned char*)B->X; a.selected_feat = (unsigned char*)selected_feat;
    a.stride = B->total_rows; a.C = B->C; a.row_bytes = B->D * moc_elem_size(B->dtype);
    a.compact = (B->flags & MOC_STATS_COMPACT) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    a.cand_inline = B->C <= 4;
    compact_kernel<<<B->n_slides, 1024, 0, s>>>(a);
    if (!a.cand_inline && !(B->flags & MOC_CAND_FROM_STATS)) {     // (evaluation: the forward reads the statistics itself)
        MOC_CHECK_LAUNCH("moc_gather_candidates(compact)");
        cand_gather_kernel<<<dim3(moc_cdiv(B->max_rows, 256), moc_cdiv(2 * B->C + 2, CAND_COLS), B->n_slides), 256, 0, s>>>(a);
    }
    MOC_CHECK_LAUNCH("moc_gather_candidates(compact)");
    if (selected_feat) {
        int gx = moc_cdiv(B->max_rows, 16);
        if (gx > 4096) gx = 4096;
        gather_rows_kernel<<<dim3(gx, B->n_slides), 256, 0, s>>>(a);
        MOC_CHECK_LAUNCH("moc_gather_candidates(gather)");
    }
    return MOC_OK;
}

extern "C" int moc_pack_selected(const moc_batch_t* B, int slide0, int n, int cap, void* feat_out, float* cand_out,
                                 moc_stream_t stream) {
    if (int rc = moc_check_batch(B, "moc_pack_selected")) return rc;
    MOC_REQUIRE(B->sel_row && B->n_sel && B->cand && !(B->flags & MOC_CAND_FROM_STATS), "moc_pack_selected: batch has no phase-A outputs");
    MOC_REQUIRE(slide0 >= 0 && n >= 1 && slide0 + n <= B->n_slides, "moc_pack_selected: bad slide range");
    MOC_REQUIRE(cap >= 1 && feat_out && cand_out, "moc_pack_selected: bad cap/outputs");
    PackArgs a;
    a.X = (const unsigned char*)B->X; a.row_off = B->row_off; a.sel_row = B->sel_row; a.n_sel = B->n_sel; a.cand = B->cand;
    a.feat_out = (unsigned char*)feat_out; a.cand_out = cand_out; a.stride = B->total_rows;
    a.row_bytes = B->D * moc_elem_size(B->dtype); a.C = B->C; a.cap = cap; a.slide0 = slide0;
    a.packed = 0; a.out_rows = 0;
    pack_selected_kernel<<<dim3(moc_cdiv(cap, 16), n), 256, 0, (hipStream_t)stream>>>(a);
    MOC_CHECK_LAUNCH("moc_pack_selected");
    return MOC_OK;
}

extern "C" int moc_pack_selected_rows(const moc_batch_t* B, int slide0, int n, int cap, void* feat_out, float* cand_out,
                                      int64_t out_rows, moc_stream_t stream) {
    if (int rc = moc_check_batch(B, "moc_pack_selected_rows")) return rc;
    MOC_REQUIRE(B->sel_row && B->n_sel && B->cand && !(B->flags & MOC_CAND_FROM_STATS), "moc_pack_selected_rows: batch has no phase-A outputs");
    MOC_REQUIRE(slide0 >= 0 && n >= 1 && slide0 + n <= B->n_slides, "moc_pack_selected_rows: bad slide range");
    MOC_REQUIRE(cap >= 1 && out_rows >= 1 && feat_out && cand_out, "moc_pack_selected_rows: bad cap/outputs");
    PackArgs a;
    a.X = (const unsigned char*)B->X; a.row_off = B->row_off; a.sel_row = B->sel_row; a.n_sel = B->n_sel; a.cand = B->cand;
    a.feat_out = (unsigned char*)feat_out; a.cand_out = cand_out; a.stride = B->total_rows;
    a.row_bytes = B->D * moc_elem_size(B->dtype); a.C = B->C; a.cap = cap; a.slide0 = slide0;
    a.packed = 1; a.out_rows = out_rows;
    pack_selected_kernel<<<dim3(moc_cdiv(cap, 16), n), 256, 0, (hipStream_t)stream>>>(a);
    MOC_CHECK_LAUNCH("moc_pack_selected_rows");
    return MOC_OK;
}

int moc_launch_topk_mean(const float* keys, int64_t key_stride, const float* vals, int64_t val_stride,
                         const int64_t* seg_off, const int32_t* seg_len, int seg0, int n_seg, int C, int K,
                         int smallest, float* pooled, int32_t* idx_out, int32_t* cnt_out, hipStream_t s) {
    MOC_REQUIRE(keys && vals && seg_off && pooled, "moc_topk_mean: null pointer");
    MOC_REQUIRE(n_seg >= 1 && C >= 1, "moc_topk_mean: bad n_seg=%d C=%d", n_seg, C);
    MOC_REQUIRE(K >= 1 && K <= 4096, "moc_topk_mean: K=%d outside [1, 4096]", K);
    TopkArgs a;
    a.keys = keys; a.vals = vals; a.seg_off = seg_off; a.seg_len = seg_len; a.pooled = pooled;
    a.idx_out = idx_out; a.cnt_out = cnt_out; a.key_stride = key_stride; a.val_stride = val_stride;
    a.C = C; a.K = K; a.smallest = smallest; a.seg0 = seg0;
    int P = 1;
    while (P < K) P <<= 1;
    // thousands of small tasks (an evaluation pass of a wide bank): one wave each, all in flight at once
    static const int wave_min = getenv("MOC_TOPK_WAVE_MIN") ? atoi(getenv("MOC_TOPK_WAVE_MIN")) : 2048;     // (diagnostic knob)
    if (K <= 16 && (int64_t)C * n_seg >= wave_min) {
        const int n_tasks = C * n_seg;
        topk_mean_wave_kernel<<<moc_cdiv(n_tasks, WT_WAVES), 64 * WT_WAVES, 0, s>>>(a, n_tasks);
        MOC_CHECK_LAUNCH("moc_topk_mean(wave)");
        return MOC_OK;
    }
    // 1024 threads: the select passes are chains of dependent key reads, one per blockDim.x keys
    topk_mean_kernel<<<dim3(C, n_seg), 1024, (size_t)P * 12 + 288 * sizeof(int) + (K <= 16 ? 96 + TK_CAND * 8 : 0), s>>>(a);
    MOC_CHECK_LAUNCH("moc_topk_mean");
    return MOC_OK;
}

extern "C" int moc_topk_mean(const float* keys, int64_t key_stride, const float* vals, int64_t val_stride,
                             const int64_t* seg_off, const int32_t* seg_len, int n_seg, int C, int K,
                             int smallest, float* pooled, int32_t* idx_out, int32_t* cnt_out,
                             moc_stream_t stream) {
    return moc_launch_topk_mean(keys, key_stride, vals, val_stride, seg_off, seg_len, 0, n_seg, C, K, smallest,
                                pooled, idx_out, cnt_out, (hipStream_t)stream);
}
