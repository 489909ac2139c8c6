// Phase A, part 1: classifier-bank preparation, row-mask compaction and the
// streaming score pass  X . [W | W_ext[:, C:]]  with its per-row statistics.
//
// Reference semantics: main_moc.py:329-337 (mask, two matmuls), :360-365 and
// utils/patch_selection_classifier_index.py:34, :46-48, :75 (per-row keys).
//
// Kernel shape (gfx950): one wave owns a 16-row tile.  The tile's rows are loaded
// straight into MFMA A-fragment registers (16 B per lane per k-step), the bank
// image sits in LDS in B-fragment order (one conflict-free ds_read_b128 per
// MFMA), products accumulate in fp32:
//   bf16 bags: v_mfma_f32_16x16x32_bf16 against a 3-term bf16 split of the fp32
//              weights (hi+mid+lo carries 24 mantissa bits, every bf16*bf16
//              product is exact in fp32);
//   fp32 bags: v_mfma_f32_16x16x4_f32 (exact fp32 fma chain).
// The 16 x Ct result goes through a per-wave LDS tile so that one lane per row
// can form softmax / top-2 gap / background sum+max, which are stored
// column-major ([stat][slot]) for the column-wise selectors that follow.
#include <stdlib.h>
#include "moc_common.h"
#include <hip/hip_ext.h>
#include <type_traits>

namespace {

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ------------------------------------------------------------------ bank image
__device__ __forceinline__ float bank_elem(const float* W, const float* We, int C, int Ce,
                                           int fg_from_ext, int k, int n) {
    if (n >= Ce) return 0.f;
    if (n < C && !fg_from_ext) return W[(int64_t)k * C + n];
    return We[(int64_t)k * Ce + n];
}

// |w| of every bank element < 2 ?  (fp16 image only: w * 2^14 must stay inside fp16)
__global__ void bank_range_kernel(const float* W, int64_t nW, const float* We, int64_t nWe, int* bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nW && !(fabsf(W[i]) < 2.f)) atomicOr(bad, 1);
    if (i < nWe && !(fabsf(We[i]) < 2.f)) atomicOr(bad, 1);
}

// 16-bit image: [nt][kk][term][lane][8] bf16 / fp16, element j of lane l = Wcat[kk*32 + (l>>4)*8 + j][nt*16 + (l&15)]
template <bool F16>
__global__ void prepare_bank_half_kernel(const float* W, const float* We, int D, int C, int Ce,
                                         int fg_from_ext, uint16_t* out) {
    const int KK = D / 32, NT = (Ce + 15) / 16;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NT * KK * 64) return;
    const int lane = idx & 63, kk = (idx >> 6) % KK, nt = (idx >> 6) / KK;
    const int n = nt * 16 + (lane & 15);
    uint16_t* o = out + ((int64_t)(nt * KK + kk) * 3 * 64 + lane) * 8;
    for (int j = 0; j < 8; ++j) {
        const int k = kk * 32 + (lane >> 4) * 8 + j;
        const float w = bank_elem(W, We, C, Ce, fg_from_ext, k, n);
        uint16_t hi, mid, lo;
        moc_split3<F16>(w, MOC_F16_BANK_SCALE, hi, mid, lo);
        o[j] = hi;
        o[64 * 8 + j] = mid;
        o[2 * 64 * 8 + j] = lo;
    }
}

// f32 image: [nt][kq][lane][4] float, element m of lane l = Wcat[kq*16 + (l>>4)*4 + m][nt*16 + (l&15)]
__global__ void prepare_bank_f32_kernel(const float* W, const float* We, int D, int C, int Ce,
                                        int fg_from_ext, float* out) {
    const int KQ = D / 16, NT = (Ce + 15) / 16;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NT * KQ * 64) return;
    const int lane = idx & 63, kq = (idx >> 6) % KQ, nt = (idx >> 6) / KQ;
    const int n = nt * 16 + (lane & 15);
    float* o = out + (int64_t)idx * 4;
    for (int m = 0; m < 4; ++m)
        o[m] = bank_elem(W, We, C, Ce, fg_from_ext, kq * 16 + (lane >> 4) * 4 + m, n);
}

// ------------------------------------------------------------------ mask -> kept
// One workgroup per slide; a thread owns 16 consecutive rows: its 16 flag bytes are requested at once (the
// flags may sit in pinned host memory: one PCIe round trip per 16,384 rows instead of one per 1,024), one
// block scan of the per-thread counts places them.
__global__ __launch_bounds__(1024) void mask_compact_kernel(const uint8_t* mask, const int64_t* row_off,
                                                            int32_t* kept, int32_t* n_kept) {
    __shared__ int wave_tot[17];
    const int b = blockIdx.x;
    const int64_t base = row_off[b];
    const int n = (int)(row_off[b + 1] - base);
    int running = 0;
    for (int c0 = 0; c0 < n; c0 += 16 * 1024) {
        const int i0 = c0 + (int)threadIdx.x * 16;
        unsigned bits = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = i0 + q;
            const uint8_t f = mask[base + (i < n ? i : n - 1)];
            bits |= (i < n && f != 0 ? 1u : 0u) << q;
        }
        int tot;
        int pos = moc_block_count_scan(__popc(bits), wave_tot, &tot);
        int32_t* out = kept + base + running + pos;
        while (bits) {                                   // ascending: lowest set bit first
            const int q = __ffs((int)bits) - 1;
            *out++ = i0 + q;
            bits &= bits - 1;
        }
        running += tot;
    }
    if (threadIdx.x == 0) n_kept[b] = running;
}

// ------------------------------------------------------------------ score pass
struct ScoresArgs {
    const unsigned char* X;
    const unsigned char* bank;
    const int64_t* row_off;
    const int64_t* x_off;    // nullable
    const int32_t* kept;     // nullable
    const int32_t* n_kept;   // valid iff kept != nullptr
    float* stats;
    uint8_t* sel_flag;
    int64_t stride;          // total_rows
    int D, C, Ce, NT;
    int tpw;                 // 16-row tiles per wave (1 when NT > 1)
    float oscale;            // products -> logits: 1, or 2^-14 for the scaled fp16 image
    int compact;             // MOC_STATS_COMPACT: logits[C] | m1 | 1/den | gap | bg_sum | bg_max (no softmax columns)
    const uint32_t* cu_reserved;   // nullable: compute units the streaming form stays off (moc_batch_t.cu_reserved)
    int32_t* ticket;               // nullable: the streaming form's tile counter (moc_batch_t.tile_ticket), zero at launch
};

// one lane per row: reads the 16 x Ctp tile the wave just wrote, emits the statistics
__device__ __forceinline__ void row_epilogue(const ScoresArgs& a, const float* tile, int ldt,
                                             int64_t slot_base, int slot, bool valid) {
    if (!valid) return;
    const int C = a.C, Ce = a.Ce;
    const float* r = tile;
    float m1 = -INFINITY, m2 = -INFINITY;
    for (int c = 0; c < C; ++c) {
        const float v = r[c];
        if (v > m1) { m2 = m1; m1 = v; } else if (v > m2) { m2 = v; }
    }
    float bsum = 0.f, bmax = -INFINITY;
    for (int c = C; c < Ce; ++c) { const float v = r[c]; bsum += v; bmax = fmaxf(bmax, v); }
    float* s = a.stats + slot_base + slot;
    if (a.compact) {
        // the consumers re-form softmax[c] = exp2((v - m1) log2 e) * (1 / den): den is summed from those very terms
        float den = 0.f;
        for (int c = 0; c < C; ++c) den += __builtin_amdgcn_exp2f((r[c] - m1) * 1.44269504088896340736f);
        for (int c = 0; c < C; ++c) s[(int64_t)c * a.stride] = r[c];
        s[(int64_t)C * a.stride] = m1;
        s[(int64_t)(C + 1) * a.stride] = 1.f / den;
        s[(int64_t)(C + 2) * a.stride] = fabsf(m1 - m2);
        s[(int64_t)(C + 3) * a.stride] = bsum;
        s[(int64_t)(C + 4) * a.stride] = bmax;
        a.sel_flag[slot_base + slot] = 0;
        return;
    }
    float den = 0.f;
    for (int c = 0; c < C; ++c) den += expf(r[c] - m1);
    for (int c = 0; c < C; ++c) {
        const float v = r[c];
        s[(int64_t)c * a.stride] = v;
        s[(int64_t)(C + c) * a.stride] = expf(v - m1) / den;
    }
    s[(int64_t)(2 * C) * a.stride] = fabsf(m1 - m2);
    s[(int64_t)(2 * C + 1) * a.stride] = bsum;
    s[(int64_t)(2 * C + 2) * a.stride] = bmax;
    a.sel_flag[slot_base + slot] = 0;
    (void)ldt;
}

// The two partners of a lane under xor 16 / xor 32 without LDS (ds_bpermute) or its lgkmcnt: the gfx950 row swaps.
// permlane16_swap(x, x) leaves (x[row 0], x[row 1]) in BOTH rows 0 and 1 of the pair (rows 2, 3 alike), permlane32_swap
// the two halves: every lane of a pair then holds the same (lo, hi), so a commutative merge gives both the same bits.
template <int OFF>
__device__ __forceinline__ void xor_pair(float x, float& lo, float& hi) {
    static_assert(OFF == 16 || OFF == 32, "row swaps");
    const unsigned xi = __float_as_uint(x);
    if constexpr (OFF == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(xi, xi, false, false);
        lo = __uint_as_float(r[0]); hi = __uint_as_float(r[1]);
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
        lo = __uint_as_float(r[0]); hi = __uint_as_float(r[1]);
    }
}

// The statistics of one row from its Q*4 column values spread over the row's four lanes (lane = (part, row), part owning
// columns part, part + 4, ...): class columns q < qc, extension columns qc <= q < qe of this lane.  The wave spends more
// cycles here than in its MFMAs when it is alone on its SIMD (three n-tiles: 4,400 of 9,800 cycles per tile before this
// form, scripts/diag_score_phases.py), so the form is branch-free and short: masked values instead of predicated code,
// top-2 by min / max, the cross-lane merges by row swaps, exp as v_exp_f32 of (v - max) log2(e) -- exactly 1 at the
// maximum, relative error 2e-7 elsewhere, no denormal tail -- and ONE division per row (p = e * (1 / den)).
template <int Q>
__device__ __forceinline__ void row_stats_emit(const ScoresArgs& a, const float (&v)[Q], int64_t slot_base, int row0, int nk,
                                               const float* tile, int ldt) {
    const int lane = threadIdx.x & 63, row = lane & 15, part = lane >> 4;
    const int C = a.C, Ce = a.Ce;
    const int qc = (C - part + 3) >> 2, qe = (Ce - part + 3) >> 2;          // c = 4 q + part < C  <=>  q < qc
    float vm[Q];
    float m1 = -INFINITY, m2 = -INFINITY, bsum = 0.f, bmax = -INFINITY;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const bool cls = q < qc, ext = q >= qc && q < qe;
        vm[q] = cls ? v[q] : -INFINITY;
        m2 = fmaxf(m2, fminf(m1, vm[q]));                  // duplicates of the maximum count (gap 0), as topk(2)
        m1 = fmaxf(m1, vm[q]);
        bsum += ext ? v[q] : 0.f;
        bmax = fmaxf(bmax, ext ? v[q] : -INFINITY);
    }
    auto merge = [&](auto off) {
        constexpr int OFF = decltype(off)::value;
        float a1, b1, a2, b2, s0, s1, x0, x1;
        xor_pair<OFF>(m1, a1, b1); xor_pair<OFF>(m2, a2, b2); xor_pair<OFF>(bsum, s0, s1); xor_pair<OFF>(bmax, x0, x1);
        m2 = fmaxf(fminf(a1, b1), fmaxf(a2, b2));
        m1 = fmaxf(a1, b1);
        bsum = s0 + s1;
        bmax = fmaxf(x0, x1);
    };
    merge(std::integral_constant<int, 16>{});
    merge(std::integral_constant<int, 32>{});
    float e[Q], den = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        e[q] = __builtin_amdgcn_exp2f((vm[q] - m1) * 1.44269504088896340736f);       // exp2(-inf) = 0: the masked columns
        den += e[q];
    }
    {
        float d0, d1;
        xor_pair<16>(den, d0, d1); den = d0 + d1;
        xor_pair<32>(den, d0, d1); den = d0 + d1;
    }
    const bool own_row = (unsigned)(row0 + row) < (unsigned)nk;  // else: beyond the slide (or, in a short first tile, before it)
    if (!own_row && !a.compact) return;                          // (the compact form's 16-byte stores use another lane mapping)
#ifdef MOC_STAMPS
    if (a.tpw == 7) {                                  // diagnostic (MOC_EPI_MODE=7): everything computed, nothing stored
#pragma unroll
        for (int q = 0; q < Q; ++q) { const float pq = e[q] / den; asm volatile("" ::"v"(pq), "v"(v[q])); }
        asm volatile("" ::"v"(m1), "v"(m2), "v"(bsum), "v"(bmax));
        return;
    }
#endif
    const float rden = 1.f / den;
    // column c of the statistics starts at stats + c * stride: a uniform base per q, one 32-bit lane offset for all
    const int64_t stride = a.stride;
    float* sv = a.stats + slot_base;
    float* sp = sv + (int64_t)C * stride;
    const unsigned off = (unsigned)((int64_t)part * stride + (row0 + row));
    if (a.compact) {
        // C + 5 columns: the logits, then m1 | 1/den | gap | bg_sum | bg_max.  The softmax columns are what the selector
        // and the candidate gather re-form from (v, m1, 1/den) with exactly the arithmetic above: e * rden.
        if (own_row) {
            float* st = a.stats + slot_base + (int64_t)C * stride + (row0 + row);
            if (part == 0) { st[0] = m1; st[2 * stride] = fabsf(m1 - m2); }
            else if (part == 1) { st[stride] = rden; st[3 * stride] = bsum; }
            else if (part == 2) st[4 * stride] = bmax;
            else a.sel_flag[slot_base + row0 + row] = 0;
        }
        // The logits leave as 16-byte stores: lane -> column 16 p + (lane >> 2), rows 4 (lane & 3) .. + 3 of the tile, read
        // back from the wave's LDS tile (where they are the v[] above).  One store instruction carries 16 columns x 64 B
        // instead of 4: the tile's C columns take ceil(C / 16) instructions, not ceil(C / 4) -- the wave was held by its
        // store issue (round 2: 2,000 of a tile's 8,000 cycles at thirty classes).  Tiles are cut at absolute multiples
        // of 16 slots, so a whole quad is 16-byte aligned; quads that straddle the slide's ends go row by row.
        const int rg = lane & 3, cq = lane >> 2;
        const int r_lo = row0 + rg * 4;
        const bool whole = r_lo >= 0 && r_lo + 3 < nk;
        for (int c0 = 0; c0 < C; c0 += 16) {
            const int c = c0 + cq;
            if (c >= C) continue;
            const float* tp = tile + (rg * 4) * ldt + c;
            const float x0 = tp[0], x1 = tp[ldt], x2 = tp[2 * ldt], x3 = tp[3 * ldt];
            float* dst = a.stats + slot_base + (int64_t)c * stride + r_lo;
            if (whole) {
                // (4-byte aligned in general -- a column starts at c * total_rows floats -- and 16-byte aligned whenever the
                // batch's total is a multiple of four: global_store_dwordx4 takes either)
                typedef float f32x4u_t __attribute__((ext_vector_type(4), aligned(4)));
                f32x4u_t pk = {x0, x1, x2, x3};
                *reinterpret_cast<f32x4u_t*>(dst) = pk;
            } else {
                if ((unsigned)(r_lo + 0) < (unsigned)nk) dst[0] = x0;
                if ((unsigned)(r_lo + 1) < (unsigned)nk) dst[1] = x1;
                if ((unsigned)(r_lo + 2) < (unsigned)nk) dst[2] = x2;
                if ((unsigned)(r_lo + 3) < (unsigned)nk) dst[3] = x3;
            }
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        if (q < qc) {
            sv[off] = v[q];
            sp[off] = e[q] * rden;
        }
        sv += 4 * stride;
        sp += 4 * stride;
    }
    if (part < 3) {
        const float t = part == 0 ? fabsf(m1 - m2) : part == 1 ? bsum : bmax;
        a.stats[slot_base + (int64_t)(2 * C + part) * stride + row0 + row] = t;
    } else {
        a.sel_flag[slot_base + row0 + row] = 0;
    }
}

// all 64 lanes: the row's values are read from the wave's 16 x (NT*16) tile in one batch of independent LDS reads
template <int NT>
__device__ __forceinline__ void row_epilogue_wide(const ScoresArgs& a, const float* tile, int64_t slot_base,
                                                  int row0, int nk) {
    constexpr int LDT = NT * 16 + 1, Q = NT * 4;
    const int lane = threadIdx.x & 63, row = lane & 15, part = lane >> 4;
    float v[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) v[q] = tile[row * LDT + q * 4 + part];
    row_stats_emit<Q>(a, v, slot_base, row0, nk, tile, LDT);
}

// CH = elements of K held in registers per chunk (512 or 256).  BF16: bf16 bag.
template <int CH, bool BF16, bool F16 = false>
__global__ __launch_bounds__(256) void scores_kernel(ScoresArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ESZ = BF16 ? 2 : 4;
    constexpr int NFRAG = CH * ESZ / 64;              // 16-B fragments per lane per chunk
    const int D = a.D;
    const int nchunk = D / CH;
    const int b = blockIdx.y;
    const int64_t base = a.row_off[b];
    const int64_t xbase = a.x_off ? a.x_off[b] : base;
    const int n = (int)(a.row_off[b + 1] - base);
    const int nk = a.kept ? a.n_kept[b] : n;
    const int rows_per_wg = 64 * a.tpw;
    const int wg_row0 = blockIdx.x * rows_per_wg;
    if (wg_row0 >= nk) return;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Ctp = a.NT * 16, ldt = Ctp + 1;
    // LDS: [bank image of one n-tile for all of D][4 waves x 16 x ldt floats]
    const int img_bytes = BF16 ? (D / 32) * 3 * 1024 : (D / 16) * 1024;
    uint4* lds_b = reinterpret_cast<uint4*>(smem);
    float* tile = reinterpret_cast<float*>(smem + img_bytes) + wave * 16 * ldt;
    const int64_t row_bytes = (int64_t)D * ESZ;

    for (int nt = 0; nt < a.NT; ++nt) {
        if (nt > 0) __syncthreads();
        {   // stage this n-tile's bank image
            const uint4* src = reinterpret_cast<const uint4*>(a.bank + (int64_t)nt * img_bytes);
            for (int i = threadIdx.x; i < img_bytes / 16; i += 256) lds_b[i] = src[i];
        }
        __syncthreads();
        for (int t = 0; t < a.tpw; ++t) {
            const int row0 = wg_row0 + (wave * a.tpw + t) * 16;
            if (row0 >= nk) break;                                  // wave-uniform
            const int slot = row0 + (lane & 15);
            const int slot_c = slot < nk ? slot : nk - 1;           // clamp: loads stay in bounds
            const int r = a.kept ? a.kept[base + slot_c] : slot_c;
            const unsigned char* rp = a.X + (xbase + r) * row_bytes + (lane >> 4) * 16;
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            for (int ch = 0; ch < nchunk; ++ch) {
                uint4 af[NFRAG];
#pragma unroll
                for (int f = 0; f < NFRAG; ++f)
                    af[f] = *reinterpret_cast<const uint4*>(rp + (int64_t)ch * CH * ESZ + f * 64);
                if constexpr (BF16) {
                    const uint4* bp = lds_b + (ch * NFRAG) * 3 * 64 + lane;
#pragma unroll
                    for (int f = 0; f < NFRAG; ++f) {
#pragma unroll
                        for (int term = 0; term < 3; ++term) {
                            acc = moc_mfma_half<F16>(af[f], bp[(f * 3 + term) * 64], acc);
                        }
                    }
                } else {
                    const uint4* bp = lds_b + (ch * NFRAG) * 64 + lane;
#pragma unroll
                    for (int f = 0; f < NFRAG; ++f) {
                        const uint4 bv = bp[f * 64];
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[f].x), __uint_as_float(bv.x), acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[f].y), __uint_as_float(bv.y), acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[f].z), __uint_as_float(bv.z), acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[f].w), __uint_as_float(bv.w), acc, 0, 0, 0);
                    }
                }
            }
            // C/D layout: acc[i] = out[row (lane>>4)*4 + i][col lane&15]
            wave_lds_sync();   // previous epilogue reads of this tile are done
#pragma unroll
            for (int i = 0; i < 4; ++i) tile[((lane >> 4) * 4 + i) * ldt + nt * 16 + (lane & 15)] = acc[i] * a.oscale;
            wave_lds_sync();
            if (nt == a.NT - 1 && lane < 16)
                row_epilogue(a, tile + lane * ldt, ldt, base, row0 + lane, row0 + lane < nk);
        }
    }
}

// ---- hand-counted vector loads for the streaming kernel --------------------------------
// LDS ordering inside one wave without touching vmcnt (the workgroup-scope fence used above
// emits s_waitcnt vmcnt(0) and would drain the loads in flight).  LDS executes a wave's
// operations in order; only the compiler must not reorder them.
__device__ __forceinline__ void wave_lds_order() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

typedef unsigned __attribute__((ext_vector_type(4))) u32x4_t;   // native vector: a legal "v" asm operand
template <int OFF>
__device__ __forceinline__ void asm_load16(u32x4_t& dst, const unsigned char* p) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(dst) : "v"(p), "n"(OFF) : "memory");
}
template <int F, int NF>
__device__ __forceinline__ void asm_issue(u32x4_t (&buf)[NF], const unsigned char* p) {
    if constexpr (F < NF) {
        asm_load16<F * 64>(buf[F], p);
        asm_issue<F + 1, NF>(buf, p);
    }
}
// wait until at most KEEP vector-memory operations are outstanding, then re-define every register
// of `buf` through an empty asm so that no use of it can be scheduled above the wait
template <int F, int NF>
__device__ __forceinline__ void asm_touch(u32x4_t (&buf)[NF]) {
    if constexpr (F < NF) {
        asm volatile("" : "+v"(buf[F]));
        asm_touch<F + 1, NF>(buf);
    }
}
template <int KEEP, int NF>
__device__ __forceinline__ void asm_wait_keep(u32x4_t (&buf)[NF]) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
    asm_touch<0, NF>(buf);
}

// LDS reads of the B fragments, issued and awaited by hand for the same reason: left to hipcc, every
// ds_read_b128 sinks next to the MFMA that uses it (read, wait, MFMA, 144 times per tile at NT = 3) and
// with one workgroup per CU nothing hides that latency.  LDS returns in order, so after issuing batch
// n+1, "lgkmcnt(size of batch n+1)" means batch n has landed.  No scalar load is outstanding inside
// compute() (locate() consumes its own), so lgkmcnt counts only these reads.
template <int OFF>
__device__ __forceinline__ void asm_lds16(u32x4_t& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int KEEP, int NB>
__device__ __forceinline__ void asm_lds_wait_keep(u32x4_t (&buf)[NB]) {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(KEEP) : "memory");
    asm_touch<0, NB>(buf);
}
// batch of B fragments for A fragment F: PER reads (terms) at F*PER*1024 + t*1024 from each of NT bases
template <int F, int PER, int NT, int I, int NB>
__device__ __forceinline__ void asm_lds_batch(u32x4_t (&Bv)[NB], const unsigned (&base)[NT]) {
    if constexpr (I < PER * NT) {
        constexpr int t = I / NT, nt = I % NT;
        asm_lds16<(F * PER + t) * 1024>(Bv[I], base[nt]);
        asm_lds_batch<F, PER, NT, I + 1, NB>(Bv, base);
    }
}

// two A fragments per step: issue the batch for F+1, wait for batch F, MFMAs of F; issue F+2, wait F+1, MFMAs
template <int F, int NF, int PER, int NT, int NB, typename Mac>
__device__ __forceinline__ void compute_pairs_impl(const u32x4_t (&buf)[NF], u32x4_t (&B0)[NB], u32x4_t (&B1)[NB],
                                                   const unsigned (&base)[NT], Mac& mac) {
    if constexpr (F < NF) {
        asm_lds_batch<F + 1, PER, NT, 0, NB>(B1, base);
        asm_lds_wait_keep<NB, NB>(B0);
        mac(B0, buf[F]);
        if constexpr (F + 2 < NF) {
            asm_lds_batch<F + 2, PER, NT, 0, NB>(B0, base);
            asm_lds_wait_keep<NB, NB>(B1);
        } else {
            asm_lds_wait_keep<0, NB>(B1);
        }
        mac(B1, buf[F + 1]);
        compute_pairs_impl<F + 2, NF, PER, NT, NB>(buf, B0, B1, base, mac);
    }
}

// ---- streaming form: persistent workgroups ------------------------------------------
// (A second load shape was built and measured against this one -- whole-row contiguous loads staged
// through a swizzled LDS tile with the bank image in registers: a tie on large batches, slower on the
// 32-slide masked launch (3.9 vs 5.1 TB/s).  It is gone from the tree; DESIGN.md section 5 has the numbers.)
// The generic kernel above leaves the load schedule to the compiler, which keeps two
// 1-KiB loads in flight per wave (24 KB per CU: ~3 TB/s).  Here every wave walks a flat
// list of 16-row tiles (all slides of the batch), the bank image is staged once per
// workgroup, and the A fragments are double buffered by hand: all NF loads of the NEXT
// unit are issued before the MFMAs of the current one, so each wave keeps >= NF KiB in
// flight.  A unit = NF*64 bytes of each of the tile's 16 rows (NF = 16: 1 KiB).
// NT n-tiles (Ct <= 16*NT) are accumulated side by side from ONE read of the A fragments, the whole
// bank image (NT x img) resident in LDS: NT > 1 leaves room for one workgroup per CU only.
// The launch covers slides [slide0, slide0 + n_slides) (the launcher chunks a batch whose per-slide
// metadata would not fit beside the image).
// diagnostic build (-DMOC_STAMPS): shader cycles wave 0 of workgroup 0 spends in each phase of the walk -- 0: locate,
// 4: issue of the next unit's loads, 1: wait for this unit's loads, 2: LDS reads + MFMAs, 3: row epilogue -- summed over
// its units into g_moc_stamps[40 + phase], units in [46].  Every stamp sits where no counted LDS read is outstanding
// (s_memtime returns through lgkmcnt).  scripts/diag_score_phases.py reads them.
#ifdef MOC_STAMPS
#define MOC_PHASE_DECL() unsigned long long ph_t = 0, ph_acc[5] = {0, 0, 0, 0, 0}, ph_n = 0
#define MOC_PHASE_BEGIN() ph_t = __builtin_amdgcn_s_memtime()
#define MOC_PHASE(id) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph_acc[id] += t_ - ph_t; ph_t = t_; if ((id) == 1) ++ph_n; } while (0)
#define MOC_PHASE_END() do { if (threadIdx.x == 0 && blockIdx.x == 0) { for (int q_ = 0; q_ < 5; ++q_) g_moc_stamps[40 + q_] = ph_acc[q_]; g_moc_stamps[46] = ph_n; g_moc_stamps[47] = NT; } } while (0)
#else
#define MOC_PHASE_DECL() do { } while (0)
#define MOC_PHASE_BEGIN() do { } while (0)
#define MOC_PHASE(id) do { } while (0)
#define MOC_PHASE_END() do { } while (0)
#endif
template <int NF, bool BF16, int NT, bool F16 = false, bool DYN = false>
__global__ __launch_bounds__(256, NT == 1 ? 2 : 1) void scores_stream_kernel(ScoresArgs a, int slide0, int n_slides) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ESZ = BF16 ? 2 : 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // compute units left to the meta-steps of the pass in progress (include/moc_hip.h): a workgroup that finds itself on
    // one ends here, before it has touched anything; the tiles are handed out by ticket, so whoever stays does all the
    // work.  The first eight workgroups stay wherever they are: progress does not hang on the table.
    if (a.cu_reserved != nullptr && blockIdx.x >= 8) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const uint32_t slot = (xcc & 15u) * 256u + ((hw >> 8) & 255u);
        if ((a.cu_reserved[slot >> 5] >> (slot & 31u)) & 1u) return;          // (uniform: a workgroup lives on one CU)
    }
    // ticketed walk: which of its XCD's counters this workgroup draws from -- by its rank among the workgroups of the XCD
    // that STAY (one atomic per workgroup, answered while the image is staged), so that the counters of an XCD have the
    // same number of workgroups whatever the placement did (by blockIdx, a reserved set of 72 CUs left some counters
    // with no workgroup at all: their tiles went through the slow sweep, 158 instead of 120 us)
    // (in the launch's dynamic LDS, behind the per-slide metadata: the 16 bytes of slack the launcher adds)
    int& s_counter = *(reinterpret_cast<int*>(smem + (size_t)NT * (BF16 ? (a.D / 32) * 3 * 1024 : (a.D / 16) * 1024) +
                                                  4 * 16 * (NT * 16 + 1) * sizeof(float)) + 6 * n_slides + 1);
    if constexpr (DYN) {
        if (threadIdx.x == 0) {
            uint32_t xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            const int x = (int)(xcc & 7u);
            const int r = atomicAdd(a.ticket + (MOC_TICKET_QUEUES + x) * MOC_TICKET_STRIDE, 1);
            s_counter = x * (MOC_TICKET_QUEUES / 8) + r % (MOC_TICKET_QUEUES / 8);
        }
    }
    const int64_t row_bytes = (int64_t)a.D * ESZ;
    const int U = (int)(row_bytes / (NF * 64));                  // units per tile
    const int img_bytes = BF16 ? (a.D / 32) * 3 * 1024 : (a.D / 16) * 1024;
    constexpr int LDT = NT * 16 + 1;
    const int img_vec = img_bytes / 16;                          // uint4 per n-tile
    uint4* lds_b = reinterpret_cast<uint4*>(smem);
    float* tile = reinterpret_cast<float*>(smem + (size_t)NT * img_bytes) + wave * 16 * LDT;
    // [n_slides] first slot, [n_slides] first X row, [n_slides + 1] tile prefix, [n_slides] kept rows
    int64_t* s_base = reinterpret_cast<int64_t*>(smem + (size_t)NT * img_bytes + 4 * 16 * LDT * sizeof(float));
    int64_t* s_xbase = s_base + n_slides;
    int* prefix = reinterpret_cast<int*>(s_xbase + n_slides);
    int* s_nk = prefix + n_slides + 1;

    {   // bank image -> LDS, four 16-B loads in flight per thread
        const uint4* src = reinterpret_cast<const uint4*>(a.bank);
        const int nvec = NT * img_vec;
        int i = threadIdx.x;
        for (; i + 3 * 256 < nvec; i += 4 * 256) {
            const uint4 t0 = src[i], t1 = src[i + 256], t2 = src[i + 512], t3 = src[i + 768];
            lds_b[i] = t0; lds_b[i + 256] = t1; lds_b[i + 512] = t2; lds_b[i + 768] = t3;
        }
        for (; i < nvec; i += 256) lds_b[i] = src[i];
    }
    // per-slide metadata in LDS so that locating a tile costs no dependent global loads
    for (int b = threadIdx.x; b < n_slides; b += 256) {
        const int64_t base = a.row_off[slide0 + b];
        s_base[b] = base;
        s_xbase[b] = a.x_off ? a.x_off[slide0 + b] : base;
        s_nk[b] = a.kept ? a.n_kept[slide0 + b] : (int)(a.row_off[slide0 + b + 1] - base);
    }
    __syncthreads();
    if (wave == 0) {   // prefix[b] = tiles of slides < b
        int carry = 0;
        for (int c0 = 0; c0 < n_slides; c0 += 64) {
            const int b = c0 + lane;
            int v = 0;
            // tiles are cut at ABSOLUTE multiples of 16 slots (the first tile of a slide is short when its base is not
            // one): every 64-byte piece of statistics a tile writes is then 64-byte aligned -- 7 % of the launch at
            // three n-tiles, 2.5 % at one, against bases at odd multiples of 8
            if (b < n_slides && s_nk[b] > 0) v = (s_nk[b] + (int)(s_base[b] & 15) + 15) >> 4;
            int inc = v;
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(inc, off, 64);
                if (lane >= off) inc += o;
            }
            if (b < n_slides) prefix[b + 1] = carry + inc;
            carry += __shfl(inc, 63, 64);
        }
        if (lane == 0) prefix[0] = 0;
    }
    __syncthreads();
    const int total = prefix[n_slides];
    const int stride = gridDim.x * 4;

    struct Unit { const unsigned char* p; int64_t base; int row0, nk, kk0; bool last; };
    // kept[] through the scalar cache: one tile's 16 indices are one uniform 64-B read that
    // counts on lgkmcnt, so it never drains the vector loads in flight (constant address space:
    // the array was written by the previous kernel and is read-only here).
    typedef const int32_t __attribute__((address_space(4))) * kept_sptr;
    kept_sptr kept_s = (kept_sptr)(uintptr_t)a.kept;
    // The slide of a tile.  A wave's next tile lies `stride` tiles on -- rarely more than a few slides away -- so the
    // search starts at the slide of the previous tile and reads the next three boundaries at once (one LDS round trip
    // for up to three steps); only the first tile of a wave takes the binary search.  (One binary search per tile --
    // seven dependent LDS reads with nothing to hide them behind when the wave is alone on its SIMD -- was 1,700 of a
    // tile's 7,400 cycles at three n-tiles: scripts/diag_score_phases.py.)
    int cb = -1;                                                // slide of the previous tile (scalar)
    auto locate = [&](int g, int ch, Unit& u) {
        int b;
        if (cb < 0) {
            int lo = 0, hi = n_slides;                         // prefix[lo] <= g < prefix[hi]
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (prefix[mid] <= g) lo = mid; else hi = mid; }
            b = __builtin_amdgcn_readfirstlane(lo);
        } else {
            b = cb;
            for (;;) {                                         // (uniform: every lane reads the same words)
                const int i1 = b + 1 < n_slides ? b + 1 : n_slides, i2 = b + 2 < n_slides ? b + 2 : n_slides;
                const int i3 = b + 3 < n_slides ? b + 3 : n_slides;
                const int p1 = prefix[i1], p2 = prefix[i2], p3 = prefix[i3];
                const int adv = (g >= p1) + (g >= p2) + (g >= p3);                   // prefix is non-decreasing; g < prefix[n_slides]
                b = __builtin_amdgcn_readfirstlane(b + adv);
                if (adv < 3) break;
            }
        }
        cb = b;
        u.base = s_base[b];
        const int64_t xbase = s_xbase[b];
        u.nk = __builtin_amdgcn_readfirstlane(s_nk[b]);
        u.row0 = __builtin_amdgcn_readfirstlane((g - prefix[b]) * 16 - (int)(u.base & 15));     // < 0 in a short first tile
        int sel = lane & 15;
        const int first_valid = u.row0 < 0 ? -u.row0 : 0, last_valid = u.nk - 1 - u.row0;          // the tile holds a row
        sel = sel > first_valid ? sel : first_valid;               // clamp: loads stay in the slide
        sel = sel < last_valid ? sel : last_valid;
        int r = u.row0 + sel;
        if (a.kept) {
            const int idx = __builtin_amdgcn_readfirstlane((int)u.base + u.row0);
            int k[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                k[i] = kept_s[idx + i];
                asm volatile("" : "+s"(k[i]));     // pin to an SGPR: keeps hipcc from folding two of
            }                                      // them into one lane-indexed VECTOR load (vmcnt)
            r = k[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) r = sel == i ? k[i] : r;
        }
        u.p = a.X + (xbase + r) * row_bytes + (lane >> 4) * 16 + (int64_t)ch * NF * 64;
        u.kk0 = ch * NF;
        u.last = ch == U - 1;
    };
    MOC_PHASE_DECL();
    f32x4_t acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](const u32x4_t (&buf)[NF], const Unit& u) {
        // B fragments come from LDS one A fragment AHEAD of the MFMAs that use them (asm_lds_*)
        constexpr int PER = BF16 ? 3 : 1;                       // 16-B B fragments per A fragment and n-tile
        constexpr int NB = PER * NT;
        static_assert(NB <= 15, "lgkmcnt is a 4-bit counter");
        unsigned bbase[NT];                                    // LDS byte address of this unit's B rows, per n-tile
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            bbase[nt] = (unsigned)(uintptr_t)(lds_b + nt * img_vec + u.kk0 * PER * 64 + lane);
        u32x4_t B0[NB], B1[NB];
        auto mac = [&](const u32x4_t (&Bv)[NB], const u32x4_t& A) {
            if constexpr (BF16) {
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)            // independent accumulators back to back
                        acc[nt] = moc_mfma_half<F16>(A, Bv[term * NT + nt], acc[nt]);
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(A[m]), __uint_as_float(Bv[nt][m]),
                                                                       acc[nt], 0, 0, 0);
            }
        };
        asm_lds_batch<0, PER, NT, 0, NB>(B0, bbase);
        compute_pairs_impl<0, NF, PER, NT, NB>(buf, B0, B1, bbase, mac);
        MOC_PHASE(2);
        if (u.last) {
            wave_lds_order();
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) tile[((lane >> 4) * 4 + i) * LDT + nt * 16 + (lane & 15)] = acc[nt][i] * a.oscale;
            wave_lds_order();
            row_epilogue_wide<NT>(a, tile, u.base, u.row0, u.nk);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            MOC_PHASE(3);
        }
    };
    // Flattened (tile, unit) walk, two register buffers.  The tile loads are inline asm so that the
    // compiler's own wait insertion does not see them: it would make compute(cur) wait for ALL of
    // the next unit's loads (vmcnt is in order; at the loop back edge hipcc assumes the worst).
    // Instead: issue the NF loads of the next unit, then wait until only those NF are outstanding
    // -- everything older (the current unit's loads, the previous epilogue's stores) has landed.
    // Tiles by ticket (DYN) or by a static stride per wave.  Tickets: MOC_TICKET_QUEUES counters, 256 bytes apart; counter
    // q hands out the tiles t * Q + q (t = 0, 1, ...), two per ticket.  A wave draws from ONE counter in the pipelined loop
    // (one of its XCD's eight: s_counter above); when that has run out it leaves the loop and takes what any OTHER
    // counter still holds, one tile at a time
    // (the sweep behind the loop: slow, and idle unless an XCD has no workgroup of this launch left) -- so every tile is
    // handed out whatever workgroups exist and wherever they sit.  (ONE counter serialised the whole launch: a returning
    // atomic on one address takes ~16 ns on this part, 15,000 tiles x 16 ns = 2.3x the kernel's own time; eight counters
    // 1.26x, thirty-two or sixty-four 1.07x.)  A wave's
    // state: `g` the tile whose loads are being issued, `g_next` the one after it (known), and tickets in flight.  A
    // request is an atomic with return issued by hand (lane 0 only, exec switched inside the asm) so that the compiler's
    // wait insertion does not see it.
    // (DYN is a template parameter, not a branch: a join of the two walks behind the request made the compiler copy the
    // ticket register while the value was still in flight.)
    constexpr bool dyn = DYN;
    constexpr int NQ = MOC_TICKET_QUEUES, QS = MOC_TICKET_STRIDE;
    static_assert(NQ % 8 == 0 && NQ <= 64, "one ballot over the counters");
    constexpr int DONE = 0x7fffffff;
    int g, g_next = 0, ch = 0;
    int q = 0;                                                    // this wave's counter
    // vmcnt retires IN ORDER, so where the request sits among the loads decides who waits for it.  Every unit issues
    // exactly one ticket operation right BEFORE its NF loads -- the request itself at a tile boundary, a dummy 4-byte load
    // otherwise -- so that every counted wait is vmcnt(NF + 1): it retires the previous unit's loads and the ticket
    // operation before THEM, never the one just issued.  A request made in unit j has therefore landed after unit
    // j + 1's wait and is taken up in unit j + 2 at the earliest: the two halves of the two-buffer loop below each own a
    // ticket register (pend1 / pend2), requested and taken up at that half's tile boundaries.  Rows of 2 or 4 units:
    // all boundaries fall into the second half; 1 or 3 units: they alternate.
    unsigned pend1 = 0, pend2 = 0, dummy1 = 0, dummy2 = 0;
    // a ticket is TPT consecutive slots of its counter; slot t of counter q is tile t * NQ + q
    constexpr int TPT = MOC_TILES_PER_TICKET;
    static_assert(TPT % 2 == 0, "tickets are taken up at every TPT-th tile boundary: always in the same half of the loop");
    int base = 0, sub = 0;
    const unsigned tpt_inc = (unsigned)TPT;
    auto opaque_vgpr = [](int uniform) { int v; asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(uniform)); return v; };
    // ("+v": a ticket lives in ONE register across the loop -- with a fresh output register the compiler copied it into the
    // loop-carried one right behind the request, i.e. before the value had arrived: tests/test_isa_hazards_cpu.py)
    // one asm statement for both kinds (the branch is INSIDE it): with `if (boundary) request(pend) else dummy(d)` in C++
    // the compiler merged the two behind a pointer select and kept the registers in scratch memory -- stored right
    // behind the request, i.e. before the value had arrived
    auto ticket_op = [&](unsigned& pend, unsigned& d, bool boundary) {
        unsigned long long sv;
        asm volatile("s_mov_b64 %2, exec\n\t"
                     "s_mov_b64 exec, 1\n\t"
                     "s_cmp_lg_u32 %5, 0\n\t"
                     "s_cbranch_scc0 1f\n\t"
                     "global_atomic_add %0, %3, %4, off sc0\n\t"
                     "s_branch 2f\n"
                     "1:\n\t"
                     "global_load_dword %1, %6, off\n"
                     "2:\n\t"
                     "s_mov_b64 exec, %2"
                     : "+v"(pend), "+v"(d), "=&s"(sv)
                     : "v"(a.ticket + q * QS), "v"(tpt_inc), "s"(__builtin_amdgcn_readfirstlane((int)boundary)), "v"(a.bank) : "memory", "scc");
    };
    auto tile_of = [&](int t) -> int { return (t < (1 << 24) && t * NQ + q < total) ? t * NQ + q : DONE; };
    if constexpr (dyn) {
        q = __builtin_amdgcn_readfirstlane(s_counter);               // (NQ / 8 counters per XCD; written before the barriers above)
        // the first two tickets in one request (compiler-managed wait: nothing else is in flight yet): the first one's
        // tiles go to g and g_next, the second sits in the ticket register as if it had just landed (in both halves':
        // only the half that takes tickets up -- the first for an odd number of units per tile, else the second --
        // ever looks at its own)
        int t0 = 0;
        if (lane == 0) t0 = atomicAdd(a.ticket + q * QS, 2 * TPT);
        t0 = __builtin_amdgcn_readfirstlane(t0);
        // (the tickets are uniform, but the walk keeps them in VECTOR registers the compiler cannot see through, as the
        // static walk's g = 4 * blockIdx + wave is: with scalar loop control the compiler rotates the two-buffer loop
        // into one body plus register copies -- copies of load destinations still in flight)
        g = opaque_vgpr(tile_of(t0));
        g_next = opaque_vgpr(tile_of(t0 + 1));
        base = t0;
        sub = 1;
        pend1 = pend2 = (unsigned)opaque_vgpr(t0 + TPT);
    } else {
        g = blockIdx.x * 4 + wave;
    }
    auto advance = [&](unsigned& pend, unsigned& dummy) {
        const bool boundary = ++ch == U;
        if (boundary) ch = 0;
        if constexpr (dyn) {
            bool req = false;
            if (boundary) {
                g = g_next;
                if (sub + 1 < TPT) {                              // the ticket in hand has another tile
                    ++sub;
                } else {
                    asm volatile("" : "+v"(pend));                // (landed: see above; no use may move above the last wait)
                    base = __builtin_amdgcn_readfirstlane((int)pend);
                    sub = 0;
                    req = true;
                }
                g_next = opaque_vgpr(tile_of(base + sub));
            }
            ticket_op(pend, dummy, req);                          // (a request also past the end: one operation per unit)
        } else {
            if (boundary) g += stride;
        }
    };
    constexpr int KEEP = NF + (DYN ? 1 : 0);                      // loads of the unit just issued (+ its ticket operation)
    u32x4_t bufA[NF], bufB[NF];
    Unit uA, uB;
    if (g < total) { locate(g, ch, uA); asm_issue<0, NF>(bufA, uA.p); }
    MOC_PHASE_BEGIN();
    while (g < total) {
        advance(pend1, dummy1);
        const bool moreB = g < total;
        if (moreB) { locate(g, ch, uB); MOC_PHASE(0); asm_issue<0, NF>(bufB, uB.p); MOC_PHASE(4); asm_wait_keep<KEEP, NF>(bufA); }
        else asm_wait_keep<0, NF>(bufA);
        MOC_PHASE(1);
        compute(bufA, uA);
        if (!moreB) break;
        advance(pend2, dummy2);
        const bool moreA = g < total;
        if (moreA) { locate(g, ch, uA); MOC_PHASE(0); asm_issue<0, NF>(bufA, uA.p); MOC_PHASE(4); asm_wait_keep<KEEP, NF>(bufB); }
        else asm_wait_keep<0, NF>(bufB);
        MOC_PHASE(1);
        compute(bufB, uB);
    }
    if constexpr (dyn) {
        // ---- the sweep behind the loop: tiles other counters still hold (see above).  One at a time, nothing pipelined;
        // counter values only grow, so a stale read can only make a counter look open that is not -- `dead` remembers those.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long dead = 1ull << q;
        for (;;) {
            int v = 1 << 24;
            if (lane < NQ) v = __hip_atomic_load(a.ticket + lane * QS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool open = lane < NQ && !((dead >> lane) & 1ull) && v < (1 << 24) && v * NQ + lane < total;
            const unsigned long long m = __ballot(open);
            if (m == 0) break;
            const int qq = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
            int t = 0;
            if (lane == 0) t = atomicAdd(a.ticket + qq * QS, 1);
            t = __builtin_amdgcn_readfirstlane(t);
            if (!(t < (1 << 24) && t * NQ + qq < total)) { dead |= 1ull << qq; continue; }
            const int gs = opaque_vgpr(t * NQ + qq);
            cb = -1;
            for (int c2 = 0; c2 < U; ++c2) {
                locate(gs, c2, uA);
                asm_issue<0, NF>(bufA, uA.p);
                asm_wait_keep<0, NF>(bufA);
                compute(bufA, uA);
            }
        }
    }
    MOC_PHASE_END();
}

// ---- wide banks (16-bit storage, 4..8 n-tiles): K-split form --------------------------------------
// The whole image no longer fits in LDS (C = 64, D = 1024: 5 x 96 KiB), and the generic kernel's answer --
// one n-tile at a time, rows re-read per n-tile, image re-staged per 64 rows -- moves 9x the bag's bytes
// through L2.  Here a workgroup owns 256 rows (a wave 4 row tiles) and keeps ALL their accumulators
// (4 x NT tiles per wave, in AGPRs) while the contraction dimension goes by in chunks of 64 columns:
// per chunk the image slice of every n-tile (NT x 6 KiB) and the workgroup's A fragments (32 KiB) arrive
// by LDS-DMA into the other half of a double buffer while the matrix cores work on this half.  Every bag
// row is read once; the image slice is read once per 256 rows and each B fragment feeds 4 MFMAs.  All LDS
// reads of the loop are hand-issued with counted waits (an ordinary LDS read makes hipcc wait vmcnt(0)
// first -- the DMA in flight writes LDS too -- which would serialise DMA and MFMA).
constexpr int WD_R = 4;               // row tiles per wave
// k-steps (of 32 columns) per chunk: up to 6 n-tiles one k-step per chunk and TWO workgroups per CU (62 KiB of LDS
// and at most 246 registers each: one's DMA waits and row epilogues run beside the other's MFMAs, +12 % at
// 5 n-tiles); wider banks need the registers of a whole SIMD and take two k-steps per chunk
constexpr int wd_kc(int NT) { return NT <= 6 ? 1 : 2; }

template <int NT, bool F16>
__global__ __launch_bounds__(256, (NT <= 6 ? 2 : 1)) void scores_wide_kernel(ScoresArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WD_KC = wd_kc(NT);
    constexpr int B_BYTES = NT * WD_KC * 3 * 1024;                  // image slice of one chunk
    constexpr int A_BYTES = 4 * WD_R * WD_KC * 1024;                // 4 waves x R row tiles x KC k-steps x 1 KiB
    constexpr int BUF = B_BYTES + A_BYTES;
    constexpr int LDT = NT * 16 + 1;
    float* tile = reinterpret_cast<float*>(smem) + (threadIdx.x >> 6) * 16 * LDT;      // (the epilogue's tiles reuse the chunk buffers)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int64_t base = a.row_off[b];
    const int64_t xbase = a.x_off ? a.x_off[b] : base;
    const int n = (int)(a.row_off[b + 1] - base);
    const int nk = a.kept ? a.n_kept[b] : n;
    const int wg_row0 = blockIdx.x * (4 * WD_R * 16);
    if (wg_row0 >= nk) return;
    const int64_t row_bytes = (int64_t)a.D * 2;
    const int KK = a.D / 32, nchunk = KK / WD_KC;
    const int img_bytes = KK * 3 * 1024;                            // one n-tile of the image

    const unsigned char* rp[WD_R];                                  // this lane's row in each of the wave's row tiles
#pragma unroll
    for (int r = 0; r < WD_R; ++r) {
        int slot = wg_row0 + (wave * WD_R + r) * 16 + (lane & 15);
        slot = slot < nk ? slot : nk - 1;                           // clamp: loads stay in bounds
        const int row = a.kept ? a.kept[base + slot] : slot;
        rp[r] = a.X + (xbase + row) * row_bytes + (lane >> 4) * 16;
    }
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto issue_chunk = [&](int c, int buf) {
        unsigned char* dst = smem + buf * BUF;
        // image: NT * KC * 3 pieces of 1 KiB, dealt round-robin to the 4 waves
#pragma unroll
        for (int i = 0; i < (NT * WD_KC * 3 + 3) / 4; ++i) {
            const int piece = i * 4 + wave;                          // [nt][kl][term]
            if (piece < NT * WD_KC * 3) {
                const int nt = piece / (WD_KC * 3), rest = piece - nt * (WD_KC * 3);
                __builtin_amdgcn_global_load_lds((gptr_t)(a.bank + (int64_t)nt * img_bytes + ((int64_t)c * WD_KC * 3 + rest) * 1024 + lane * 16),
                                                 (lptr_t)(dst + piece * 1024), 16, 0, 0);
            }
        }
        // A fragments of this wave: [r][kl] pieces of 1 KiB
#pragma unroll
        for (int r = 0; r < WD_R; ++r)
#pragma unroll
            for (int kl = 0; kl < WD_KC; ++kl)
                __builtin_amdgcn_global_load_lds((gptr_t)(rp[r] + ((int64_t)c * WD_KC + kl) * 64),
                                                 (lptr_t)(dst + B_BYTES + ((wave * WD_R + r) * WD_KC + kl) * 1024), 16, 0, 0);
    };

    f32x4_t acc[WD_R * NT];
#pragma unroll
    for (int q = 0; q < WD_R * NT; ++q) acc[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    issue_chunk(0, 0);
    __syncthreads();                                                // (its fence waits for the DMA: vmcnt(0))
    for (int c = 0; c < nchunk; ++c) {
        if (c + 1 < nchunk) issue_chunk(c + 1, (c + 1) & 1);
        const unsigned buf = (unsigned)(uintptr_t)(smem + (c & 1) * BUF);
        const unsigned b_base = buf + lane * 16;
        const unsigned a_base = buf + B_BYTES + wave * WD_R * WD_KC * 1024 + lane * 16;
        // steps s = (kl, term); B fragments of step s+1 and (at term 0) the A fragments of its k-step are
        // requested before the MFMAs of step s
        u32x4_t A[2][WD_R], Bf[2][NT];
#pragma unroll
        for (int r = 0; r < WD_R; ++r) asm_lds16<0>(A[0][r], a_base + (r * WD_KC + 0) * 1024);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) asm_lds16<0>(Bf[0][nt], b_base + ((nt * WD_KC + 0) * 3 + 0) * 1024);
#pragma unroll
        for (int s2 = 0; s2 < WD_KC * 3; ++s2) {
            const int kl = s2 / 3, term = s2 % 3, cur = s2 & 1, nxt = cur ^ 1;
            const int kl_n = (s2 + 1) / 3, term_n = (s2 + 1) % 3;
            int pending = 0;
            if (s2 + 1 < WD_KC * 3) {
                if (term_n == 0) {
#pragma unroll
                    for (int r = 0; r < WD_R; ++r) asm_lds16<0>(A[kl_n & 1][r], a_base + (r * WD_KC + kl_n) * 1024);
                    pending += WD_R;
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) asm_lds16<0>(Bf[nxt][nt], b_base + ((nt * WD_KC + kl_n) * 3 + term_n) * 1024);
                pending += NT;
            }
            // wait for everything older than the `pending` reads just issued
            if (pending == NT + WD_R) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NT + WD_R) : "memory");
            else if (pending == NT) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NT) : "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm_touch<0, NT>(Bf[cur]);
            if (term == 0) asm_touch<0, WD_R>(A[kl & 1]);
#pragma unroll
            for (int r = 0; r < WD_R; ++r)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[r * NT + nt] = moc_mfma_half<F16>(A[kl & 1][r], Bf[cur][nt], acc[r * NT + nt]);
        }
        __syncthreads();                                            // next chunk landed and visible; this half free
    }
    // ---- epilogue: one row tile at a time through the wave's LDS tile
#pragma unroll
    for (int r = 0; r < WD_R; ++r) {
        const int row0 = wg_row0 + (wave * WD_R + r) * 16;
        if (row0 >= nk) break;                                      // wave-uniform
        wave_lds_order();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) tile[((lane >> 4) * 4 + i) * LDT + nt * 16 + (lane & 15)] = acc[r * NT + nt][i] * a.oscale;
        wave_lds_order();
        row_epilogue_wide<NT>(a, tile, base, row0, nk);
    }
}

// ---- wide banks, 4..5 n-tiles: the K-split walk with TWO chunks in flight behind the one in use ---------------------
// The double-buffered kernel above waits, once per 32-column chunk, for a DMA it issued one chunk (~1,000 cycles of
// MFMAs) earlier; two workgroups per CU cover for each other (3.2 TB/s with nothing stored, against 10 TB/s of MFMA
// rate).  Measured +5 % over it at 5 n-tiles (2.94 against 2.81 TB/s on 64 x 50 k x 1024 fp16, 3.07 against 2.86
// unmasked): what is left is neither the look-ahead nor the 64-byte pieces the rows arrive in (a read-only sweep in
// the same order reaches 6.1 TB/s, scripts/native/piece_bench.hip) but the per-chunk bubble and the serial row epilogue
// (DESIGN.md section 12).  Here
//   * the image slices go through a ring of THREE LDS slots (16 KiB each at 5 n-tiles), loaded by LDS-DMA;
//   * the A fragments do not touch LDS at all: every wave loads its own four row tiles' 16-B pieces straight into a
//     ring of three register sets (asm loads the compiler does not see), two chunks ahead (a fourth set spills);
//   * both kinds of request are issued together, P per wave and chunk, so ONE in-order counter covers them:
//       iteration c:  s_waitcnt vmcnt(P)   -- chunk c (image pieces of this wave + its A fragments) has landed
//                     raw s_barrier         -- ... everybody's image pieces too; all waves are done reading slot c-1
//                     request chunk c+2 (image -> the slot chunk c-1 just left, A -> register set (c+2) mod 3)
//                     the MFMAs of chunk c
//   * 48 KiB of LDS and <= 256 registers: two workgroups per CU, as before -- 128 KiB in flight per CU instead of 62.
// No __syncthreads() in the loop (its fence would wait vmcnt(0) and drain the ring); every LDS read is hand-issued.
template <int NT, bool F16>
__global__ __launch_bounds__(256, 2) void scores_wide_ring_kernel(ScoresArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RING = 3;
    constexpr int B_PIECES = NT * 3, B_PER_WAVE = (B_PIECES + 3) / 4;      // image pieces of one chunk, dealt to 4 waves
    constexpr int P = B_PER_WAVE + WD_R;                                    // requests per wave per chunk (uniform)
    constexpr int SLOT = B_PER_WAVE * 4 * 1024;                             // (padded to a whole round of the deal)
    constexpr int LDT = NT * 16 + 1;
    float* tile = reinterpret_cast<float*>(smem) + (threadIdx.x >> 6) * 16 * LDT;      // (the epilogue's tiles reuse the ring)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int64_t base = a.row_off[b];
    const int64_t xbase = a.x_off ? a.x_off[b] : base;
    const int n = (int)(a.row_off[b + 1] - base);
    const int nk = a.kept ? a.n_kept[b] : n;
    const int wg_row0 = blockIdx.x * (4 * WD_R * 16);
    if (wg_row0 >= nk) return;
    const int64_t row_bytes = (int64_t)a.D * 2;
    const int nchunk = a.D / 32;                                    // a multiple of 8 (D is a multiple of 256)
    const int img_bytes = nchunk * 3 * 1024;                        // one n-tile of the image

    const unsigned char* rp[WD_R];                                  // this lane's row in each of the wave's row tiles
#pragma unroll
    for (int r = 0; r < WD_R; ++r) {
        int slot = wg_row0 + (wave * WD_R + r) * 16 + (lane & 15);
        slot = slot < nk ? slot : nk - 1;                           // clamp: loads stay in bounds
        const int row = a.kept ? a.kept[base + slot] : slot;
        rp[r] = a.X + (xbase + row) * row_bytes + (lane >> 4) * 16;
    }
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto issue_image = [&](int c) {
        unsigned char* dst = smem + (c % RING) * SLOT;
        // piece = [nt][term]; slot i*4 + wave of the padded deal -- a slot past the last piece re-loads piece 0 (into its
        // own padding slot), so that EVERY wave issues exactly B_PER_WAVE image DMAs per chunk
#pragma unroll
        for (int i = 0; i < B_PER_WAVE; ++i) {
            const int sl = i * 4 + wave;
            const int piece = sl < B_PIECES ? sl : 0;
            const int nt = piece / 3, term = piece - nt * 3;
            __builtin_amdgcn_global_load_lds((gptr_t)(a.bank + (int64_t)nt * img_bytes + ((int64_t)c * 3 + term) * 1024 + lane * 16),
                                             (lptr_t)(dst + sl * 1024), 16, 0, 0);
        }
    };
    u32x4_t A0[WD_R], A1[WD_R], A2[WD_R];                           // the A fragments of chunks = 0, 1, 2 (mod 3)
    f32x4_t acc[WD_R * NT];
#pragma unroll
    for (int q = 0; q < WD_R * NT; ++q) acc[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int c, const u32x4_t (&A)[WD_R]) {
        const unsigned b_base = (unsigned)(uintptr_t)(smem + (c % RING) * SLOT) + lane * 16;
        // steps = the three terms; B fragments of term t+1 are requested before the MFMAs of term t
        u32x4_t Bf[2][NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) asm_lds16<0>(Bf[0][nt], b_base + (nt * 3 + 0) * 1024);
#pragma unroll
        for (int term = 0; term < 3; ++term) {
            const int cur = term & 1, nxt = cur ^ 1;
            if (term + 1 < 3) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) asm_lds16<0>(Bf[nxt][nt], b_base + (nt * 3 + term + 1) * 1024);
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NT) : "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            asm_touch<0, NT>(Bf[cur]);
#pragma unroll
            for (int r = 0; r < WD_R; ++r)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[r * NT + nt] = moc_mfma_half<F16>(A[r], Bf[cur][nt], acc[r * NT + nt]);
        }
    };
    // chunk c = g + J of the group of three starting at g: its A fragments sit at byte offset J*64 of rp[] (which
    // advances by 192 per group); the request for chunk c+2 uses offset (J+2)*64 of the same base
#define MOC_WIDE_REQ(CH, AREG, OFF)                                                                     \
    do {                                                                                                \
        issue_image(CH);                                                                                \
        asm_load16<OFF>(AREG[0], rp[0]); asm_load16<OFF>(AREG[1], rp[1]);                               \
        asm_load16<OFF>(AREG[2], rp[2]); asm_load16<OFF>(AREG[3], rp[3]);                               \
    } while (0)
#define MOC_WIDE_STEP(J, ACUR, ANEXT2)                                                                  \
    do {                                                                                                \
        const int c = g + J;                                                                            \
        if (c < nchunk) {                                                                               \
            if (c + 1 < nchunk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");                \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                       \
            asm_touch<0, WD_R>(ACUR);                                                                   \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                          \
            __builtin_amdgcn_s_barrier();                                                               \
            if (c + 2 < nchunk) MOC_WIDE_REQ(c + 2, ANEXT2, (J + 2) * 64);                              \
            compute(c, ACUR);                                                                           \
        }                                                                                               \
    } while (0)
    static_assert(WD_R == 4, "MOC_WIDE_REQ spells out four row tiles");
    MOC_WIDE_REQ(0, A0, 0);
    MOC_WIDE_REQ(1, A1, 64);
    for (int g = 0; g < nchunk; g += 3) {
        MOC_WIDE_STEP(0, A0, A2);
        MOC_WIDE_STEP(1, A1, A0);
        MOC_WIDE_STEP(2, A2, A1);
#pragma unroll
        for (int r = 0; r < WD_R; ++r) rp[r] += 192;
    }
#undef MOC_WIDE_STEP
#undef MOC_WIDE_REQ
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                   // the ring is free: the epilogue's tiles reuse it
    // ---- epilogue: one row tile at a time through the wave's LDS tile
#pragma unroll
    for (int r = 0; r < WD_R; ++r) {
        const int row0 = wg_row0 + (wave * WD_R + r) * 16;
        if (row0 >= nk) break;                                      // wave-uniform
        wave_lds_order();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) tile[((lane >> 4) * 4 + i) * LDT + nt * 16 + (lane & 15)] = acc[r * NT + nt][i] * a.oscale;
        wave_lds_order();
        row_epilogue_wide<NT>(a, tile, base, row0, nk);
    }
}

// Row statistics from a given logits matrix [N, Ct] (row-major): same columns as the score
// pass writes.  One thread per row; used by the helpers that take logits, not bags.
__global__ __launch_bounds__(256) void row_stats_kernel(const float* logits, int64_t N, int Ct, int C,
                                                        float* stats) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float* r = logits + i * Ct;
    float m1 = -INFINITY, m2 = -INFINITY;
    for (int c = 0; c < C; ++c) {
        const float v = r[c];
        if (v > m1) { m2 = m1; m1 = v; } else if (v > m2) { m2 = v; }
    }
    float den = 0.f;
    for (int c = 0; c < C; ++c) den += expf(r[c] - m1);
    float bsum = 0.f, bmax = -INFINITY;
    for (int c = C; c < Ct; ++c) { const float v = r[c]; bsum += v; bmax = fmaxf(bmax, v); }
    float* s = stats + i;
    for (int c = 0; c < C; ++c) {
        s[(int64_t)c * N] = r[c];
        s[(int64_t)(C + c) * N] = expf(r[c] - m1) / den;
    }
    s[(int64_t)(2 * C) * N] = fabsf(m1 - m2);
    s[(int64_t)(2 * C + 1) * N] = bsum;
    s[(int64_t)(2 * C + 2) * N] = bmax;
}

// ---- the score pass from a cache (round 4, opt-in) ---------------------------------------------------------------------
// The per-row statistics of a bag depend on nothing but the row and the frozen classifier bank -- not on the epoch, not
// on the mask, not on which other rows share its MFMA tile (every output element is one row's dot products, accumulated in
// a fixed order) -- yet the reference (and moc_scores) recompute them on every visit (main_moc.py:336-337).  With 288 GB of
// HBM the statistics of EVERY row of a resident split are 28 bytes per 2-KiB row: computed once by an unmasked moc_scores,
// kept, and a train pass then only copies the kept rows' statistics into slot order -- the same bits the score pass would
// write, without reading the bags.  NOT what bench.py's `value` measures (its score pass reads the bags every pass).
// grid (ceil(max_rows / 256), n_slides)
__global__ __launch_bounds__(256) void stats_from_cache_kernel(const float* all, int64_t all_stride, float* stats, int64_t stride,
                                                               uint8_t* sel_flag, const int64_t* row_off, const int64_t* x_off,
                                                               const int32_t* kept, const int32_t* n_kept, int NS) {
    const int b = blockIdx.y;
    const int64_t base = row_off[b];
    const int n = (int)(row_off[b + 1] - base);
    const int nk = kept ? n_kept[b] : n;
    const int slot = blockIdx.x * 256 + threadIdx.x;
    if (slot >= nk) return;
    const int r = kept ? kept[base + slot] : slot;
    const int64_t xr = (x_off ? x_off[b] : base) + r;
    for (int c = 0; c < NS; ++c) stats[(int64_t)c * stride + base + slot] = all[(int64_t)c * all_stride + xr];
    sel_flag[base + slot] = 0;
}

}  // namespace

// ------------------------------------------------------------------ host entry points
static int bank_nt(int Ce) { return (Ce + 15) / 16; }

extern "C" size_t moc_bank_bytes(int D, int Ce, int dtype) {
    if (D <= 0 || Ce <= 0) return 0;
    const size_t per_nt = dtype != MOC_F32 ? (size_t)(D / 32) * 3 * 1024 : (size_t)(D / 16) * 1024;
    return per_nt * bank_nt(Ce);
}

extern "C" int moc_prepare_bank(const float* W, const float* W_ext, int D, int C, int Ce, int dtype,
                                int fg_from_ext, void* bank_out, moc_stream_t stream) {
    MOC_REQUIRE(W && W_ext && bank_out, "moc_prepare_bank: null pointer");
    MOC_REQUIRE(dtype == MOC_F32 || dtype == MOC_BF16 || dtype == MOC_F16, "moc_prepare_bank: bad dtype %d", dtype);
    MOC_REQUIRE(C >= 2 && Ce > C, "moc_prepare_bank: need 2 <= C < Ce (logits should have more bg classes), got C=%d Ce=%d", C, Ce);
    MOC_REQUIRE(D > 0 && D % 256 == 0, "moc_prepare_bank: D=%d must be a multiple of 256", D);
    hipStream_t s = (hipStream_t)stream;
    const int NT = bank_nt(Ce);
    if (dtype == MOC_F16) {
        // the fp16 image holds w * 2^14: every |w| must be < 2 (cosine classifiers are unit-norm columns).
        // Checked here, once per bank, with a host round trip.
        int* bad = nullptr;
        int host_bad = 0;
        if (hipMalloc((void**)&bad, sizeof(int)) != hipSuccess) MOC_FAIL(MOC_ELAUNCH, "moc_prepare_bank: hipMalloc failed");
        (void)hipMemsetAsync(bad, 0, sizeof(int), s);
        const int64_t nW = (int64_t)D * C, nWe = (int64_t)D * Ce;
        bank_range_kernel<<<moc_cdiv(nWe, 256), 256, 0, s>>>(W, nW, W_ext, nWe, bad);
        const hipError_t e = hipMemcpyAsync(&host_bad, bad, sizeof(int), hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        (void)hipFree(bad);
        if (e != hipSuccess) MOC_FAIL(MOC_ELAUNCH, "moc_prepare_bank: range check failed: %s", hipGetErrorString(e));
        MOC_REQUIRE(!host_bad, "moc_prepare_bank: fp16 bags need classifier weights with |w| < 2 (unit-norm columns); "
                    "use bf16 or fp32 storage for this bank");
        const int total = NT * (D / 32) * 64;
        prepare_bank_half_kernel<true><<<moc_cdiv(total, 256), 256, 0, s>>>(W, W_ext, D, C, Ce, fg_from_ext, (uint16_t*)bank_out);
    } else if (dtype == MOC_BF16) {
        const int total = NT * (D / 32) * 64;
        prepare_bank_half_kernel<false><<<moc_cdiv(total, 256), 256, 0, s>>>(W, W_ext, D, C, Ce, fg_from_ext, (uint16_t*)bank_out);
    } else {
        const int total = NT * (D / 16) * 64;
        prepare_bank_f32_kernel<<<moc_cdiv(total, 256), 256, 0, s>>>(W, W_ext, D, C, Ce, fg_from_ext, (float*)bank_out);
    }
    MOC_CHECK_LAUNCH("moc_prepare_bank");
    return MOC_OK;
}

int moc_check_batch(const moc_batch_t* B, const char* who) {
    MOC_REQUIRE(B, "%s: null batch", who);
    MOC_REQUIRE(B->X && B->row_off, "%s: null X/row_off", who);
    MOC_REQUIRE(B->dtype == MOC_F32 || B->dtype == MOC_BF16 || B->dtype == MOC_F16, "%s: bad dtype %d", who, B->dtype);
    MOC_REQUIRE(B->D > 0 && B->D % 256 == 0, "%s: D=%d must be a multiple of 256", who, B->D);
    MOC_REQUIRE(((uintptr_t)B->X & 15) == 0, "%s: X must be 16-byte aligned", who);
    MOC_REQUIRE(B->n_slides > 0 && B->total_rows > 0 && B->max_rows > 0 && B->max_rows <= B->total_rows,
                "%s: bad sizes n_slides=%d total_rows=%lld max_rows=%d", who, B->n_slides, (long long)B->total_rows, B->max_rows);
    MOC_REQUIRE(B->total_rows < (1ll << 31), "%s: total_rows must fit int32", who);
    MOC_REQUIRE(B->C >= 2 && B->Ce > B->C && B->Ce <= 256, "%s: need 2 <= C < Ce <= 256 (logits should have more bg classes), got C=%d Ce=%d", who, B->C, B->Ce);
    MOC_REQUIRE(B->topj >= 1 && B->topk >= 1, "%s: topj/topk must be >= 1", who);
    MOC_REQUIRE(!B->mask || (B->kept && B->n_kept), "%s: mask given but kept/n_kept work arrays are null", who);
    return MOC_OK;
}

extern "C" int moc_mask_compact(const moc_batch_t* B, moc_stream_t stream) {
    if (int rc = moc_check_batch(B, "moc_mask_compact")) return rc;
    if (!B->mask) return MOC_OK;
    mask_compact_kernel<<<B->n_slides, 1024, 0, (hipStream_t)stream>>>(B->mask, B->row_off, B->kept, B->n_kept);
    MOC_CHECK_LAUNCH("moc_mask_compact");
    return MOC_OK;
}

// ev0 / ev1 (nullable): HIP events that take the score kernel's OWN start and end time stamps (hipExtLaunchKernel: the
// dispatch's profiling stamps, not the moment a marker packet reaches the queue -- an event pair recorded around the
// launch on a busy GPU measured 59 us for a kernel rocprofv3 times at 47).  A batch launched in chunks: first / last.
// the ticketed form of the streaming kernel (up to three n-tiles; see scores_impl)
template <int NF, bool BF, int NTT, bool FH>
static void launch_stream_ticketed(const ScoresArgs& a, int wgs, size_t smem, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1,
                                   int s0, int ns) {
    if constexpr (NTT < 4) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)scores_stream_kernel<NF, BF, NTT, FH, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
        hipExtLaunchKernelGGL((scores_stream_kernel<NF, BF, NTT, FH, true>), dim3(wgs), dim3(256), smem, s, ev0, ev1, 0, a, s0, ns);
    }
}

#ifndef MOC_LOOKAHEAD_WGS_DEFAULT
#define MOC_LOOKAHEAD_WGS_DEFAULT 0
#endif
static int scores_impl(const moc_batch_t* B, const void* bank, moc_stream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    if (int rc = moc_check_batch(B, "moc_scores")) return rc;
    MOC_REQUIRE(bank && B->stats && B->sel_flag, "moc_scores: null bank/stats/sel_flag");
    ScoresArgs a;
    a.X = (const unsigned char*)B->X;
    a.bank = (const unsigned char*)bank;
    a.row_off = B->row_off;
    a.x_off = B->x_off;
    a.kept = B->mask ? B->kept : nullptr;
    a.n_kept = B->n_kept;
    a.stats = B->stats;
    a.sel_flag = B->sel_flag;
    a.stride = B->total_rows;
    a.D = B->D; a.C = B->C; a.Ce = B->Ce; a.NT = bank_nt(B->Ce);
    a.compact = (B->flags & MOC_STATS_COMPACT) ? 1 : 0;
    a.cu_reserved = nullptr;
    a.ticket = nullptr;
    const bool bf = B->dtype != MOC_F32;          // 16-bit storage (bf16 or fp16): 3-term image, K = 32 per MFMA
    const bool f16 = B->dtype == MOC_F16;
    a.oscale = f16 ? 1.f / MOC_F16_BANK_SCALE : 1.f;
    const size_t img = bf ? (size_t)(B->D / 32) * 3 * 1024 : (size_t)(B->D / 16) * 1024;
    hipStream_t s = (hipStream_t)stream;
    // streaming form: persistent workgroups over the flat tile list, the whole bank image (all NT
    // n-tiles) resident in LDS.  Applies while image + epilogue tiles + at least one slide's metadata fit.
    const size_t fixed = (size_t)a.NT * img + 4 * 16 * (a.NT * 16 + 1) * sizeof(float) + 16;
    // (bf16 stops at 3 n-tiles: with 4, the 12 B fragments per batch leave hipcc short of registers and it
    // parks in-flight load destinations in AGPRs -- tests/test_isa_hazards_cpu.py)
    if (a.NT <= (bf ? 3 : 4) && fixed + 24 <= 160 * 1024) {
        a.tpw = 0;
#ifdef MOC_STAMPS
        if (const char* em = getenv("MOC_EPI_MODE")) a.tpw = atoi(em);     // diagnostic: row_stats_emit
#endif
        const int chunk_max = (int)((160 * 1024 - fixed) / 24);
        const int chunk = B->n_slides < chunk_max ? B->n_slides : chunk_max;
        const size_t smem = fixed + (size_t)chunk * 24;
        MOC_REQUIRE(a.NT > 1 || chunk == B->n_slides, "moc_scores: D=%d / n_slides=%d need %zu B of LDS (> 160 KiB)",
                    B->D, B->n_slides, fixed + (size_t)B->n_slides * 24);
        int64_t tiles = 0;   // upper bound from the host-known sizes
        tiles = (B->total_rows + 15) / 16 + 2 * (int64_t)B->n_slides;
        int wgs = (int)((tiles + 3) / 4);
        const int resident = 256 * (smem <= 80 * 1024 ? 2 : 1);
        if (wgs > resident) wgs = resident;
        // A look-ahead launch (tile_ticket set: it runs beside the meta-steps of the pass before) is kept to fewer
        // persistent workgroups: every wave of this kernel holds up to 32 KiB of loads in flight, 2,048 waves 64 MB --
        // several times what 8 TB/s x the memory latency can use, and every dependent load of a meta-step queues
        // behind them (profiles/NOTES.md round 4: the steps crawl while a full-width score pass streams, whether or
        // not it leaves them compute units).  MOC_LOOKAHEAD_WGS overrides (0: no cap).
        if (B->tile_ticket) {
            static const int cap_env = getenv("MOC_LOOKAHEAD_WGS") ? atoi(getenv("MOC_LOOKAHEAD_WGS")) : MOC_LOOKAHEAD_WGS_DEFAULT;
            if (cap_env > 0 && wgs > cap_env) wgs = cap_env;
        }
        const int row_b = B->D * moc_elem_size(B->dtype);
#define MOC_LAUNCH_STREAM(NF, BF, NTT, FH)                                                              \
        do {                                                                                            \
            static bool attr_set = false;                                                               \
            if (!attr_set) {                                                                            \
                (void)hipFuncSetAttribute((const void*)scores_stream_kernel<NF, BF, NTT, FH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                attr_set = true;                                                                        \
            }                                                                                           \
            if (a.ticket) launch_stream_ticketed<NF, BF, NTT, FH>(a, wgs, smem, s, (s0 == 0 ? ev0 : nullptr),       \
                                                                   (s0 + ns == B->n_slides ? ev1 : nullptr), s0, ns); \
            if (!a.ticket)                                                                                      \
                hipExtLaunchKernelGGL((scores_stream_kernel<NF, BF, NTT, FH, false>), dim3(wgs), dim3(256), smem, s,  \
                                      (s0 == 0 ? ev0 : nullptr), (s0 + ns == B->n_slides ? ev1 : nullptr), 0, a, s0, ns); \
        } while (0)
#define MOC_LAUNCH_STREAM_NT(NF, BF, FH)                                                                \
        do {                                                                                            \
            if (a.NT == 1) MOC_LAUNCH_STREAM(NF, BF, 1, FH);                                            \
            else if (a.NT == 2) MOC_LAUNCH_STREAM(NF, BF, 2, FH);                                       \
            else if (a.NT == 3) MOC_LAUNCH_STREAM(NF, BF, 3, FH);                                       \
        } while (0)
        MOC_REQUIRE(B->cu_reserved == nullptr || B->tile_ticket != nullptr, "moc_scores: cu_reserved needs tile_ticket");
        a.ticket = B->tile_ticket;
        a.cu_reserved = B->cu_reserved;
        // (four n-tiles, fp32 only: the ticketed form leaves the compiler short of registers and it re-uses load
        // destinations still in flight -- tests/test_isa_hazards_cpu.py; that shape keeps the static walk, whole chip)
        if (a.NT == 4) { a.ticket = nullptr; a.cu_reserved = nullptr; }
        for (int s0 = 0; s0 < B->n_slides; s0 += chunk) {
            const int ns = B->n_slides - s0 < chunk ? B->n_slides - s0 : chunk;
            if (a.ticket && hipMemsetAsync(a.ticket, 0, sizeof(int32_t) * (MOC_TICKET_QUEUES + 8) * MOC_TICKET_STRIDE, s) != hipSuccess)
                MOC_FAIL(MOC_ELAUNCH, "moc_scores: clearing the tile counter failed");
            if (row_b % 1024 == 0) {
                if (f16) MOC_LAUNCH_STREAM_NT(16, true, true);
                else if (bf) MOC_LAUNCH_STREAM_NT(16, true, false);
                else if (a.NT == 4) MOC_LAUNCH_STREAM(16, false, 4, false);      // fp32 only: four n-tiles
                else MOC_LAUNCH_STREAM_NT(16, false, false);
            } else {
                if (f16) MOC_LAUNCH_STREAM_NT(8, true, true);
                else if (bf) MOC_LAUNCH_STREAM_NT(8, true, false);
                else if (a.NT == 4) MOC_LAUNCH_STREAM(8, false, 4, false);
                else MOC_LAUNCH_STREAM_NT(8, false, false);
            }
            MOC_CHECK_LAUNCH("moc_scores(stream)");
        }
#undef MOC_LAUNCH_STREAM_NT
#undef MOC_LAUNCH_STREAM
        return MOC_OK;
    }
    // wide banks on 16-bit storage: the K-split form (every bag row read once)
    if (bf && a.NT >= 4 && a.NT <= 8 && B->D % 64 == 0) {
        const size_t kc = wd_kc(a.NT);
        const size_t buf = (size_t)a.NT * kc * 3 * 1024 + (size_t)4 * WD_R * kc * 1024;
        const size_t tiles = (size_t)4 * 16 * (a.NT * 16 + 1) * sizeof(float);
        const size_t smem_w = 2 * buf > tiles ? 2 * buf : tiles;
        dim3 grid_w(moc_cdiv(B->max_rows, 4 * WD_R * 16), B->n_slides);
#define MOC_LAUNCH_WIDE(NTT)                                                                            \
        do {                                                                                            \
            static bool attr_set[2] = {false, false};                                                   \
            if (!attr_set[f16]) {                                                                       \
                if (f16) (void)hipFuncSetAttribute((const void*)scores_wide_kernel<NTT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                else (void)hipFuncSetAttribute((const void*)scores_wide_kernel<NTT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                attr_set[f16] = true;                                                                   \
            }                                                                                           \
            if (f16) hipExtLaunchKernelGGL((scores_wide_kernel<NTT, true>), grid_w, dim3(256), smem_w, s, ev0, ev1, 0, a); \
            else hipExtLaunchKernelGGL((scores_wide_kernel<NTT, false>), grid_w, dim3(256), smem_w, s, ev0, ev1, 0, a);    \
        } while (0)
        if (a.NT <= 5) {                                               // (6 n-tiles: the register ring spills)
            const size_t ring = (size_t)3 * ((size_t)((a.NT * 3 + 3) / 4) * 4 * 1024);         // three image slots
            const size_t smem_r = ring > tiles ? ring : tiles;
#define MOC_LAUNCH_RING(NTT)                                                                            \
            do {                                                                                        \
                static bool attr_set[2] = {false, false};                                               \
                if (!attr_set[f16]) {                                                                   \
                    if (f16) (void)hipFuncSetAttribute((const void*)scores_wide_ring_kernel<NTT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                    else (void)hipFuncSetAttribute((const void*)scores_wide_ring_kernel<NTT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                    attr_set[f16] = true;                                                               \
                }                                                                                       \
                if (f16) hipExtLaunchKernelGGL((scores_wide_ring_kernel<NTT, true>), grid_w, dim3(256), smem_r, s, ev0, ev1, 0, a); \
                else hipExtLaunchKernelGGL((scores_wide_ring_kernel<NTT, false>), grid_w, dim3(256), smem_r, s, ev0, ev1, 0, a);    \
            } while (0)
            if (a.NT == 4) MOC_LAUNCH_RING(4);
            else MOC_LAUNCH_RING(5);
#undef MOC_LAUNCH_RING
            MOC_CHECK_LAUNCH("moc_scores(wide ring)");
            return MOC_OK;
        }
        if (smem_w < 160 * 1024) {
            switch (a.NT) {
                case 4: MOC_LAUNCH_WIDE(4); break;
                case 5: MOC_LAUNCH_WIDE(5); break;
                case 6: MOC_LAUNCH_WIDE(6); break;
                case 7: MOC_LAUNCH_WIDE(7); break;
                default: MOC_LAUNCH_WIDE(8); break;
            }
            MOC_CHECK_LAUNCH("moc_scores(wide)");
            return MOC_OK;
        }
#undef MOC_LAUNCH_WIDE
    }
    a.tpw = 1;
    const size_t smem = img + (size_t)4 * 16 * (a.NT * 16 + 1) * sizeof(float);
    MOC_REQUIRE(smem <= 160 * 1024, "moc_scores: D=%d needs %zu B of LDS (> 160 KiB)", B->D, smem);
    dim3 grid(moc_cdiv(B->max_rows, 64 * a.tpw), B->n_slides), block(256);
#define MOC_LAUNCH_SCORES(CH, BF, FH)                                                                   \
    do {                                                                                                \
        static bool attr_set = false;                                                                   \
        if (!attr_set) {                                                                                \
            (void)hipFuncSetAttribute((const void*)scores_kernel<CH, BF, FH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_set = true;                                                                            \
        }                                                                                               \
        hipExtLaunchKernelGGL((scores_kernel<CH, BF, FH>), grid, block, smem, s, ev0, ev1, 0, a);       \
    } while (0)
    if (B->D % 512 == 0) {
        if (f16) MOC_LAUNCH_SCORES(512, true, true);
        else if (bf) MOC_LAUNCH_SCORES(512, true, false);
        else MOC_LAUNCH_SCORES(512, false, false);
    } else {
        if (f16) MOC_LAUNCH_SCORES(256, true, true);
        else if (bf) MOC_LAUNCH_SCORES(256, true, false);
        else MOC_LAUNCH_SCORES(256, false, false);
    }
#undef MOC_LAUNCH_SCORES
    MOC_CHECK_LAUNCH("moc_scores");
    return MOC_OK;
}

extern "C" int moc_scores(const moc_batch_t* B, const void* bank, moc_stream_t stream) {
    return scores_impl(B, bank, stream, nullptr, nullptr);
}

extern "C" int moc_scores_timed(const moc_batch_t* B, const void* bank, moc_stream_t stream, void* start_event, void* stop_event) {
    MOC_REQUIRE(start_event && stop_event, "moc_scores_timed: null event");
    return scores_impl(B, bank, stream, (hipEvent_t)start_event, (hipEvent_t)stop_event);
}

extern "C" int moc_scores_from_cache(const moc_batch_t* B, const float* stats_all, int64_t all_rows, moc_stream_t stream) {
    if (int rc = moc_check_batch(B, "moc_scores_from_cache")) return rc;
    MOC_REQUIRE(stats_all && B->stats && B->sel_flag && all_rows > 0, "moc_scores_from_cache: null stats / cache");
    MOC_REQUIRE(!B->mask || (B->kept && B->n_kept), "moc_scores_from_cache: a masked batch needs its kept-row lists (moc_mask_compact)");
    const int NS = (B->flags & MOC_STATS_COMPACT) ? B->C + 5 : 2 * B->C + 3;
    dim3 grid(moc_cdiv(B->max_rows, 256), B->n_slides);
    stats_from_cache_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(stats_all, all_rows, B->stats, B->total_rows, B->sel_flag, B->row_off,
                                                                   B->x_off, B->mask ? B->kept : nullptr, B->n_kept, NS);
    MOC_CHECK_LAUNCH("moc_scores_from_cache");
    return MOC_OK;
}

extern "C" int moc_row_stats(const float* logits, int64_t N, int Ct, int C, float* stats, moc_stream_t stream) {
    MOC_REQUIRE(logits && stats, "moc_row_stats: null pointer");
    MOC_REQUIRE(N >= 1 && C >= 1 && Ct >= C, "moc_row_stats: bad shape N=%lld Ct=%d C=%d", (long long)N, Ct, C);
    row_stats_kernel<<<moc_cdiv(N, 256), 256, 0, (hipStream_t)stream>>>(logits, N, Ct, C, stats);
    MOC_CHECK_LAUNCH("moc_row_stats");
    return MOC_OK;
}
