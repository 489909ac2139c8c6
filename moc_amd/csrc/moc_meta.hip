// Phase B: the meta-learner ("senet") over the selected rows, the gated mix,
// top-K mean pooling, cross entropy, the analytically sparse backward and Adam.
//
// Reference semantics:
//   senet ................ main_moc.py:299-312  (Linear D->64, ReLU, Linear 64->4, Sigmoid)
//   gated mix ............ main_moc.py:391-403 (train), :482-492 (eval)
//   pooling .............. utils/patch_selection_classifier.py:18-32
//   loss / backward ...... main_moc.py:406-409 (F.cross_entropy, loss.backward())
//   Adam ................. main_moc.py:316, :410 (lr 1e-3, weight_decay 1e-4 coupled, torch defaults)
//
// Only the <= K*C rows that reach the top-K of some class carry gradient, so the
// backward never touches the other S-K*C rows: d mixed[s,c] = dpooled[c]/k for
// those (s,c) "pairs", then the usual chain through sigmoid, Linear, ReLU, Linear.
//
// Per meta-step four launches on one stream, no host synchronisation:
//   meta_forward (S/16 workgroups, f32 MFMA) -> topk_mean (C workgroups)
//   -> finish (1 workgroup: CE, pair gradients, Adam on b1/W2/b2)
//   -> w1_update (W1 gradient from <= K*C gathered rows + Adam, one thread per element)
#include <cstdlib>
#include <cstring>
#include <new>
#include "moc_common.h"
#include <type_traits>
#include "moc_p2p.h"

int moc_check_batch(const moc_batch_t* B, const char* who);
int moc_launch_topk_mean(const float* keys, int64_t key_stride, const float* vals, int64_t val_stride,
                         const int64_t* seg_off, const int32_t* seg_len, int seg0, int n_seg, int C, int K,
                         int smallest, float* pooled, int32_t* idx_out, int32_t* cnt_out, hipStream_t s);

namespace {

constexpr int H = MOC_HIDDEN;

// Scalars exactly as torch hands them to its fp32 kernels: computed in Python doubles,
// rounded to fp32 once.
struct AdamCoef {
    float wd, one_minus_b1, beta2, one_minus_b2, eps;
    float neg_step_size;   // -(lr / (1 - beta1^t))
    float bc2_sqrt;        // sqrt(1 - beta2^t)
    float grad_scale;
};

// torch.optim.Adam (single-tensor path, coupled L2), operation by operation:
//   g = g + wd*p ; m.lerp_(g, 1-b1) ; v.mul_(b2).addcmul_(g, g, 1-b2)
//   denom = sqrt(v)/bc2_sqrt + eps ; p.addcdiv_(m, denom, -step_size)
__device__ __forceinline__ void adam_update(float& p, float& m, float& v, float g, const AdamCoef& k) {
    g = moc_fadd(g, moc_fmul(k.wd, p));
    m = moc_fadd(m, moc_fmul(k.one_minus_b1, moc_fsub(g, m)));
    v = moc_fadd(moc_fmul(v, k.beta2), moc_fmul(moc_fmul(k.one_minus_b2, g), g));
    const float denom = moc_fadd(moc_fdiv(moc_fsqrt(v), k.bc2_sqrt), k.eps);
    p = moc_fadd(p, moc_fdiv(moc_fmul(k.neg_step_size, m), denom));
}

// ------------------------------------------------------------------ tile records (round 4)
// The step kernel used to request and rank ALL S x C mixed scores of the slide the forward had just written -- a memory
// round trip of sixteen loads per thread, then the candidate search over them on sixteen waves of one CU.  The forward's
// workgroups now leave, per 16-row tile and class, the tile's TILE_R largest scores as RECORDS -- the key (score | ~row),
// the row's id in the bag, its four gates and its four candidate scores: everything the backward needs of a pooled row
// except its hidden activations and the row itself -- plus rho = the key of the tile's (TILE_R + 1)-th largest score.
// The step kernel reads a class's ceil(S / 16) x TILE_R keys with ONE wave, finds a lower bound T0 of the K-th largest
// score (the K-th largest of sixteen group maxima: group g = the tiles g, g + 16, ...) and pools among the records >= T0.
// That is EXACT whenever every score >= T0 is a record, i.e. when no tile holds more than TILE_R of them: rho < T0 for
// every tile -- checked; otherwise (and for more than 64 candidates) the kernel falls back to the full scores, which the
// forward still writes.  Same rows, same (value desc, row asc) order, same sum: same bits.
constexpr int TILE_R = 4;
constexpr int MOC_MAX_RUNS = 16;        // meta-learners one launch can serve (moc_train_steps_runs)
constexpr int MOC_TILE_PATH_RECORDS = 1000001, MOC_TILE_PATH_FULL = 1000002;   // left in ws->n_pair[0] by the tile-record step
struct TileWs {
    float4* lam;                 // [slots] gates of the record's row
    float4* sc;                  // [slots] its candidate scores s_p[c], s_sigma[c], s_delta, s_beta
    unsigned long long* key;     // [slots] (moc_key_desc(score) << 32) | ~row; high word 0: empty
    int64_t* rid;                // [slots] sel_row of the row
    uint32_t* rho;               // [slots / TILE_R] high word of the (TILE_R + 1)-th largest key of the (tile, class); 0: none
};
// Slide b owns the slots from slot0 = ((row_off[b] >> 4) + b) * C * TILE_R on: C * cap * TILE_R of them, cap = ceil(rows of
// the slide / 16), class-major: record r of (tile x, class c) is slot0 + (c * cap + x) * TILE_R + r (a class's records are
// contiguous: one wave reads them); rho of (x, c) is entry slot0 / TILE_R + c * cap + x.
__host__ __device__ inline int64_t tile_slots(int64_t total_rows, int n_slides, int C) {
    return ((total_rows >> 4) + n_slides + 2) * (int64_t)C * TILE_R;
}
__host__ __device__ inline TileWs tile_carve(void* p, int64_t ns) {
    TileWs T;
    unsigned char* q = (unsigned char*)p;
    T.lam = (float4*)q; q += ns * 16;
    T.sc = (float4*)q; q += ns * 16;
    T.key = (unsigned long long*)q; q += ns * 8;
    T.rid = (int64_t*)q; q += ns * 8;
    T.rho = (uint32_t*)q;
    return T;
}
__host__ __device__ inline size_t tile_bytes(int64_t ns) { return (size_t)ns * 48 + (size_t)(ns / TILE_R) * 4 + 16; }

// Epilogue of a forward workgroup: thread t = class * 16 + row holds the mixed score v of (row0 + row, class); the 16 rows
// of a class are 16 consecutive lanes (one DPP row).  Every lane ranks its key among the row's sixteen by fifteen row
// rotations (keys are unique: absent rows carry (0 | ~row)), ranks 0 .. TILE_R-1 write their record, rank TILE_R writes rho.
__device__ __forceinline__ void tile_emit(const TileWs& T, int64_t tc, bool ok, float v, int row, int64_t rid, float4 lam, float4 sc) {
    const unsigned hi = ok ? moc_key_desc(v) : 0u, lo = ~(unsigned)row;
    int rank = 0;
#define MOC_ROR(n)                                                                                              \
    {                                                                                                           \
        const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, 0x120 + n, 0xf, 0xf, false);     \
        const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0x120 + n, 0xf, 0xf, false);     \
        rank += (ohi > hi || (ohi == hi && olo > lo)) ? 1 : 0;                                                  \
    }
    MOC_ROR(1) MOC_ROR(2) MOC_ROR(3) MOC_ROR(4) MOC_ROR(5) MOC_ROR(6) MOC_ROR(7) MOC_ROR(8)
    MOC_ROR(9) MOC_ROR(10) MOC_ROR(11) MOC_ROR(12) MOC_ROR(13) MOC_ROR(14) MOC_ROR(15)
#undef MOC_ROR
    if (rank < TILE_R) {
        const int64_t slot = tc * TILE_R + rank;
        T.key[slot] = ((unsigned long long)hi << 32) | lo;
        T.rid[slot] = rid;
        T.lam[slot] = lam;
        T.sc[slot] = sc;
    } else if (rank == TILE_R) {
        T.rho[tc] = hi;
    }
}

// ------------------------------------------------------------------ forward
struct FwdArgs {
    const unsigned char* X;
    const int64_t* row_off;
    const int64_t* sel_row;
    const int32_t* n_sel;
    const float* cand;
    const float *W1, *b1, *W2, *b2;
    const unsigned char* W1img;     // W1 in MFMA B-operand order (see w1_image_* below)
    float *H1, *gates, *mixed;
    int64_t stride;
    int64_t base_host;              // >= 0: first slot of the (single) slide, known to the host
    int D, C, slide0;
    uint32_t use_bits;
    // where the four candidate scores of a selected row come from (main_moc.py:359-366).  0: the materialised `cand`
    // columns (moc_gather_candidates).  1 / 2: straight from the score pass's statistics (full / compact layout) through
    // sel_idx -- MOC_CAND_FROM_STATS: evaluation passes, which then never write or read the [2C+2, S] candidate array
    // (at thirty classes 545 MB written and read back per 202 slides).  Same values, same bits.
    int cand_mode;
    const float* stats;
    const int32_t* sel_idx;
    // tile records for the step kernel (training step of ONE slide, base_host >= 0): tile_on != 0
    TileWs tile;
    int tile_on, tile_cap;          // tile_cap = ceil(rows of the slide / 16)
    int64_t tile_slot0;             // first slot of the slide's region
    // batched runs (round 4; n_runs > 0): ONE launch serves n_runs independent meta-learners, grid.y = run.  Run r works on
    // slide slide0 + r * slide_stride with its own parameters, par_stride floats (W2: w2_stride; operand image: img_stride
    // bytes) behind run 0's; the per-run scalars the host knows come as arrays (kernel arguments, indexed by the run)
    int n_runs, slide_stride;
    int64_t par_stride, w2_stride, img_stride;
    int64_t base_r[MOC_MAX_RUNS], tile_slot0_r[MOC_MAX_RUNS];
    int32_t tile_cap_r[MOC_MAX_RUNS];
    // the slide's selected-row count S when the HOST knows it (moc_batch_t.n_sel_host; -1: load n_sel[b]): arguments -> sel_row
    // -> rows is one dependent round trip less than arguments -> n_sel -> sel_row -> rows.  Runs: S_r[run] (S_host >= 0 says so).
    int S_host;
    int32_t S_r[MOC_MAX_RUNS];
};

// An entry of an array inside the kernel's (single, by-value) argument struct, read straight from the kernel-argument
// segment by a uniform index: indexing the struct itself with a runtime index makes hipcc copy all of it to scratch.
template <typename T>
__device__ __forceinline__ T kernarg_at(size_t off) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(4))) const char* kp_t;
    kp_t p = (kp_t)__builtin_amdgcn_kernarg_segment_ptr();
    return *(__attribute__((address_space(4))) const T*)(p + off);
#else
    return T();
#endif
}

// what a forward workgroup works on: its slide and, with batched runs, its run's tensors (the argument block is not modified)
struct FwdRun {
    int b, S;                      // S: the host-known row count, or -1
    int64_t base, tile_slot0;
    int tile_cap;
    const unsigned char* W1img;
    const float *W2, *b1, *b2;
};
__device__ __forceinline__ FwdRun fwd_run_setup(const FwdArgs& a) {
    FwdRun r;
    r.W1img = a.W1img; r.W2 = a.W2; r.b1 = a.b1; r.b2 = a.b2; r.tile_slot0 = a.tile_slot0; r.tile_cap = a.tile_cap;
    r.S = a.S_host;
    if (a.n_runs > 0) {
        const int run = blockIdx.y;
        if (a.S_host >= 0) r.S = kernarg_at<int32_t>(offsetof(FwdArgs, S_r) + 4 * (size_t)run);
        r.b = a.slide0 + run * a.slide_stride;
        r.base = kernarg_at<int64_t>(offsetof(FwdArgs, base_r) + 8 * (size_t)run);
        r.tile_slot0 = kernarg_at<int64_t>(offsetof(FwdArgs, tile_slot0_r) + 8 * (size_t)run);
        r.tile_cap = kernarg_at<int32_t>(offsetof(FwdArgs, tile_cap_r) + 4 * (size_t)run);
        r.W1img += (int64_t)run * a.img_stride;
        r.W2 += (int64_t)run * a.w2_stride;
        r.b1 += (int64_t)run * a.par_stride;
        r.b2 += (int64_t)run * a.par_stride;
    } else {
        r.b = a.slide0 + blockIdx.y;
        r.base = a.base_host >= 0 ? a.base_host : a.row_off[r.b];
    }
    return r;
}

// the row of a selected slot o in whichever array holds its candidate scores: column k of it is ptr[k * stride]
__device__ __forceinline__ const float* cand_row(const FwdArgs& a, int64_t base, int o) {
    if (a.cand_mode == 0) return a.cand + base + o;
    return a.stats + base + a.sel_idx[base + o];
}
// the two per-row scores s_delta = |top1 - top2| and s_beta = max background logit
__device__ __forceinline__ void cand_row_scores(const FwdArgs& a, const float* cd, float& s2, float& s3) {
    const int C = a.C;
    if (a.cand_mode == 2) { s2 = cd[(int64_t)(C + 2) * a.stride]; s3 = cd[(int64_t)(C + 4) * a.stride]; }
    else if (a.cand_mode == 1) { s2 = cd[(int64_t)(2 * C) * a.stride]; s3 = cd[(int64_t)(2 * C + 2) * a.stride]; }
    else { s2 = cd[(int64_t)(2 * C) * a.stride]; s3 = cd[(int64_t)(2 * C + 1) * a.stride]; }
}
// (m1, 1/den) of the row: only the compact statistics need them (s_sigma is re-formed)
__device__ __forceinline__ void cand_row_norm(const FwdArgs& a, const float* cd, float& m1, float& rden) {
    m1 = 0.f; rden = 0.f;
    if (a.cand_mode == 2) { m1 = cd[(int64_t)a.C * a.stride]; rden = cd[(int64_t)(a.C + 1) * a.stride]; }
}
// s_p and s_sigma of class c
__device__ __forceinline__ void cand_class_scores(const FwdArgs& a, const float* cd, int c, float m1, float rden, float& s0, float& s1) {
    s0 = cd[(int64_t)c * a.stride];
    if (a.cand_mode == 2) s1 = moc_softmax_from(s0, m1, rden);
    else s1 = cd[(int64_t)(a.C + c) * a.stride];
}

// ---- W1 image ----------------------------------------------------------------------------
// The forward's B operand is W1^T: B[k][n] = W1[n][k].  Read from the [H][D] parameter tensor,
// a wave's fragment load touches 64 scattered 16-B pieces (16 rows 2 KB apart) and the address
// unit, not HBM, sets the pace.  So the kernels keep a second copy in fragment order -- one
// contiguous 1 KiB per wave-load -- rewritten by the W1 update itself, hence always in sync:
//   bf16 bags: [nt][kk][term][lane][8] bf16, element j of lane l = term t of W1[nt*16 + (l&15)][kk*32 + (l>>4)*8 + j]
//              (hi/mid/lo split, 24 mantissa bits: bf16 MFMA with fp32-exact products)
//   fp32 bags: [nt][kq][lane][4] f32,         element m of lane l = W1[nt*16 + (l&15)][kq*16 + (l>>4)*4 + m]
//   fp16 bags: as bf16, the three fp16 terms of W1 * 2^10 (moc_common.h); the forward scales back
template <bool F16>
__device__ __forceinline__ void w1_image_store_half(unsigned char* img, int D, int h, int d, float w) {
    const int KK = D / 32;
    const int nt = h >> 4, kk = d >> 5, lane = (((d & 31) >> 3) << 4) | (h & 15), j = d & 7;
    uint16_t* o = reinterpret_cast<uint16_t*>(img) + ((size_t)(nt * KK + kk) * 3 * 64 + lane) * 8 + j;
    uint16_t hi, mid, lo;
    moc_split3<F16>(w, MOC_F16_W1_SCALE, hi, mid, lo);
    o[0] = hi;
    o[64 * 8] = mid;
    o[2 * 64 * 8] = lo;
}
__device__ __forceinline__ void w1_image_store_bf16(unsigned char* img, int D, int h, int d, float w) {
    w1_image_store_half<false>(img, D, h, d, w);
}
__device__ __forceinline__ void w1_image_store_f32(unsigned char* img, int D, int h, int d, float w) {
    const int KQ = D / 16;
    const int nt = h >> 4, kq = d >> 4, lane = (((d & 15) >> 2) << 4) | (h & 15), m = d & 3;
    reinterpret_cast<float*>(img)[((size_t)(nt * KQ + kq) * 64 + lane) * 4 + m] = w;
}
// storage code (MOC_F32 / MOC_BF16 / MOC_F16) known at run time
__device__ __forceinline__ void w1_image_store(int dt, unsigned char* img, int D, int h, int d, float w) {
    if (dt == MOC_F16) w1_image_store_half<true>(img, D, h, d, w);
    else if (dt == MOC_BF16) w1_image_store_half<false>(img, D, h, d, w);
    else w1_image_store_f32(img, D, h, d, w);
}
__global__ __launch_bounds__(256) void w1_image_kernel(const float* W1, int D, unsigned char* img, int dt, int64_t par_stride = 0,
                                                       int64_t img_stride = 0) {
    const int e = blockIdx.x * 256 + threadIdx.x;       // grid = (H*D/256, runs)
    const int h = e / D, d = e - h * D;
    w1_image_store(dt, img + (int64_t)blockIdx.y * img_stride, D, h, d, W1[(int64_t)blockIdx.y * par_stride + e]);
}

// fp32 bags: one quarter's chain joins the running sum (first quarter: taken as it is), the chain starts over
__device__ __forceinline__ void fwd_fold_quarter(f32x4_t& tot, f32x4_t& acc, bool first) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        tot[i] = first ? acc[i] : moc_fadd(tot[i], acc[i]);
        acc[i] = 0.f;
    }
}

// grid (ceil(S_bound/16), n): one workgroup = 16 selected rows, wave w = hidden units 16w..16w+15.
// TILES: the tile-record variant (training step over tile records, batched runs); without it the kernel is the round-3 code --
// the extra live values cost this 256-register kernel 4.6 us at thirty classes when they were unconditional
template <bool BF16, bool F16 = false, bool TILES = false>
__global__ __launch_bounds__(256) void meta_forward_kernel(FwdArgs a) {
    __shared__ __attribute__((aligned(16))) uint4 xt[16 * 64];     // 16 rows x 1 KiB, chunk-swizzled
    __shared__ float Hs[16][H + 1];
    __shared__ float Gs[16][4];
    __shared__ float W2s[4 * H];
    FwdRun fr;
    if constexpr (TILES) fr = fwd_run_setup(a);
    else {
        fr.b = a.slide0 + blockIdx.y;
        fr.base = a.base_host >= 0 ? a.base_host : a.row_off[fr.b];
        fr.W1img = a.W1img; fr.W2 = a.W2; fr.b1 = a.b1; fr.b2 = a.b2; fr.tile_slot0 = 0; fr.tile_cap = 0;
        fr.S = a.S_host;
    }
    const int b = fr.b;
    const int64_t base = fr.base;
    const int S = fr.S >= 0 ? fr.S : a.n_sel[b];            // (host-known: one dependent load less in front of the rows)
    const int row0 = blockIdx.x * 16;
    if (row0 >= S) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int ESZ = BF16 ? 2 : 4;
    MOC_STAMP(0);
    // Epilogue operands that do not depend on the product are requested first.
    const int C = a.C;
    float pre_c[4] = {0.f, 0.f, 0.f, 0.f};
    if (threadIdx.x < 16 * C && row0 + (threadIdx.x & 15) < S) {
        const int r = threadIdx.x & 15, c = threadIdx.x >> 4;
        const float* cd = cand_row(a, base, row0 + r);
        float m1, rden;
        cand_row_norm(a, cd, m1, rden);
        cand_class_scores(a, cd, c, m1, rden, pre_c[0], pre_c[1]);
        cand_row_scores(a, cd, pre_c[2], pre_c[3]);
    }
    const float w2_pre = fr.W2[threadIdx.x & 255];
    const float bias = fr.b1[wave * 16 + (lane & 15)];
    const float b2_pre = fr.b2[threadIdx.x & 3];
    int64_t rid_e = 0;                                               // tile records: the bag row of this thread's (row, class)
    if constexpr (TILES) { if (a.tile_on) rid_e = a.sel_row[base + min(row0 + (int)(threadIdx.x & 15), S - 1)]; }
    // The 16 x D tile of x goes through LDS once per workgroup: wave w fetches rows 4w..4w+3 with
    // whole-row contiguous loads (UB bytes per row per unit) and stores 16-B chunk c of row r at
    // chunk c ^ (r & 15), so that the A-fragment reads (lane l: row l&15, chunk 4*kk + (l>>4)) hit
    // distinct banks.  W1 comes from its fragment-ordered image: one contiguous 1 KiB per load.
    const int64_t row_bytes = (int64_t)a.D * ESZ;
    const int UB = (row_bytes % 1024 == 0) ? 1024 : 512;          // bytes of a row per unit
    const int U = (int)(row_bytes / UB), cpr = UB / 16;           // units, 16-B chunks per row per unit
    const int ksteps = UB / 64;                                   // MFMA k-steps (bf16: 32 el, f32: 16 el) per unit
    const unsigned char* rp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int sr = min(row0 + wave * 4 + i, S - 1);
        rp[i] = a.X + a.sel_row[base + sr] * row_bytes;
    }
    const int KST = (int)(row_bytes / 64);                        // k-steps over all of D
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    // fp32 bags: the sum over D is formed as ((p0 + p1) + p2) + p3, p_q = the MFMA chain over the q-th quarter of the
    // columns -- the association of meta_forward_ksplit_kernel, which runs the four chains side by side (same bits)
    f32x4_t tot = {0.f, 0.f, 0.f, 0.f};
    const int QS = a.D / 64;                                      // k-steps of 16 columns per quarter
    int qc = 0, qi = 0;
    for (int u = 0; u < U; ++u) {
        if (u > 0) __syncthreads();                               // every wave is done reading the previous tile
        uint4 xv[4];
        const int xc = lane < cpr ? lane : 0;                     // lanes past the unit re-read chunk 0 (not stored)
#pragma unroll
        for (int i = 0; i < 4; ++i) xv[i] = *reinterpret_cast<const uint4*>(rp[i] + (int64_t)u * UB + xc * 16);
        if constexpr (BF16) {
            // this unit's W1 image: 16 (or 8) k-steps x 3 terms, all requested before the first MFMA
            uint4 wv[16 * 3];
            const uint4* wi = reinterpret_cast<const uint4*>(fr.W1img) + ((size_t)wave * KST + (size_t)u * ksteps) * 3 * 64 + lane;
#pragma unroll
            for (int q = 0; q < 16 * 3; ++q) if (q < ksteps * 3) wv[q] = wi[q * 64];
            if (lane < cpr) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = wave * 4 + i;
                    xt[r * 64 + (lane ^ (r & 15))] = xv[i];
                }
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                if (kk < ksteps) {
                    const int r = lane & 15;
                    const uint4 A = xt[r * 64 + ((kk * 4 + (lane >> 4)) ^ r)];
#pragma unroll
                    for (int t = 0; t < 3; ++t) acc = moc_mfma_half<F16>(A, wv[kk * 3 + t], acc);
                }
            }
        } else {
            uint4 wv[16];
            const uint4* wi = reinterpret_cast<const uint4*>(fr.W1img) + ((size_t)wave * KST + (size_t)u * ksteps) * 64 + lane;
#pragma unroll
            for (int q = 0; q < 16; ++q) if (q < ksteps) wv[q] = wi[q * 64];
            if (lane < cpr) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = wave * 4 + i;
                    xt[r * 64 + (lane ^ (r & 15))] = xv[i];
                }
            }
            __syncthreads();
#pragma unroll
            for (int kq = 0; kq < 16; ++kq) {
                if (kq < ksteps) {
                    const int r = lane & 15;
                    const uint4 xa = xt[r * 64 + ((kq * 4 + (lane >> 4)) ^ r)];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(xa.x), __uint_as_float(wv[kq].x), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(xa.y), __uint_as_float(wv[kq].y), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(xa.z), __uint_as_float(wv[kq].z), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(xa.w), __uint_as_float(wv[kq].w), acc, 0, 0, 0);
                    if (++qc == QS) {
                        qc = 0;
                        fwd_fold_quarter(tot, acc, qi++ == 0);
                    }
                }
            }
        }
    }
    if constexpr (!BF16) acc = tot;
    MOC_STAMP(1);
    {   // acc[i] = pre-activation of row (lane>>4)*4+i, hidden unit wave*16 + (lane&15)
        const int hcol = wave * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float pre = F16 ? acc[i] * (1.f / MOC_F16_W1_SCALE) : acc[i];     // exact power-of-two scaling
            Hs[(lane >> 4) * 4 + i][hcol] = fmaxf(moc_fadd(pre, bias), 0.f);
        }
        W2s[threadIdx.x] = w2_pre;
    }
    __syncthreads();
    if (a.H1) {                          // needed by the backward pass only: evaluation passes NULL
        for (int e = threadIdx.x; e < 16 * H; e += 256) {
            const int r = e >> 6, h = e & 63;
            if (row0 + r < S) a.H1[(base + row0 + r) * H + h] = Hs[r][h];
        }
    }
    if (threadIdx.x < 64) {
        const int r = threadIdx.x >> 2, i = threadIdx.x & 3;
        float z = 0.f;
        for (int h = 0; h < H; ++h) z = fmaf(Hs[r][h], W2s[i * H + h], z);
        z += b2_pre;
        const float g = 1.f / (1.f + expf(-z));
        Gs[r][i] = g;
        if (a.gates && row0 + r < S) a.gates[(base + row0 + r) * 4 + i] = g;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 16 * C; e += 256) {
        const int r = e & 15, c = e >> 4;
        if (row0 + r >= S) continue;
        float s0 = pre_c[0], s1 = pre_c[1], s2 = pre_c[2], s3 = pre_c[3];
        if (e >= 256) {   // C > 16: beyond the prefetched element
            const float* cd = cand_row(a, base, row0 + r);
            float m1, rden;
            cand_row_norm(a, cd, m1, rden);
            cand_class_scores(a, cd, c, m1, rden, s0, s1);
            cand_row_scores(a, cd, s2, s3);
        }
        float v = 0.f;   // 0 + x == x exactly, so this is the reference's running sum in both modes
        if (a.use_bits & 1u) v = moc_fadd(v, moc_fmul(Gs[r][0], s0));
        if (a.use_bits & 2u) v = moc_fadd(v, moc_fmul(Gs[r][1], s1));
        if (a.use_bits & 4u) v = moc_fadd(v, moc_fmul(Gs[r][2], s2));
        if (a.use_bits & 8u) v = moc_fadd(v, moc_fmul(Gs[r][3], s3));
        a.mixed[(int64_t)c * a.stride + base + row0 + r] = v;
    }
    if (TILES && a.tile_on && threadIdx.x < 16 * C) {                // (C <= 16 here: one element per thread)
        const int r = threadIdx.x & 15, c = threadIdx.x >> 4;
        const bool ok = row0 + r < S;
        float v = 0.f;
        if (a.use_bits & 1u) v = moc_fadd(v, moc_fmul(Gs[r][0], pre_c[0]));
        if (a.use_bits & 2u) v = moc_fadd(v, moc_fmul(Gs[r][1], pre_c[1]));
        if (a.use_bits & 4u) v = moc_fadd(v, moc_fmul(Gs[r][2], pre_c[2]));
        if (a.use_bits & 8u) v = moc_fadd(v, moc_fmul(Gs[r][3], pre_c[3]));
        const float4 lam4 = {Gs[r][0], Gs[r][1], Gs[r][2], Gs[r][3]};
        const float4 sc4 = {pre_c[0], pre_c[1], pre_c[2], pre_c[3]};
        tile_emit(a.tile, fr.tile_slot0 / TILE_R + (int64_t)c * fr.tile_cap + blockIdx.x, ok, v, row0 + r, rid_e, lam4, sc4);
    }
    MOC_STAMP(2);
}

// ---- one slide, fp32 bags (the training step of the default storage): the columns split over four wave groups ------
// meta_forward_kernel<false> is a chain of D/4 v_mfma_f32_16x16x4_f32 per wave (128 x 32 cycles = 1.7 us at D = 512, with
// three of every four SIMD cycles idle: one wave per SIMD) behind TWO dependent rounds of row loads (2-KiB rows in two
// 1-KiB units, the second requested after the first has been multiplied) -- 5.5 us from kernel start to the last MFMA
// against 3.2 us for bf16 bags (phase stamps, profiles/NOTES.md round 3).  Here a workgroup is 16 waves: wave w holds
// hidden units 16 (w & 3) .. +15 and the (w >> 2)-th QUARTER of the columns, so that the four chains of a hidden tile run
// side by side on the four waves of a SIMD (32 MFMAs each), every row is requested whole at once (wave w fetches row w:
// D / 256 sixteen-byte loads per lane, one wave-uniform row id), and the four partial tiles meet in LDS as
// ((p0 + p1) + p2) + p3 -- the association the 16- and 128-row kernels keep for fp32 bags, hence the same bits.
// grid (ceil(S_bound/16), n), 1024 threads; D <= 1024.
constexpr int FKS_PSTR = H + 4;                             // row stride of a partial tile in LDS (floats)
__host__ __device__ constexpr int fks_lds_bytes(int D) {
    return 16 * D * 4 + 4 * 16 * FKS_PSTR * 4 + 16 * (H + 1) * 4 + 16 * 4 * 4 + 4 * H * 4;
}
// DQ = D / 256: every loop over a row's pieces or a quarter's fragments has a compile-time trip count.  STATS: the candidate
// scores come from the score pass's statistics through sel_idx (cand_mode != 0: evaluation of a few slides); the training
// step reads the materialised columns -- no dependent load, no branch between the barrier and the chain.
template <int DQ, bool STATS>
__global__ __launch_bounds__(1024) void meta_forward_ksplit_kernel(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int D = DQ * 256;
    uint4* xt = reinterpret_cast<uint4*>(smem);                                     // [16][D/4] chunks, swizzled
    float* part = reinterpret_cast<float*>(smem + (size_t)16 * D * 4);              // [4][16][FKS_PSTR]
    float (*Hs)[H + 1] = reinterpret_cast<float (*)[H + 1]>(part + 4 * 16 * FKS_PSTR);
    float (*Gs)[4] = reinterpret_cast<float (*)[4]>(reinterpret_cast<float*>(Hs) + 16 * (H + 1));
    float* W2s = reinterpret_cast<float*>(Gs) + 16 * 4;
    const FwdRun fr = fwd_run_setup(a);
    const int b = fr.b;
    const int64_t base = fr.base;
    const int S = fr.S >= 0 ? fr.S : a.n_sel[b];            // (host-known: one dependent load less in front of the rows)
    const int row0 = blockIdx.x * 16;
    if (row0 >= S) return;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ht = wave & 3, kq = wave >> 2;
    MOC_STAMP(0);
    MOC_STAMP_MIN(31);                                      // (diagnostic: the first workgroup of the launch to start)
    MOC_STAMP_MAX(32);                                      // (... and the last one to start)
    // this wave's row, whole: the row id is one scalar load, the row D / 256 loads of 1 KiB per wave
    constexpr int64_t row_bytes = (int64_t)D * 4;
    constexpr int cpl = DQ;                                  // 16-byte chunks per lane
    const int sr = min(row0 + wave, S - 1);
    const int64_t rid = a.sel_row[base + sr];
    const unsigned char* rp = a.X + rid * row_bytes + lane * 16;
    MOC_STAMP_DRAIN(3);
    // W1 image of (hidden tile, quarter): D / 64 fragments of 1 KiB.  (rid >> 63 is zero: the address is made to depend on
    // the row id so that hipcc cannot hoist these loads above the row id's wait -- they must queue BEHIND the rows.)
    constexpr int QS = D / 64;
    const uint4* wi = reinterpret_cast<const uint4*>(fr.W1img) + ((size_t)ht * (D / 16) + (size_t)kq * QS) * 64 + lane + (rid >> 63);
    // (the rows first: loads return in issue order, and the tile must be in LDS before the first MFMA, while the image
    // fragments -- 128 KiB per workgroup through one CU's 64 B/clk -- may keep arriving under the chain)
    uint4 xv[cpl];
#pragma unroll
    for (int i = 0; i < cpl; ++i) xv[i] = *reinterpret_cast<const uint4*>(rp + i * 1024);
    uint4 wv[QS];
#pragma unroll
    for (int q = 0; q < QS; ++q) wv[q] = wi[q * 64];
    constexpr int cpr = D / 4;                               // chunks per row
#pragma unroll
    for (int i = 0; i < cpl; ++i) xt[wave * cpr + ((lane + i * 64) ^ wave)] = xv[i];     // chunk c of row r at c ^ r (r < 16)
    MOC_STAMP(5);
    __syncthreads();
    MOC_STAMP(6);
    // epilogue operands that do not depend on the product: requested now (straight-line code up to the barrier), consumed
    // after the chain
    const int C = a.C;
    float pre_c[4] = {0.f, 0.f, 0.f, 0.f};
    const int er = t & 15, ec = t >> 4;
    const bool e_ok = ec < C && row0 + er < S;
    if constexpr (STATS) {
        if (e_ok) {
            const float* cd = cand_row(a, base, row0 + er);
            float m1, rden;
            cand_row_norm(a, cd, m1, rden);
            cand_class_scores(a, cd, ec, m1, rden, pre_c[0], pre_c[1]);
            cand_row_scores(a, cd, pre_c[2], pre_c[3]);
        }
    } else {                                                 // unconditional (clamped): exact vmcnt counts under the chain
        const float* cd = a.cand + base + min(row0 + er, S - 1);
        const int cc = ec < C ? ec : C - 1;
        pre_c[0] = cd[(int64_t)cc * a.stride];
        pre_c[1] = cd[(int64_t)(C + cc) * a.stride];
        pre_c[2] = cd[(int64_t)(2 * C) * a.stride];
        pre_c[3] = cd[(int64_t)(2 * C + 1) * a.stride];
    }
    const float w2_pre = fr.W2[t & 255];
    const float bias = fr.b1[t & 63];
    const float b2_pre = fr.b2[t & 3];
    int64_t rid_e = 0;                                       // tile records: the bag row of this thread's (row, class)
    if (a.tile_on) rid_e = a.sel_row[base + min(row0 + er, S - 1)];
    __builtin_amdgcn_sched_barrier(0);                       // (hipcc otherwise sinks these requests below the chain)
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    {
        const int r = lane & 15;
        const uint4* xr = xt + r * cpr;
        const int c0 = kq * QS * 4 + (lane >> 4);
#pragma unroll
        for (int q = 0; q < QS; ++q) {
            const uint4 xa = xr[(c0 + q * 4) ^ r];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(xa.x), __uint_as_float(wv[q].x), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(xa.y), __uint_as_float(wv[q].y), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(xa.z), __uint_as_float(wv[q].z), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(xa.w), __uint_as_float(wv[q].w), acc, 0, 0, 0);
        }
    }
#ifdef MOC_STAMPS
    asm volatile("v_add_f32 %0, %0, 0" : "+v"(acc[0]));      // the stamp waits for the chain
#endif
    MOC_STAMP(1);
    {   // acc[i] = partial pre-activation of row (lane>>4)*4+i, hidden unit ht*16 + (lane&15), quarter kq
        float* pp = part + (size_t)(kq * 16 + (lane >> 4) * 4) * FKS_PSTR + ht * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i) pp[i * FKS_PSTR] = acc[i];
        if (t < 4 * H) W2s[t] = w2_pre;
    }
    __syncthreads();
    MOC_STAMP(7);
    {   // thread = (row t >> 6, hidden unit t & 63)
        const int r = t >> 6, h = t & 63;
        const float* pp = part + (size_t)r * FKS_PSTR + h;
        const float pre = moc_fadd(moc_fadd(moc_fadd(pp[0], pp[16 * FKS_PSTR]), pp[32 * FKS_PSTR]), pp[48 * FKS_PSTR]);
        const float hv = fmaxf(moc_fadd(pre, bias), 0.f);
        Hs[r][h] = hv;
        if (a.H1 && row0 + r < S) a.H1[(base + row0 + r) * H + h] = hv;      // needed by the backward pass only
    }
    __syncthreads();
    MOC_STAMP(8);
    if (t < 64) {
        const int r = t >> 2, i = t & 3;
        float z = 0.f;
        for (int h = 0; h < H; ++h) z = fmaf(Hs[r][h], W2s[i * H + h], z);
        z += b2_pre;
        const float g = 1.f / (1.f + expf(-z));
        Gs[r][i] = g;
        if (a.gates && row0 + r < S) a.gates[(base + row0 + r) * 4 + i] = g;
    }
    __syncthreads();
    MOC_STAMP(9);
    if (ec < C) {                                            // (uniform over a class's sixteen lanes)
        float v = 0.f;   // 0 + x == x exactly, so this is the reference's running sum in both modes
        if (a.use_bits & 1u) v = moc_fadd(v, moc_fmul(Gs[er][0], pre_c[0]));
        if (a.use_bits & 2u) v = moc_fadd(v, moc_fmul(Gs[er][1], pre_c[1]));
        if (a.use_bits & 4u) v = moc_fadd(v, moc_fmul(Gs[er][2], pre_c[2]));
        if (a.use_bits & 8u) v = moc_fadd(v, moc_fmul(Gs[er][3], pre_c[3]));
        if (e_ok) a.mixed[(int64_t)ec * a.stride + base + row0 + er] = v;
        if (a.tile_on) {
            const float4 lam4 = {Gs[er][0], Gs[er][1], Gs[er][2], Gs[er][3]};
            const float4 sc4 = {pre_c[0], pre_c[1], pre_c[2], pre_c[3]};
            tile_emit(a.tile, fr.tile_slot0 / TILE_R + (int64_t)ec * fr.tile_cap + blockIdx.x, e_ok, v, row0 + er, rid_e, lam4, sc4);
        }
    }
    MOC_STAMP(2);
    MOC_STAMP_MAX(30);                                      // (diagnostic: the last workgroup of the launch to get here)
}

// Many slides at once (evaluation): 30,000 sixteen-row workgroups each re-read the whole 192 KiB W1 image and
// the pass is bound by L2 (5.9 GB at 29 TB/s for 202 slides).  Here a workgroup owns 64 rows (four row tiles):
// every W1 fragment a wave loads feeds four MFMAs, a quarter of the L2 traffic.  Units of 512 bytes of a row
// (8 k-steps: 96 registers of fragments, three workgroups per CU).  Same products in the same order per
// (row, hidden unit) as meta_forward_kernel: bit-identical outputs.  grid (ceil(S_bound/64), n); 16-bit storage.
template <bool F16>
__global__ __launch_bounds__(256) void meta_forward64_kernel(FwdArgs a) {
    __shared__ __attribute__((aligned(16))) uint4 xt[64 * 32];     // 64 rows x 512 B of the current unit, chunk-swizzled
    __shared__ float Hs[64][H + 1];
    __shared__ float Gs[64][4];
    __shared__ float W2s[4 * H];
    const int b = a.slide0 + blockIdx.y;
    const int64_t base = a.row_off[b];
    const int S = a.n_sel[b];
    const int row0 = blockIdx.x * 64;
    if (row0 >= S) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = a.C;
    // Epilogue operands do not depend on the product: requested first.  Thread -> row tid & 63 (the same row in
    // every round), classes (tid >> 6) + 4 it: the two per-row scores once, the per-class pairs of the first
    // 32 classes here (later chunks of 32 are requested a chunk at a time, all loads before the first store).
    const int er = threadIdx.x & 63, ec0 = threadIdx.x >> 6;
    const bool erow_ok = row0 + er < S;
    const float* ecd = cand_row(a, base, erow_ok ? row0 + er : row0);
    float es2 = 0.f, es3 = 0.f, em1 = 0.f, erd = 0.f, es0[8], es1[8];
    if (erow_ok) {
        cand_row_scores(a, ecd, es2, es3);
        cand_row_norm(a, ecd, em1, erd);
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int c = ec0 + 4 * it;
        es0[it] = es1[it] = 0.f;
        if (erow_ok && c < C) cand_class_scores(a, ecd, c, em1, erd, es0[it], es1[it]);
    }
    const float w2_pre = a.W2[threadIdx.x & 255];
    const float bias = a.b1[wave * 16 + (lane & 15)];
    const float b2_pre = a.b2[threadIdx.x & 3];
    const int64_t row_bytes = (int64_t)a.D * 2;
    const int U = (int)(row_bytes / 512), KST = (int)(row_bytes / 64);
    // wave w fetches rows 16w..16w+15 of the tile: one load = two rows x 512 B (lane l: row 2j + (l >> 5), chunk l & 31)
    const unsigned char* rp[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int sr = min(row0 + wave * 16 + 2 * j + (lane >> 5), S - 1);
        rp[j] = a.X + a.sel_row[base + sr] * row_bytes + (lane & 31) * 16;
    }
    f32x4_t acc[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) acc[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int u = 0; u < U; ++u) {
        if (u > 0) __syncthreads();                               // every wave is done reading the previous unit
        uint4 xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = *reinterpret_cast<const uint4*>(rp[j] + (int64_t)u * 512);
        uint4 wv[8 * 3];
        const uint4* wi = reinterpret_cast<const uint4*>(a.W1img) + ((size_t)wave * KST + (size_t)u * 8) * 3 * 64 + lane;
#pragma unroll
        for (int q = 0; q < 8 * 3; ++q) wv[q] = wi[q * 64];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = wave * 16 + 2 * j + (lane >> 5);
            xt[r * 32 + ((lane & 31) ^ (r & 15))] = xv[j];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                const int r = rt * 16 + (lane & 15);
                const uint4 A = xt[r * 32 + ((kk * 4 + (lane >> 4)) ^ (r & 15))];
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[rt] = moc_mfma_half<F16>(A, wv[kk * 3 + t], acc[rt]);
            }
        }
    }
    {   // acc[rt][i] = pre-activation of row rt*16 + (lane>>4)*4 + i, hidden unit wave*16 + (lane&15)
        const int hcol = wave * 16 + (lane & 15);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float pre = F16 ? acc[rt][i] * (1.f / MOC_F16_W1_SCALE) : acc[rt][i];     // exact power-of-two scaling
                Hs[rt * 16 + (lane >> 4) * 4 + i][hcol] = fmaxf(moc_fadd(pre, bias), 0.f);
            }
        W2s[threadIdx.x] = w2_pre;
    }
    __syncthreads();
    if (a.H1) {                          // needed by the backward pass only: evaluation passes NULL
        for (int e = threadIdx.x; e < 64 * H; e += 256) {
            const int r = e >> 6, h = e & 63;
            if (row0 + r < S) a.H1[(base + row0 + r) * H + h] = Hs[r][h];
        }
    }
    {
        const int r = threadIdx.x >> 2, i = threadIdx.x & 3;
        float z = 0.f;
        for (int h = 0; h < H; ++h) z = fmaf(Hs[r][h], W2s[i * H + h], z);
        z += b2_pre;
        const float g = 1.f / (1.f + expf(-z));
        Gs[r][i] = g;
        if (a.gates && row0 + r < S) a.gates[(base + row0 + r) * 4 + i] = g;
    }
    __syncthreads();
    for (int cb = 0; cb < C; cb += 32) {
        if (cb > 0) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int c = cb + ec0 + 4 * it;
                if (erow_ok && c < C) cand_class_scores(a, ecd, c, em1, erd, es0[it], es1[it]);
            }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int c = cb + ec0 + 4 * it;
            if (!erow_ok || c >= C) continue;
            float v = 0.f;   // 0 + x == x exactly, so this is the reference's running sum in both modes
            if (a.use_bits & 1u) v = moc_fadd(v, moc_fmul(Gs[er][0], es0[it]));
            if (a.use_bits & 2u) v = moc_fadd(v, moc_fmul(Gs[er][1], es1[it]));
            if (a.use_bits & 4u) v = moc_fadd(v, moc_fmul(Gs[er][2], es2));
            if (a.use_bits & 8u) v = moc_fadd(v, moc_fmul(Gs[er][3], es3));
            a.mixed[(int64_t)c * a.stride + base + row0 + er] = v;
        }
    }
}

// ---- evaluation forward, 128 rows per workgroup, rows by LDS-DMA -------------------------------------------------
// meta_forward64_kernel re-reads the whole 192 KiB W1 image per 64 rows (3 KiB of image per 1 KiB of row, all of it L2 -> CU
// traffic), its loads, LDS stores, barrier and MFMAs run one after the other, and its mix waits for operands it asks for
// late: 1,000-1,070 us for the 2.0 M selected rows of a 202-slide thirty-class evaluation (1.9 TB/s of rows).  Peeling
// (scripts/bench_forward.py on a first, 256-row form of this kernel: 1,068 us whole, 724 without the mix, 336 with
// nothing but its skeleton) showed where the time is: NOT in the product (340 us of the 1,068) but in what a workgroup
// does alone on its CU before and after it -- the dependent first touches of its prologue and the mix's three round
// trips, 7 us per workgroup with nothing to hide them behind.  So:
//   * 128 rows per workgroup of four waves and 70 KiB of LDS: TWO workgroups per CU, one's prologue and epilogue beside
//     the other's product;
//   * the rows arrive by LDS-DMA (global_load_lds, 16 B per lane, per-lane source address = the gather through sel_row),
//     straight into MFMA A-fragment order -- one instruction = one 16-row x 32-column fragment, 1 KiB -- in chunks of four
//     k-steps (32 KiB), double buffered: chunk c+1 streams in while chunk c is multiplied;
//   * wave w owns hidden units 16 w .. +15 of all 128 rows: eight accumulator tiles, every W1 fragment it loads feeds
//     eight MFMAs (1.5 KiB of image per row instead of 3), the fragments of chunk c+1 requested together with its DMA,
//     so that the only vector-memory wait of the loop is the barrier's;
//   * A fragments are read from LDS by hand-issued ds_read_b128 in batches of four, one batch ahead of the MFMAs that
//     use them (an ordinary LDS read would make hipcc wait vmcnt(0) first: the DMA in flight writes LDS too);
//   * the mix's operands (the row's candidate scores, the first 16 classes) are requested at kernel start, consumed last.
// Same products in the same order per (row, hidden unit) as the 16- and 64-row kernels: bit-identical outputs.
// k-steps (64 bytes of every row) per chunk of the double-buffered row tile, and workgroups per CU.  The kernel is bound by
// dependent latency per tile (row ids -> rows -> product -> gates -> mix), not by bytes or MFMAs, so what pays is MORE
// workgroups per CU: small chunks (8 KiB per k-step) leave LDS and registers for four (16-bit bags: 128 VGPRs) or three
// (fp32 bags: 168; at one k-step per chunk they spill) instead of two with four k-steps per chunk (256 VGPRs).  Same box,
// 202 x 15,000: thirty classes bf16 868 -> 729-737 us, fp32 1,494 -> 1,375-1,396; two classes bf16 108 -> 99, fp32 290 -> 257-265.
constexpr int F128_ROWS = 128;
constexpr int f128_kc(int st) { return st == 2 ? 2 : 1; }
constexpr int f128_wgs(int st) { return st == 2 ? 3 : 4; }
constexpr int f128_buf(int kc) { return (F128_ROWS / 16) * kc * 1024; }       // one chunk of the workgroup's rows
constexpr int f128_xb(int kc) { return 2 * f128_buf(kc) > F128_ROWS * (H + 1) * 4 ? 2 * f128_buf(kc) : F128_ROWS * (H + 1) * 4; }
constexpr int f128_lds(int kc) { return f128_xb(kc) + F128_ROWS * 4 * 4 + 4 * H * 4; }     // + gates + W2: 35.5 KiB
typedef unsigned __attribute__((ext_vector_type(4))) fu32x4_t;
template <int OFF>
__device__ __forceinline__ void fwd_lds16(fu32x4_t& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void fwd_touch4(fu32x4_t (&v)[4]) {
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
}

// ST: storage of the bag -- 0 bf16, 1 fp16 (three 16-bit terms of W1 per k-step of 32 columns, v_mfma_f32_16x16x32),
// 2 fp32 (one fp32 fragment per k-step of 16 columns, four v_mfma_f32_16x16x4_f32)
template <int ST>
__global__ __launch_bounds__(256, f128_wgs(ST)) void meta_forward128_kernel(FwdArgs a) {
    constexpr int F128_KC = f128_kc(ST), F128_BUF = f128_buf(F128_KC), F128_XB = f128_xb(F128_KC);
    constexpr bool F16 = ST == 1;
    constexpr int PER = ST == 2 ? 1 : 3;                   // W1 fragments per k-step
    constexpr int ESZ = ST == 2 ? 4 : 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float (*Hs)[H + 1] = reinterpret_cast<float (*)[H + 1]>(smem);          // [128][H + 1]: aliases the chunk buffers, after the loop
    float (*Gs)[4] = reinterpret_cast<float (*)[4]>(smem + F128_XB);         // [128][4]
    float* W2s = reinterpret_cast<float*>(smem + F128_XB + F128_ROWS * 4 * 4);      // [4][H]
    const int b = a.slide0 + blockIdx.y;
    const int64_t base = a.row_off[b];
    const int S = a.n_sel[b];
    const int row0 = blockIdx.x * F128_ROWS;
    if (row0 >= S) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = a.C;
#ifdef MOC_FWD_DIAG
    const unsigned diag = a.use_bits >> 8;                 // peeling experiments (scripts/bench_forward.py): 1 no MFMA, 2 no W1 loads, 4 no row DMA, 8 no mix
#else
    constexpr unsigned diag = 0;
#endif
    // ---- the mix's operands: thread -> row tid & 127, classes (tid >> 7) + 2 it.  Requested now, consumed at the end.
    const int er = threadIdx.x & 127, ec0 = threadIdx.x >> 7;
    const bool erow_ok = row0 + er < S;
    const float* ecd = cand_row(a, base, erow_ok ? row0 + er : row0);
    // (s_p only: s_sigma is re-formed from it with the compact statistics, and loaded at the end otherwise -- the
    // registers of a second array are what the product needs)
    float es2 = 0.f, es3 = 0.f, em1 = 0.f, erd = 0.f, es0[8];
    if (erow_ok) {
        cand_row_scores(a, ecd, es2, es3);
        cand_row_norm(a, ecd, em1, erd);
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int c = ec0 + 2 * it;
        es0[it] = 0.f;
        if (erow_ok && c < C) es0[it] = ecd[(int64_t)c * a.stride];
    }
    const float w2_pre = a.W2[threadIdx.x & 255];
    const float bias = a.b1[wave * 16 + (lane & 15)];
    const float b2_pre = a.b2[threadIdx.x & 3];
    const int64_t row_bytes = (int64_t)a.D * ESZ;
    const int KK = (int)(row_bytes / 64), nchunk = KK / F128_KC;      // k-steps of 64 bytes of a row
    // this wave fetches row tiles 2 wave, 2 wave + 1 of the workgroup: lane l = row (l & 15), 16-B piece (l >> 4) of a k-step
    const unsigned char* rp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int sr = min(row0 + (wave * 2 + j) * 16 + (lane & 15), S - 1);
        rp[j] = a.X + a.sel_row[base + sr] * row_bytes + (lane >> 4) * 16;
    }
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto issue_x = [&](int c, int buf) {
        unsigned char* dst = smem + buf * F128_BUF;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kl = 0; kl < F128_KC; ++kl)
                __builtin_amdgcn_global_load_lds((gptr_t)(rp[j] + ((int64_t)c * F128_KC + kl) * 64),
                                                 (lptr_t)(dst + ((wave * 2 + j) * F128_KC + kl) * 1024), 16, 0, 0);
    };
    const fu32x4_t* wimg = reinterpret_cast<const fu32x4_t*>(a.W1img) + (size_t)wave * KK * PER * 64 + lane;
    auto load_w = [&](int c, fu32x4_t (&wv)[F128_KC * PER]) {
#pragma unroll
        for (int q = 0; q < F128_KC * PER; ++q) wv[q] = wimg[((size_t)c * F128_KC * PER + q) * 64];
    };
    f32x4_t acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    f32x4_t tot[ST == 2 ? 8 : 1] = {};                     // fp32 bags: running sum of the column quarters (meta_forward_kernel)
    const int qchunks = nchunk / 4;                        // chunks per quarter (D a multiple of 256)
    const unsigned lds0 = (unsigned)(uintptr_t)smem + lane * 16;
    auto body = [&](int c, const fu32x4_t (&cur)[F128_KC * PER], fu32x4_t (&nxt)[F128_KC * PER]) {
        if (c + 1 < nchunk) {                              // chunk c + 1: image fragments and rows, all waited for at the barrier
            if (!(diag & 2u)) load_w(c + 1, nxt);
            if (!(diag & 4u)) issue_x(c + 1, (c + 1) & 1);
        }
        const unsigned buf = lds0 + (c & 1) * F128_BUF;
        // batches of two A fragments (row tiles 2 g, 2 g + 1 at k-step kl), one batch ahead of the six MFMAs that use them
        fu32x4_t A0[2], A1[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) fwd_lds16<0>(A0[r], buf + (r * F128_KC + 0) * 1024);
#pragma unroll
        for (int st = 0; st < F128_KC * 4; ++st) {         // step = (k-step kl, batch g)
            const int kl = st >> 2, g = st & 3;
            fu32x4_t (&Ac)[2] = (st & 1) ? A1 : A0;
            fu32x4_t (&An)[2] = (st & 1) ? A0 : A1;
            if (st + 1 < F128_KC * 4) {
                const int kl_n = (st + 1) >> 2, g_n = (st + 1) & 3;
#pragma unroll
                for (int r = 0; r < 2; ++r) fwd_lds16<0>(An[r], buf + ((g_n * 2 + r) * F128_KC + kl_n) * 1024);
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            asm volatile("" : "+v"(Ac[0]), "+v"(Ac[1]));
            if (!(diag & 1u)) {
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    if constexpr (ST == 2) {
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            acc[g * 2 + r] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(Ac[r][m]), __uint_as_float(cur[kl][m]),
                                                                                  acc[g * 2 + r], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int t = 0; t < 3; ++t) acc[g * 2 + r] = moc_mfma_half<F16>(Ac[r], cur[kl * 3 + t], acc[g * 2 + r]);
                    }
                }
            }
        }
        if constexpr (ST == 2) {                           // fp32 bags: ((p0 + p1) + p2) + p3 over the quarters of the columns
            if ((c + 1) % qchunks == 0) {
                const bool first = c + 1 == qchunks;
#pragma unroll
                for (int r = 0; r < 8; ++r) fwd_fold_quarter(tot[r], acc[r], first);
            }
        }
        __syncthreads();                                   // chunk c + 1 has landed for everybody; this buffer is free
    };
    fu32x4_t wA[F128_KC * PER], wB[F128_KC * PER];
    load_w(0, wA);
    issue_x(0, 0);
    __syncthreads();
    for (int c = 0; c < nchunk; c += 2) {                  // (two chunks per trip: the fragment sets swap roles without copies)
        body(c, wA, wB);
        if (c + 1 < nchunk) body(c + 1, wB, wA);
    }
    {   // acc[r][i] = pre-activation of row r*16 + (lane>>4)*4 + i, hidden unit wave*16 + (lane&15)
        const int hcol = wave * 16 + (lane & 15);
        if constexpr (ST == 2) {
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] = tot[r];
        }
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float pre = F16 ? acc[r][i] * (1.f / MOC_F16_W1_SCALE) : acc[r][i];     // exact power-of-two scaling
                Hs[r * 16 + (lane >> 4) * 4 + i][hcol] = fmaxf(moc_fadd(pre, bias), 0.f);
            }
        W2s[threadIdx.x] = w2_pre;
    }
    __syncthreads();
    if (a.H1) {                          // needed by the backward pass only: evaluation passes NULL
        for (int e = threadIdx.x; e < F128_ROWS * H; e += 256) {
            const int r = e >> 6, h = e & 63;
            if (row0 + r < S) a.H1[(base + row0 + r) * H + h] = Hs[r][h];
        }
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = (threadIdx.x >> 2) + rr * 64, i = threadIdx.x & 3;
        float z = 0.f;
        for (int h = 0; h < H; ++h) z = fmaf(Hs[r][h], W2s[i * H + h], z);
        z += b2_pre;
        const float g = 1.f / (1.f + expf(-z));
        Gs[r][i] = g;
        if (a.gates && row0 + r < S) a.gates[(base + row0 + r) * 4 + i] = g;
    }
    __syncthreads();
    if (erow_ok && !(diag & 8u)) {
        const float g0 = Gs[er][0], g1 = Gs[er][1], g2 = Gs[er][2], g3 = Gs[er][3];
        for (int cb = 0; cb < C; cb += 16) {
            if (cb > 0) {                                  // beyond the 16 classes requested at the start
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int c = cb + ec0 + 2 * it;
                    if (c < C) es0[it] = ecd[(int64_t)c * a.stride];
                }
            }
            float es1[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int c = cb + ec0 + 2 * it;
                es1[it] = 0.f;
                if (c < C) es1[it] = a.cand_mode == 2 ? moc_softmax_from(es0[it], em1, erd) : ecd[(int64_t)(C + c) * a.stride];
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int c = cb + ec0 + 2 * it;
                if (c >= C) continue;
                const float s1 = es1[it];
                float v = 0.f;   // 0 + x == x exactly, so this is the reference's running sum in both modes
                if (a.use_bits & 1u) v = moc_fadd(v, moc_fmul(g0, es0[it]));
                if (a.use_bits & 2u) v = moc_fadd(v, moc_fmul(g1, s1));
                if (a.use_bits & 4u) v = moc_fadd(v, moc_fmul(g2, es2));
                if (a.use_bits & 8u) v = moc_fadd(v, moc_fmul(g3, es3));
                a.mixed[(int64_t)c * a.stride + base + row0 + er] = v;
            }
        }
    }
}

// ablation mixes (main_moc.py:538-553): grid (ceil(S_bound/256), n), thread -> selected row
__global__ __launch_bounds__(256) void fixed_mix_kernel(FwdArgs a, int mode) {
    const int b = a.slide0 + blockIdx.y;
    const int64_t base = a.row_off[b];
    const int S = a.n_sel[b], C = a.C;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    const float* cd = a.cand + base + s;
    const float s2 = cd[(int64_t)(2 * C) * a.stride], s3 = cd[(int64_t)(2 * C + 1) * a.stride];
    for (int c = 0; c < C; ++c) {
        const float s0 = cd[(int64_t)c * a.stride], s1 = cd[(int64_t)(C + c) * a.stride];
        float v;
        if (mode == 0) v = moc_fadd(moc_fadd(moc_fadd(moc_fmul(0.25f, s0), moc_fmul(0.25f, s1)), moc_fmul(0.25f, s2)), moc_fmul(0.25f, s3));
        else if (mode == 1) v = moc_fadd(moc_fadd(moc_fadd(s0, s1), s2), s3);
        else v = fmaxf(fmaxf(s0, s1), fmaxf(s2, s3));
        a.mixed[(int64_t)c * a.stride + base + s] = v;
    }
}

// ------------------------------------------------------------------ loss (+ pair gradients)
struct FinishArgs {
    const int64_t* row_off;
    const int64_t* sel_row;
    const int32_t* n_sel;
    const float* cand;
    const float *H1, *gates, *pooled;
    const float* mixed_in;          // fused kernel: [C, stride] mixed scores
    const int32_t *topk_idx, *topk_cnt;
    const int64_t* labels;
    float* loss;
    int32_t* pred;
    // train only
    float *W2, *b2, *b1;
    float *m_W2, *m_b2, *m_b1, *v_W2, *v_b2, *v_b1;
    float *g_W2, *g_b2, *g_b1;
    float* pair_dh;
    const unsigned char* X;
    int64_t* pair_row;
    int32_t* n_pair;
    int64_t stride;
    int64_t base_host;              // >= 0: first slot of the (single) slide; seg_host its slot count
    int seg_host, D, xdt;     // xdt: storage code of X (MOC_F32 / MOC_BF16 / MOC_F16)
    int C, K, slide0, train, apply_adam;
    uint32_t use_bits;
    AdamCoef adam;
    // graph replay (moc_train_steps_graph): the step's coefficients are adam_tab[adam_ctr[0] + adam_pos] instead of
    // `adam` -- kernel arguments are frozen at capture, the Adam step count is not
    const AdamCoef* adam_tab;
    const int32_t* adam_ctr;
    int adam_pos;
};

// the step's Adam coefficients: the kernel argument, or (graph replay) the table entry of this position in the pass.
// Uniform addresses: two scalar loads, requested where this is called (kernel start), consumed at the very end.
__device__ __forceinline__ AdamCoef step_coef(const FinishArgs& a) {
    AdamCoef k = a.adam;
    if (a.adam_tab) k = a.adam_tab[a.adam_ctr[0] + a.adam_pos];
    return k;
}

constexpr int FIN_CHUNK = 256;   // pairs whose H1 rows are staged in LDS at a time (64 KiB)

// grid (n): one workgroup per slide.  Eval: CE + argmax only.  Train (n == 1): also the
// gradient "pairs" and the gradient/Adam of b1, W2, b2.
__global__ __launch_bounds__(256) void finish_kernel(FinishArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float dpool[256];   // d loss / d pooled[c]
    const int b = a.slide0 + blockIdx.x, C = a.C;
    const float* x = a.pooled + (int64_t)b * C;
    const int y = (int)a.labels[b];
    if (threadIdx.x == 0) {
        float mx = -INFINITY;
        int arg = 0;
        for (int c = 0; c < C; ++c) if (x[c] > mx) { mx = x[c]; arg = c; }
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(x[c] - mx);
        const float lse = mx + logf(se);
        a.loss[b] = lse - x[y];
        a.pred[b] = arg;
        if (a.train) for (int c = 0; c < C; ++c) dpool[c] = expf(x[c] - lse) - (c == y ? 1.f : 0.f);
    }
    if (!a.train) return;
    // Adam operands: requested now, consumed at the very end
    float oW = 0.f, oM = 0.f, oV = 0.f, oW2 = 0.f, oM2 = 0.f, oV2 = 0.f;
    if (a.apply_adam) {
        const int tt = threadIdx.x;
        oW = a.W2[tt]; oM = a.m_W2[tt]; oV = a.v_W2[tt];
        if (tt < 4) { oW2 = a.b2[tt]; oM2 = a.m_b2[tt]; oV2 = a.v_b2[tt]; }
        else if (tt >= 64 && tt < 64 + H) { oW2 = a.b1[tt - 64]; oM2 = a.m_b1[tt - 64]; oV2 = a.v_b1[tt - 64]; }
    }
    const int64_t base = a.row_off[b];
    const int cnt = a.topk_cnt[(int64_t)b * C];
    __syncthreads();

    // pairs p = (c, r): r-th pooled row of class c.  n_pair = sum_c cnt[c] <= C*K
    float* dz = reinterpret_cast<float*>(smem);                 // [P][4]
    int* prow = reinterpret_cast<int*>(dz + (size_t)C * a.K * 4);   // [P] position s of the pair's row
    // every class pools the same number of rows: cnt = min(K, S)
    const int P = C * cnt;
    for (int p = threadIdx.x; p < P; p += 256) {
        const int c = p / cnt, r = p - c * cnt;
        const int s = a.topk_idx[((int64_t)b * C + c) * a.K + r];
        prow[p] = s;
        const float g = dpool[c] / (float)cnt;
        const float* cd = a.cand + base + s;
        const float sc[4] = {cd[(int64_t)c * a.stride], cd[(int64_t)(C + c) * a.stride],
                             cd[(int64_t)(2 * C) * a.stride], cd[(int64_t)(2 * C + 1) * a.stride]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float lam = a.gates[(base + s) * 4 + i];
            const float dlam = (a.use_bits >> i & 1u) ? g * sc[i] : 0.f;
            dz[p * 4 + i] = dlam * lam * (1.f - lam);
        }
        a.pair_row[p] = a.sel_row[base + s];
    }
    if (threadIdx.x == 0) *a.n_pair = P;
    // H1 rows of the pairs go through LDS in chunks of FIN_CHUNK pairs (one coalesced round of gathers
    // per chunk): read per element from global memory, the W2 / b1 reductions below were P dependent
    // gathers long (68 us at C*K = 300).
    float* H1s = reinterpret_cast<float*>(prow + (size_t)C * a.K);   // [FIN_CHUNK][H]
    float* W2s = H1s + FIN_CHUNK * H;                                // [4][H]
    const int t = threadIdx.x, h = t & 63, g4 = t >> 6;
    W2s[t] = a.W2[t];
    float gW2 = 0.f, gb1 = 0.f;                 // W2[g4][h]; partial of b1[h] over the pairs p = g4 (mod 4)
    for (int c0 = 0; c0 < P; c0 += FIN_CHUNK) {
        const int n = P - c0 < FIN_CHUNK ? P - c0 : FIN_CHUNK;
        __syncthreads();                        // previous chunk consumed (first pass: dz / prow / W2s written)
        for (int e0 = 0; e0 < n * H; e0 += 16 * 256) {          // 16 independent gathers in flight per thread
            float r[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int e = e0 + q * 256 + t;
                r[q] = e < n * H ? a.H1[(base + prow[c0 + (e >> 6)]) * H + h] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int e = e0 + q * 256 + t;
                if (e < n * H) H1s[e] = r[q];
            }
        }
        __syncthreads();
        for (int pp = g4; pp < n; pp += 4) {    // dh[p][h] = (sum_i dz[p][i] * W2[i][h]) * [H1 > 0]
            const float* dzp = dz + (size_t)(c0 + pp) * 4;
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) v = fmaf(dzp[i], W2s[i * H + h], v);
            v = H1s[pp * H + h] > 0.f ? v : 0.f;
            a.pair_dh[(size_t)(c0 + pp) * H + h] = v;
            gb1 += v;
        }
        for (int pp = 0; pp < n; ++pp) gW2 = fmaf(dz[(size_t)(c0 + pp) * 4 + g4], H1s[pp * H + h], gW2);
    }
    __syncthreads();
    H1s[t] = gb1;                               // b1[h] = the four partial sums, in a fixed order
    __syncthreads();
    const float gs = a.adam.grad_scale;
    if (a.apply_adam) {
        adam_update(oW, oM, oV, gW2 * gs, a.adam);
        a.W2[t] = oW; a.m_W2[t] = oM; a.v_W2[t] = oV;
    } else a.g_W2[t] = gW2;
    if (t < 4) {
        float g = 0.f;
        for (int p = 0; p < P; ++p) g += dz[p * 4 + t];
        if (a.apply_adam) {
            adam_update(oW2, oM2, oV2, g * gs, a.adam);
            a.b2[t] = oW2; a.m_b2[t] = oM2; a.v_b2[t] = oV2;
        } else a.g_b2[t] = g;
    }
    if (t >= 64 && t < 64 + H) {
        const int hh = t - 64;
        const float g = ((H1s[hh] + H1s[64 + hh]) + H1s[128 + hh]) + H1s[192 + hh];
        if (a.apply_adam) {
            adam_update(oW2, oM2, oV2, g * gs, a.adam);
            a.b1[hh] = oW2; a.m_b1[hh] = oM2; a.v_b1[hh] = oV2;
        } else a.g_b1[hh] = g;
    }
}


// ------------------------------------------------------------------ fused pooling + loss (+ step)
// top-K mean for every class, cross entropy, argmax and (train) the pair gradients with the
// Adam step of b1/W2/b2 in ONE workgroup of 16 waves -- for K <= 16, C <= 16, S <= 4096.
//
// top-K without K block-wide reductions: element i lives on wave i%16.  The K-th largest of
// the 16 per-wave maxima is a lower bound T0 of the K-th largest element (K waves hold an
// element >= T0), so only elements >= T0 can be in the top-K: a few dozen out of thousands.
// They go to a short LDS list per class, and one wave per class extracts them in value order
// (K wave-wide max reductions over the short list, DPP row operations, no LDS round trips).
constexpr int PS_VPT = 4;       // values per thread per class  (S <= 4096)
constexpr int FS_MAX_DYN_LDS = 160 * 1024 - 64;   // the step kernel also holds a few static LDS words
constexpr int FS_MAX_XS = 20480; // one-launch step: pair rows staged in LDS, C*K*D <= this many floats (80 KiB)
constexpr int PS_CAP_MAX = 1024; // candidate list entries per class (the launch picks cap <= this)

__device__ __forceinline__ float key_to_float(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

__device__ __forceinline__ unsigned long long ps_key(const float* col, int i, int S) {
    return i < S ? ((unsigned long long)moc_key_desc(col[i]) << 32) | (unsigned)(~(unsigned)i) : 0ull;
}

struct PoolLds {
    unsigned long long *list, *wmax;     // [C][cap] candidate keys, [C][16] per-wave maxima
    float *pooled_s, *dpool;             // [C] pooled logits, d loss / d pooled
    int *ncand, *topk_s;                 // [C] candidate counts, [C][K] pooled rows in value order
};

// cross entropy of the pooled logits, argmax and (train) d loss / d pooled: wave 0, one class per lane;
// CE_OFF = widest xor offset (8: C <= 16, 32: C <= 64)
template <int CE_OFF>
__device__ __forceinline__ void ce_wave0(const FinishArgs& a, int b, int C, int y, const float* pooled_s, float* dpool,
                                         bool write_out, int which_wave = 0) {
    const int lane = threadIdx.x & 63;
    if ((int)(threadIdx.x >> 6) != which_wave) return;
    const float xv = lane < C ? pooled_s[lane] : -INFINITY;
    float mx = xv;
    int arg = lane < C ? lane : 0x7fffffff;
    for (int off = CE_OFF; off > 0; off >>= 1) {
        const float om = __shfl_xor(mx, off, 64);
        const int oa = __shfl_xor(arg, off, 64);
        if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
    }
    const float ex = lane < C ? expf(xv - mx) : 0.f;
    float se = ex;
    for (int off = CE_OFF; off > 0; off >>= 1) se += __shfl_xor(se, off, 64);
    const float lse = mx + logf(se);
    if (lane < C && a.train) dpool[lane] = expf(xv - lse) - (lane == y ? 1.f : 0.f);
    if (lane == 0 && write_out) {
        a.loss[b] = lse - pooled_s[y];
        a.pred[b] = arg;
    }
}

// Pooling + loss of slide b by ONE workgroup of 16 waves (all threads must call): leaves the pooled
// rows in L.topk_s, the pooled logits in L.pooled_s and (train) d loss/d pooled in L.dpool; returns
// k = rows pooled per class.  `write_out`: this workgroup also publishes pooled/topk/loss/pred.
// VPT = scores per thread per class (S <= 1024 * VPT); CE_OFF = widest xor offset of the cross-entropy
// reduction (8: C <= 16 on lanes 0..15; 32: C <= 64 on the whole wave).
// CG = classes handled per round (CG * VPT 64-bit keys live in registers).
// `after_first_loads()` is called once, right after the first group's score loads have been ISSUED: the place for the
// caller's own prefetches (parameters, moments) -- loads return in issue order, so whatever is requested before the
// scores delays the first thing this kernel waits for.
struct PoolNoHook { __device__ __forceinline__ void operator()() const {} };
template <int VPT = PS_VPT, int CE_OFF = 8, int CG = 4, typename Hook = PoolNoHook>
__device__ __forceinline__ int pool_phase(const FinishArgs& a, int b, const PoolLds& L, int PS_CAP, bool write_out,
                                          float* pooled_out, int32_t* topk_idx_out, int32_t* topk_cnt_out,
                                          int64_t* base_out, Hook after_first_loads = Hook()) {
    const int C = a.C, K = a.K;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long* list = L.list;
    unsigned long long* wmax = L.wmax;
    float* pooled_s = L.pooled_s;
    float* dpool = L.dpool;
    int* ncand = L.ncand;
    int* topk_s = L.topk_s;
    const int64_t base = a.base_host >= 0 ? a.base_host : a.row_off[b];
    const int seg = a.base_host >= 0 ? a.seg_host : (int)(a.row_off[b + 1] - base);   // reads below stay inside the slide's slots
    const int y = (int)a.labels[b];
    if (threadIdx.x < C) ncand[threadIdx.x] = 0;
    // ---- the mixed scores, once: element q*1024 + tid (contiguous per wave) of up to 4 classes at a
    // time.  The loads are issued before n_sel is known (clamped to the slide's slots), masked after.
    // With fewer than k non-empty waves (S < 64k) the threshold below is 0 and every element is a
    // candidate: S <= 64*15 then, which the list holds.
    const int S = a.n_sel[b];
    const int k = K < S ? K : S;
    for (int c0 = 0; c0 < C; c0 += CG) {
        unsigned long long key[CG][VPT];
        {   // the group's loads, ALL of them before the first wait: unconditional (an absent class re-reads the last one,
            // a slot past the slide its last slot), masked afterwards -- a branch per class made every class its own
            // memory round trip
            float v[CG][VPT];
#pragma unroll
            for (int cc = 0; cc < CG; ++cc) {
                const int cl = c0 + cc < C ? c0 + cc : C - 1;
                const float* col = a.mixed_in + (int64_t)cl * a.stride + base;
#pragma unroll
                for (int q = 0; q < VPT; ++q) {
                    const int i = q * 1024 + (int)threadIdx.x;
                    v[cc][q] = col[i < seg ? i : seg - 1];
                }
            }
            if (c0 == 0) {
                after_first_loads();
                __builtin_amdgcn_sched_barrier(0);                // (hipcc otherwise sinks the hook's requests past the barriers below)
            }
#pragma unroll
            for (int cc = 0; cc < CG; ++cc)
#pragma unroll
                for (int q = 0; q < VPT; ++q) {
                    const int i = q * 1024 + (int)threadIdx.x;
                    key[cc][q] = (c0 + cc < C && i < S) ? ((unsigned long long)moc_key_desc(v[cc][q]) << 32) | (unsigned)(~(unsigned)i) : 0ull;
                }
        }
#pragma unroll
        for (int cc = 0; cc < CG; ++cc) {
            if (c0 + cc < C) {
                unsigned long long m = 0;
#pragma unroll
                for (int q = 0; q < VPT; ++q) m = key[cc][q] > m ? key[cc][q] : m;
                m = wave_max_u64(m);
                if (lane == 0) wmax[(c0 + cc) * 16 + wave] = m;
            }
        }
        __syncthreads();
        MOC_STAMP(11);
        // threshold = K-th largest wave maximum (keys are unique; 0 = empty wave): wave cc finds class c0 + cc's and leaves
        // it in the class's first wmax slot -- ONE wave per class, side by side on different SIMDs, not all sixteen each
        // for every class (an instruction all sixteen waves execute costs the CU four times one wave's)
        if (wave < CG && c0 + wave < C) {
            const int c = c0 + wave;
            const unsigned long long mine = wmax[c * 16 + (lane & 15)];
            int rank = 0;
#pragma unroll
            for (int l = 0; l < 16; ++l) rank += wmax[c * 16 + l] > mine ? 1 : 0;
            const unsigned long long hit = __ballot(lane < 16 && rank == k - 1 && mine != 0ull);
            unsigned long long T0 = 0ull;
            if (hit != 0ull && k <= 16) {
                const int src = __ffsll((long long)hit) - 1;
                const unsigned lo = (unsigned)__shfl((int)(unsigned)mine, src, 64);
                const unsigned hi = (unsigned)__shfl((int)(unsigned)(mine >> 32), src, 64);
                T0 = ((unsigned long long)hi << 32) | lo;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // every lane's reads of the slots are done
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) wmax[c * 16] = T0;
        }
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < CG; ++cc) {
            const int c = c0 + cc;
            if (c >= C) break;
            const unsigned long long T0 = wmax[c * 16];
#pragma unroll
            for (int q = 0; q < VPT; ++q) {
                const unsigned long long v = key[cc][q];
                if (v != 0ull && v >= T0) {
                    const int pos = atomicAdd(&ncand[c], 1);
                    if (pos < PS_CAP) list[(size_t)c * PS_CAP + pos] = v;
                }
            }
        }
    }
    __syncthreads();
    MOC_STAMP(12);
    // ---- extraction: wave w takes classes w, w+16, ...  Short lists (<= 64 candidates, the normal
    // case) are ranked in one sweep: a candidate's rank is the number of larger ones, and rank r < k
    // IS its position in value order.  Longer lists fall back to k wave-wide maximum rounds.
    for (int c = wave; c < C; c += 16) {
        const int n = ncand[c];
        if (k > 0 && n <= 64) {
            const unsigned long long mine = lane < n ? list[(size_t)c * PS_CAP + lane] : 0ull;
            int rank = 0;
            for (int l = 0; l < n; ++l) rank += list[(size_t)c * PS_CAP + l] > mine ? 1 : 0;
            float* topv = reinterpret_cast<float*>(wmax + c * 16);       // wave maxima are spent: 16 x 8 B of scratch
            if (lane < n && rank < k) {
                topk_s[c * K + rank] = (int)(~(unsigned)mine);
                topv[rank] = key_to_float((unsigned)(mine >> 32));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {                                             // summed largest first
                float sum = 0.f;
                for (int r = 0; r < k; ++r) sum += topv[r];
                pooled_s[c] = sum / (float)k;
            }
        } else if (k > 0) {
            const bool overflow = n > PS_CAP;        // pathological ties: rank straight from global
            const float* col = a.mixed_in + (int64_t)c * a.stride + base;
            unsigned long long prev = ~0ull;
            float sum = 0.f;
            for (int r = 0; r < k; ++r) {
                unsigned long long best = 0ull;
                if (!overflow) {
                    for (int i = lane; i < n; i += 64) {
                        const unsigned long long v = list[(size_t)c * PS_CAP + i];
                        if (v < prev && v > best) best = v;
                    }
                } else {
                    for (int i = lane; i < S; i += 64) {
                        const unsigned long long v = ps_key(col, i, S);
                        if (v < prev && v > best) best = v;
                    }
                }
                best = wave_max_u64(best);
                prev = best;
                sum += key_to_float((unsigned)(best >> 32));
                if (lane == 0) topk_s[c * K + r] = (int)(~(unsigned)best);
            }
            if (lane == 0) pooled_s[c] = sum / (float)k;
        } else if (lane == 0) {
            pooled_s[c] = __uint_as_float(0x7FC00000u);     // mean over no rows = NaN, like torch
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0 && write_out) {
            pooled_out[(int64_t)b * C + c] = pooled_s[c];
            if (topk_cnt_out) topk_cnt_out[(int64_t)b * C + c] = k;
        }
        if (topk_idx_out && write_out)
            for (int r = lane; r < K; r += 64)
                topk_idx_out[((int64_t)b * C + c) * K + r] = r < k ? topk_s[c * K + r] : -1;
    }
    __syncthreads();
    MOC_STAMP(13);
    // ---- cross entropy, argmax: wave 0, one class per lane
    ce_wave0<CE_OFF>(a, b, C, y, pooled_s, dpool, write_out);
    *base_out = base;
    return k;
}

// grid (n): one workgroup (1024 threads) per slide
__global__ __launch_bounds__(1024) void pool_step_kernel(FinishArgs a, float* pooled_out, int32_t* topk_idx_out,
                                                         int32_t* topk_cnt_out, int PS_CAP) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = a.slide0 + blockIdx.x, C = a.C, K = a.K;
    // LDS carve (all dynamic)
    unsigned long long* list = reinterpret_cast<unsigned long long*>(smem);          // [C][PS_CAP]
    unsigned long long* wmax = list + (size_t)C * PS_CAP;                           // [C][16]
    float* pooled_s = reinterpret_cast<float*>(wmax + (size_t)C * 16);               // [C]
    float* dpool = pooled_s + C;                                                     // [C]
    int* ncand = reinterpret_cast<int*>(dpool + C);                                  // [C]
    int* topk_s = ncand + C;                                                         // [C][K]
    float* dz = reinterpret_cast<float*>(topk_s + C * K);                            // [P][4]   (train)
    float* H1s = dz + (size_t)C * K * 4;                                             // [P][H]
    float* dhs = H1s + (size_t)C * K * H;                                            // [P][H]
    float* W2s = dhs + (size_t)C * K * H;                                            // [4][H]
    MOC_STAMP(10);
    // operands that do not depend on this slide's scores: requested first, consumed last
    float pW = 0.f, pM = 0.f, pV = 0.f;       // parameter / exp_avg / exp_avg_sq this thread will step
    if (a.train) {
        const int t = threadIdx.x;
        if (t < 4 * H) { W2s[t] = pW = a.W2[t]; if (a.apply_adam) { pM = a.m_W2[t]; pV = a.v_W2[t]; } }
        else if (t < 4 * H + 4) { pW = a.b2[t - 4 * H]; if (a.apply_adam) { pM = a.m_b2[t - 4 * H]; pV = a.v_b2[t - 4 * H]; } }
        else if (t >= 320 && t < 320 + H) { pW = a.b1[t - 320]; if (a.apply_adam) { pM = a.m_b1[t - 320]; pV = a.v_b1[t - 320]; } }
    }
    int64_t base;
    const PoolLds L = {list, wmax, pooled_s, dpool, ncand, topk_s};
    const int k = pool_phase(a, b, L, PS_CAP, true, pooled_out, topk_idx_out, topk_cnt_out, &base);
    MOC_STAMP(14);
    if (!a.train) return;
    __syncthreads();
    MOC_STAMP(15);
    // ---- pairs p = (class c, r-th pooled row): dz and the gathered H1 rows, one round of loads.
    // Entries p >= P (a slide with fewer than K selected rows) are zero / repeat pair 0's row so the
    // W1 kernel can run over the static bound C*K without reading n_pair.
    const int P = C * k, Pmax = C * K;
    // two (pair, hidden) elements per thread per sweep, both gathers in flight together
    for (int e0 = threadIdx.x; e0 < Pmax * H; e0 += 2048) {
        float hv[2];
        int sx[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int e = e0 + t * 1024;
            const int p = e >> 6, h = e & 63;
            const int pp = p < P ? p : 0;
            sx[t] = (P > 0 && e < Pmax * H) ? topk_s[(pp / k) * K + (pp - (pp / k) * k)] : 0;
            hv[t] = (p < P && e < Pmax * H) ? a.H1[(base + sx[t]) * H + h] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int e = e0 + t * 1024;
            if (e >= Pmax * H) continue;
            const int p = e >> 6, h = e & 63, sidx = sx[t];
            H1s[e] = hv[t];
            if (h < 4) {
                const int i = h;
                float dzv = 0.f;
                if (p < P) {
                    const int c = p / k;
                    const float g = dpool[c] / (float)k;
                    const float* cd = a.cand + base + sidx;
                    const float sc = i == 0 ? cd[(int64_t)c * a.stride] : i == 1 ? cd[(int64_t)(C + c) * a.stride]
                                   : i == 2 ? cd[(int64_t)(2 * C) * a.stride] : cd[(int64_t)(2 * C + 1) * a.stride];
                    const float lam = a.gates[(base + sidx) * 4 + i];
                    const float dlam = (a.use_bits >> i & 1u) ? g * sc : 0.f;
                    dzv = dlam * lam * (1.f - lam);
                }
                dz[p * 4 + i] = dzv;
            }
            if (h == 4) a.pair_row[p] = P > 0 ? a.sel_row[base + sidx] : 0;
        }
    }
    if (threadIdx.x == 0) *a.n_pair = P;
    __syncthreads();
    MOC_STAMP(16);
    // dh[p][h] = (sum_i dz[p][i] * W2[i][h]) * [H1 > 0]
    for (int e = threadIdx.x; e < Pmax * H; e += 1024) {
        const int p = e >> 6, h = e & 63;
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) v = fmaf(dz[p * 4 + i], W2s[i * H + h], v);
        v = H1s[e] > 0.f ? v : 0.f;
        dhs[e] = v;
        a.pair_dh[e] = v;
    }
    __syncthreads();
    MOC_STAMP(17);
    const float gs = a.adam.grad_scale;
    if (threadIdx.x < 4 * H) {          // W2 [4][H]
        const int i = threadIdx.x >> 6, h = threadIdx.x & 63;
        float g = 0.f;
        for (int p = 0; p < P; ++p) g = fmaf(dz[p * 4 + i], H1s[p * H + h], g);
        if (a.apply_adam) {
            adam_update(pW, pM, pV, g * gs, a.adam);
            a.W2[threadIdx.x] = pW; a.m_W2[threadIdx.x] = pM; a.v_W2[threadIdx.x] = pV;
        } else a.g_W2[threadIdx.x] = g;
    } else if (threadIdx.x < 4 * H + 4) {
        const int i = threadIdx.x - 4 * H;
        float g = 0.f;
        for (int p = 0; p < P; ++p) g += dz[p * 4 + i];
        if (a.apply_adam) {
            adam_update(pW, pM, pV, g * gs, a.adam);
            a.b2[i] = pW; a.m_b2[i] = pM; a.v_b2[i] = pV;
        } else a.g_b2[i] = g;
    } else if (threadIdx.x >= 320 && threadIdx.x < 320 + H) {
        const int h = threadIdx.x - 320;
        float g = 0.f;
        for (int p = 0; p < P; ++p) g += dhs[p * H + h];
        if (a.apply_adam) {
            adam_update(pW, pM, pV, g * gs, a.adam);
            a.b1[h] = pW; a.m_b1[h] = pM; a.v_b1[h] = pV;
        } else a.g_b1[h] = g;
    }
    MOC_STAMP(18);
}

// ---- pooling + loss + backward + the WHOLE Adam step in one launch --------------------------------
// grid (H/4): every workgroup repeats the (cheap, latency-bound) pooling of the slide for itself --
// 16 workgroups doing it side by side cost no more time than one -- and then owns 4 hidden units of
// W1 (4 x D parameters): it forms their gradient from the <= C*K pairs it has just found and steps
// them.  No pair list goes through memory, no second launch, no dependent n_pair -> row -> load chain.
// Workgroup 0 also publishes the slide's outputs and steps b1, W2, b2; since every workgroup READS
// W2 (for dh) while workgroup 0 WRITES it, W2 is double buffered by the caller (W2 in, W2out out).
struct FusedArgs {
    FinishArgs f;
    float *W1, *m_W1, *v_W1, *g_W1;
    unsigned char* W1img;
    float* W2out;
    int img_dt;
    P2pArgs x;              // world > 1: sum the gradient over the node's ranks before the update
};


__global__ __launch_bounds__(1024) void pool_w1_step_kernel(FusedArgs g, float* pooled_out, int32_t* topk_idx_out,
                                                            int32_t* topk_cnt_out, int PS_CAP) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const FinishArgs& a = g.f;
    const int b = a.slide0, C = a.C, K = a.K, D = a.D;
    // HW hidden units of W1 per workgroup: 4 (16 workgroups: the in-kernel gradient exchange has 16 channels) or 2 (32
    // workgroups: half the Adam arithmetic per thread -- sixteen waves on one CU make that the kernel's last 2 us)
    const int HW = H / (int)gridDim.x, JN = HW >> 1;
    const int wg = blockIdx.x, h_lo = wg * HW;
    unsigned long long* list = reinterpret_cast<unsigned long long*>(smem);          // [C][PS_CAP]
    unsigned long long* wmax = list + (size_t)C * PS_CAP;                           // [C][16]
    float* pooled_s = reinterpret_cast<float*>(wmax + (size_t)C * 16);               // [C]
    float* dpool = pooled_s + C;                                                     // [C]
    int* ncand = reinterpret_cast<int*>(dpool + C);                                  // [C]
    int* topk_s = ncand + C;                                                         // [C][K]
    float* dz = reinterpret_cast<float*>(topk_s + C * K);                            // [P][4]
    float* H1s = dz + (size_t)C * K * 4;                                             // [P][H]
    float* dhs = H1s + (size_t)C * K * H;                                            // [P][H]
    float* W2s = dhs + (size_t)C * K * H;                                            // [4][H]
    // (offsets from smem, which is 16-byte aligned, not address arithmetic through integers: the latter loses the LDS
    // address space and every access through the pointer becomes a flat_ operation)
    const size_t row_off = ((size_t)(reinterpret_cast<unsigned char*>(W2s + 4 * H) - smem) + 7) & ~(size_t)7;
    int64_t* row_s = reinterpret_cast<int64_t*>(smem + row_off);                     // [P] bag rows of the pairs
    float* xs = reinterpret_cast<float*>(smem + ((row_off + (size_t)C * K * 8 + 15) & ~(size_t)15));             // [P][D], 16-B aligned
    MOC_STAMP(10);
    // thread t: column d = t % D' of hidden units h_lo + JN*(t / D') + {0 .. JN-1}, D' = min(D, 512)
    const int t = threadIdx.x;
    const int dcols = D < 512 ? D : 512;                 // columns covered per sweep by 1024 threads (2 h each)
    const int hh = t / dcols, dl = t - hh * dcols;       // hh in {0, 1} when dcols == 512
    const bool own = hh < 2;                             // D < 512: the threads beyond 2*D idle in the W1 part
    const AdamCoef ak = step_coef(a);
    // The parameters and moments this workgroup steps are requested further down, just before the gradient loop that
    // hides them, and W2 with the pairs' gathers: at the top they sat in front of the slide's scores -- the first thing
    // the kernel waits for, and loads return in issue order -- and held 12 registers through the pooling.
    int64_t base;
    const PoolLds L = {list, wmax, pooled_s, dpool, ncand, topk_s};
    const int k = pool_phase(a, b, L, PS_CAP, wg == 0, pooled_out, topk_idx_out, topk_cnt_out, &base);
    __syncthreads();
    MOC_STAMP(15);
    // ---- pairs: TWO rounds of gathers.  Round 1 -- everything that needs only a pair's position among the selected
    // rows (its row id, its gate operands, its hidden row), W2 and the parameters this workgroup steps -- is requested in
    // one go, straight from topk_s; round 2 the bag rows, a wave per pair (two pairs in flight per wave).  Sixteen waves
    // share one CU here: an instruction every thread executes costs the workgroup 16 issue cycles, so each part runs on
    // the waves that have work for it and nowhere else, and nothing divides.
    const int P = C * k;
    float pw[2][2], pm[2][2], pv[2][2];                  // [h sub-index][d sweep]  (D <= 1024): consumed by the Adam update
    float pW = 0.f, pM = 0.f, pV = 0.f;                  // workgroup 0: W2 / b2 / b1 element of this thread
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int sw = 0; sw < 2; ++sw) pw[j][sw] = pm[j][sw] = pv[j][sw] = 0.f;
    if (P > 0) {                                         // (uniform)
        // pairs per class k <= 16, pair numbers < 256: p / k = (p * ceil(2^16 / k)) >> 16 exactly
        const unsigned kinv = (65536u + (unsigned)k - 1u) / (unsigned)k;
        auto topk_of = [&](int pp) { const int c = (int)(((unsigned)pp * kinv) >> 16); return topk_s[c * K + (pp - c * k)]; };
        int64_t rid = 0;
        if (t < P) rid = a.sel_row[base + topk_of(t)];
        float sc = 0.f, lam = 0.f;
        const int dp = t >> 2, di = t & 3, dc = (int)(((unsigned)dp * kinv) >> 16);     // P <= 256: one element of dz per thread
        if (t < P * 4) {
            const int dsidx = topk_s[dc * K + (dp - dc * k)];
            const int dcol = di == 0 ? dc : di == 1 ? C + dc : di == 2 ? 2 * C : 2 * C + 1;
            sc = a.cand[(int64_t)dcol * a.stride + base + dsidx];
            lam = a.gates[(base + dsidx) * 4 + di];
        }
        float w2r = 0.f;
        if (t < 4 * H) w2r = a.W2[t];
        // H1 of the pairs: workgroup 0 needs all H columns (b1, W2), the others their 4
        const int hsh = wg == 0 ? 6 : JN, hn = 1 << hsh, h0 = wg == 0 ? 0 : h_lo, nh = P << hsh;   // (HW = 4, 2 = 1 << JN)
        float hv0 = 0.f, hv1 = 0.f;
        if (t < nh) hv0 = a.H1[(base + topk_of(t >> hsh)) * H + h0 + (t & (hn - 1))];
        if (t + 1024 < nh) hv1 = a.H1[(base + topk_of((t + 1024) >> hsh)) * H + h0 + (t & (hn - 1))];
        if (a.apply_adam) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int sw = 0; sw < 2; ++sw) {
                    const int d = dl + sw * dcols;
                    if (own && d < D && j < JN) {
                        const int e = (h_lo + hh * JN + j) * D + d;
                        pw[j][sw] = g.W1[e]; pm[j][sw] = g.m_W1[e]; pv[j][sw] = g.v_W1[e];
                    }
                }
            if (wg == 0) {
                if (t < 4 * H) { pM = a.m_W2[t]; pV = a.v_W2[t]; }
                else if (t < 4 * H + 4) { pW = a.b2[t - 4 * H]; pM = a.m_b2[t - 4 * H]; pV = a.v_b2[t - 4 * H]; }
                else if (t >= 320 && t < 320 + H) { pW = a.b1[t - 320]; pM = a.m_b1[t - 320]; pV = a.v_b1[t - 320]; }
            }
        }
        if (t < P) row_s[t] = rid;
        if (t < P * 4) {
            const float dlam = (a.use_bits >> di & 1u) ? (dpool[dc] / (float)k) * sc : 0.f;
            dz[t] = dlam * lam * (1.f - lam);
        }
        if (t < 4 * H) {
            W2s[t] = w2r;
            if (wg == 0) pW = w2r;
        }
        if (t < nh) H1s[(t >> hsh) * H + h0 + (t & (hn - 1))] = hv0;
        if (t + 1024 < nh) H1s[((t + 1024) >> hsh) * H + h0 + (t & (hn - 1))] = hv1;
        for (int e = t + 2048; e < nh; e += 1024) {        // workgroup 0 with more than 32 pairs
            const int pp = e >> hsh, h = h0 + (e & (hn - 1));
            H1s[pp * H + h] = a.H1[(base + topk_of(pp)) * H + h];
        }
        __syncthreads();
        // bag rows of the pairs -> fp32 in LDS: wave w takes pairs w, w + 16 (both in flight), w + 32, ...; a lane the
        // 16-byte (16-bit storage: 8-byte) pieces lane, lane + 64, ... of the row -- D / 256 of them
        const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
        const int dq = D >> 8;                             // 1..4
        const int esz = a.xdt == MOC_F32 ? 4 : 2;
        for (int p0 = wave; p0 < P; p0 += 32) {
            const int p1 = p0 + 16;
            const bool two = p1 < P;
            const unsigned char* r0 = a.X + row_s[p0] * (int64_t)D * esz;
            const unsigned char* r1 = a.X + row_s[two ? p1 : p0] * (int64_t)D * esz;
            float* x0 = xs + (size_t)p0 * D + lane * 4;
            float* x1 = xs + (size_t)p1 * D + lane * 4;
            if (a.xdt == MOC_F32) {
                // (named pieces: hipcc keeps an array written under a condition in scratch)
                float4 a0 = {}, a1 = {}, a2 = {}, a3 = {}, b0 = {}, b1v = {}, b2v = {}, b3 = {};
                const float4* q0 = reinterpret_cast<const float4*>(r0) + lane;
                const float4* q1 = reinterpret_cast<const float4*>(r1) + lane;
                a0 = q0[0]; b0 = q1[0];
                if (dq > 1) { a1 = q0[64]; b1v = q1[64]; }
                if (dq > 2) { a2 = q0[128]; b2v = q1[128]; }
                if (dq > 3) { a3 = q0[192]; b3 = q1[192]; }
                float4* y0 = reinterpret_cast<float4*>(x0);
                float4* y1 = reinterpret_cast<float4*>(x1);
                y0[0] = a0;
                if (dq > 1) y0[64] = a1;
                if (dq > 2) y0[128] = a2;
                if (dq > 3) y0[192] = a3;
                if (two) {
                    y1[0] = b0;
                    if (dq > 1) y1[64] = b1v;
                    if (dq > 2) y1[128] = b2v;
                    if (dq > 3) y1[192] = b3;
                }
            } else {
                const bool f16 = a.xdt == MOC_F16;
                auto widen = [&](uint2 raw) {
                    float4 o;
                    if (f16) {
                        o.x = moc_f16_to_f32((uint16_t)raw.x); o.y = moc_f16_to_f32((uint16_t)(raw.x >> 16));
                        o.z = moc_f16_to_f32((uint16_t)raw.y); o.w = moc_f16_to_f32((uint16_t)(raw.y >> 16));
                    } else {
                        o.x = __uint_as_float(raw.x << 16); o.y = __uint_as_float(raw.x & 0xFFFF0000u);
                        o.z = __uint_as_float(raw.y << 16); o.w = __uint_as_float(raw.y & 0xFFFF0000u);
                    }
                    return o;
                };
                uint2 u0[4], u1[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < dq) {
                        u0[i] = *reinterpret_cast<const uint2*>(r0 + (lane + i * 64) * 8);
                        u1[i] = *reinterpret_cast<const uint2*>(r1 + (lane + i * 64) * 8);
                    }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < dq) {
                        *reinterpret_cast<float4*>(x0 + i * 256) = widen(u0[i]);
                        if (two) *reinterpret_cast<float4*>(x1 + i * 256) = widen(u1[i]);
                    }
            }
        }
    }
    __syncthreads();
    MOC_STAMP(16);
    {   // dh[p][h] = (sum_i dz[p][i] * W2[i][h]) * [H1 > 0]  for this workgroup's columns
        const int hn = wg == 0 ? H : HW, h0 = wg == 0 ? 0 : h_lo;
        for (int e = t; e < P * hn; e += 1024) {
            const int p = e / hn, h = h0 + (e - p * hn);
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) v = fmaf(dz[p * 4 + i], W2s[i * H + h], v);
            dhs[p * H + h] = H1s[p * H + h] > 0.f ? v : 0.f;
        }
    }
    __syncthreads();
    MOC_STAMP(17);
    // ---- gradients of the owned elements: JN x 2 of W1 per thread, one of b1 / W2 / b2 in workgroup 0
    const int ha = h_lo + hh * JN;
    float gr[2][2] = {{0.f, 0.f}, {0.f, 0.f}};           // [d sweep][h sub-index]
    if (own) {
#pragma unroll
        for (int sw = 0; sw < 2; ++sw) {
            const int d = dl + sw * dcols;
            if (d < D) {
                float g0 = 0.f, g1 = 0.f;
                for (int p = 0; p < P; ++p) {
                    const float xv = xs[(size_t)p * D + d];
                    g0 = fmaf(dhs[p * H + ha], xv, g0);
                    if (JN > 1) g1 = fmaf(dhs[p * H + ha + 1], xv, g1);
                }
                gr[sw][0] = g0; gr[sw][1] = g1;
            }
        }
    }
    // tail element: flat order W1 | b1 | W2 | b2 (the order of the exchange slots)
    float gt = 0.f;
    int tail = -1;
    if (wg == 0) {
        if (t < 4 * H) {
            const int i = t >> 6, h = t & 63;
            for (int p = 0; p < P; ++p) gt = fmaf(dz[p * 4 + i], H1s[p * H + h], gt);
            tail = H + t;
        } else if (t < 4 * H + 4) {
            const int i = t - 4 * H;
            for (int p = 0; p < P; ++p) gt += dz[p * 4 + i];
            tail = H + 4 * H + i;
        } else if (t >= 320 && t < 320 + H) {
            const int h = t - 320;
            for (int p = 0; p < P; ++p) gt += dhs[p * H + h];
            tail = h;
        }
    }
    if (!a.apply_adam) {                                  // gradient out (data-parallel step with a collective)
        if (own)
            for (int sw = 0; sw < 2; ++sw) {
                const int d = dl + sw * dcols;
                if (d < D) { g.g_W1[(ha + 0) * D + d] = gr[sw][0]; if (JN > 1) g.g_W1[(ha + 1) * D + d] = gr[sw][1]; }
            }
        if (tail >= H + 4 * H) a.g_b2[tail - 5 * H] = gt;
        else if (tail >= H) a.g_W2[tail - H] = gt;
        else if (tail >= 0) a.g_b1[tail] = gt;
        return;
    }
    if (g.x.world > 1) {                                  // one-shot exchange over xGMI (moc_p2p.h)
        __shared__ int p2p_ok;
        const int64_t nW1 = (int64_t)H * D;
        if (own)
#pragma unroll
            for (int sw = 0; sw < 2; ++sw) {
                const int d = dl + sw * dcols;
                if (d < D) { p2p_push(g.x, (int64_t)(ha + 0) * D + d, gr[sw][0]); if (JN > 1) p2p_push(g.x, (int64_t)(ha + 1) * D + d, gr[sw][1]); }
            }
        if (tail >= 0) p2p_push(g.x, nW1 + tail, gt);
        if (!p2p_signal_wait(g.x, wg, &p2p_ok)) return;   // time-out: reported through g.x.error, no update
        if (own)
#pragma unroll
            for (int sw = 0; sw < 2; ++sw) {
                const int d = dl + sw * dcols;
                if (d < D) {
                    gr[sw][0] = p2p_sum(g.x, (int64_t)(ha + 0) * D + d, gr[sw][0]);
                    if (JN > 1) gr[sw][1] = p2p_sum(g.x, (int64_t)(ha + 1) * D + d, gr[sw][1]);
                }
            }
        if (tail >= 0) gt = p2p_sum(g.x, nW1 + tail, gt);
    }
    // ---- Adam: parameter, moments, operand image
    const float gs = ak.grad_scale;
    if (own) {
#pragma unroll
        for (int sw = 0; sw < 2; ++sw) {
            const int d = dl + sw * dcols;
            if (d < D) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (j >= JN) continue;
                    const int e = (ha + j) * D + d;
                    adam_update(pw[j][sw], pm[j][sw], pv[j][sw], gr[sw][j] * gs, ak);
                    g.W1[e] = pw[j][sw]; g.m_W1[e] = pm[j][sw]; g.v_W1[e] = pv[j][sw];
                    w1_image_store(g.img_dt, g.W1img, D, ha + j, d, pw[j][sw]);
                }
            }
        }
    }
    if (tail >= 0) {                                      // workgroup 0: b1, W2 (into the other buffer), b2
        adam_update(pW, pM, pV, gt * gs, ak);
        if (tail >= H + 4 * H) { const int i = tail - 5 * H; a.b2[i] = pW; a.m_b2[i] = pM; a.v_b2[i] = pV; }
        else if (tail >= H) { const int i = tail - H; g.W2out[i] = pW; a.m_W2[i] = pM; a.v_W2[i] = pV; }
        else { a.b1[tail] = pW; a.m_b1[tail] = pM; a.v_b1[tail] = pV; }
    }
    MOC_STAMP(18);
}

// ---- the one-launch step over the forward's tile records (round 4) ----------------------------------------------
// Same job as pool_w1_step_kernel -- pooling, loss, backward, the whole Adam step in one launch, every workgroup
// repeating the pooling for itself -- restructured around what the phase stamps of round 4 showed (profiles/NOTES.md):
// the step is a chain of dependent memory round trips and of dependent instructions; neither gets shorter by adding
// threads, only by taking round trips and instructions out of the chain.
//  * A workgroup is FOUR waves and owns 256 columns of ONE hidden unit of W1, or -- the last column block, the "tail"
//    workgroup of the hidden unit -- b1[h] and W2[:, h] (h = 0: b2 too) and no column (grid (D / 256 + 1) x H: 192
//    workgroups at D = 512).  Every sum over the pairs runs in the same order as in pool_w1_step_kernel: the same bits.
//  * The kernel's arguments are its own compact struct, ordered by first use, and every cache line of them is touched
//    at the top in ONE wait: the general step's 700 bytes of arguments were five scalar-cache misses one after the other
//    in front of the first load.
//  * The pooling of class c is ONE wave's business from the first load to the pooled rows, with no workgroup barrier in
//    between: it reads the class's tile records (above; VQ keys per lane), forms the sixteen group maxima in registers
//    (32-bit score keys: quad maxima, one ds_bpermute, fifteen row rotations for the ranks), lists the records >= T0,
//    requests the candidates' row ids / gates / candidate scores -- one round trip that the ranking loop hides -- and
//    leaves the pooled rows' operands in LDS.  No second gather round for them.
//  * Then ONE more round trip: the pairs' 256-column row pieces (three waves), their hidden activation of unit h,
//    while the fourth wave does the cross entropy; the parameters this workgroup steps were requested at the top.
// Falls back to the full scores inside the kernel when the records cannot prove exactness.
struct TileStepArgs {
    // ---- first cache line (64 bytes): what the first loads need -- they are issued before the other lines are touched
    const unsigned long long* tkey;
    const uint32_t* trho;
    int64_t slot0;
    int cap, ntile_bound, C, K, D, slide0;
    int n_runs, slide_stride;       // batched runs (n_runs > 0): see the end of the struct
    const int32_t* n_sel;
    // ---- what this thread steps
    const int64_t* labels;
    int PS_CAP, xdt;
    int tail_inside;                // != 0: no tail workgroups (grid.x = D / 256), column block 0 owns the hidden unit's small elements too
    int S_host;                     // the slide's selected-row count when the host knows it (one run's launch; -1: n_sel[b])
    float *W1, *m_W1, *v_W1;
    const float* W2;
    float *b1, *m_b1, *v_b1, *b2, *m_b2, *v_b2, *m_W2, *v_W2;
    int64_t base;
    AdamCoef adam;
    const AdamCoef* adam_tab;       // graph replay: coefficients adam_tab[adam_ctr[0] + adam_pos]
    const int32_t* adam_ctr;
    int adam_pos, apply_adam;
    uint32_t use_bits;
    int img_dt;
    // ---- the candidates' operands, round trip 2
    const int64_t* trid;
    const float4* tlam;
    const float4* tsc;
    const float* H1;
    const unsigned char* X;
    // ---- outputs
    float* pooled_out;
    int32_t* topk_idx_out;
    int32_t* topk_cnt_out;
    float* loss;
    int32_t* pred;
    int32_t* n_pair;
    unsigned char* W1img;
    float* W2out;
    float *g_W1, *g_b1, *g_W2, *g_b2;
    // ---- fall-back
    const float* mixed_in;
    const float* cand;
    const float* gates;
    const int64_t* sel_row;
    int64_t stride;
    // ---- batched runs (n_runs > 0): grid.z = run; run r steps the meta-learner whose tensors lie par_stride floats (W2 in:
    // w2_stride, W2 out: w2out_stride, operand image: img_stride bytes) behind run 0's, on slide slide0 + r * slide_stride
    int64_t par_stride, w2_stride, w2out_stride, img_stride;
    // ---- the slide the NEXT step works on (slide b + 1 of the same work arrays): its selected rows are pulled toward the
    // Infinity Cache by the waves that have no class to pool
    // (the slide's first slot is read from the device's row_off by the prefetching waves themselves: a per-run array of them
    // made the argument segment 984 bytes, and the two more lines of it cost every launch 0.25 us)
    const int64_t* row_off;
    int prefetch_next;              // != 0: slide b + 1 follows in the same work arrays
    int sink_off;                   // byte offset of 1 KiB of LDS nobody reads (the prefetch's LDS-DMA destination)
};
// The launch of batched runs carries the per-run scalars behind the common block; ONE run's launch does not: the argument
// segment lives in host memory and every 64-byte line of it that the kernel touches costs its start ~0.1 us (984 -> 864
// bytes: alone 16.70 -> 16.45 us per step; without the 384 bytes of per-run arrays: see profiles/NOTES.md).
struct TileStepArgsRuns {
    TileStepArgs s;
    int64_t base_r[MOC_MAX_RUNS], slot0_r[MOC_MAX_RUNS];
    int32_t cap_r[MOC_MAX_RUNS], ntb_r[MOC_MAX_RUNS];
};
static_assert(sizeof(TileStepArgsRuns) <= 1024, "a kernel-argument segment over 1 KiB takes a slow launch path (profiles/NOTES.md)");
__host__ __device__ __forceinline__ const TileStepArgs& tile_common(const TileStepArgs& x) { return x; }
__host__ __device__ __forceinline__ const TileStepArgs& tile_common(const TileStepArgsRuns& x) { return x.s; }

template <int VQ, typename ArgsT>
__global__ __launch_bounds__(256, 4) void pool_w1_step_tiles_kernel(ArgsT args) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool RUNS = std::is_same<ArgsT, TileStepArgsRuns>::value;
    const TileStepArgs& a = tile_common(args);
    static_assert(offsetof(TileStepArgs, n_sel) + 8 <= 64, "the first loads' arguments must share the first cache line");
    // batched runs: this workgroup's run -- its slide, its region of the records, the offsets of its tensors (the argument
    // block itself is not modified and its arrays are read through kernarg_at: either would send all of it to scratch)
    int b = a.slide0;
    int64_t base = 0, slot0 = a.slot0, po = 0, w2o = 0, w2outo = 0, imgo = 0;
    int cap = a.cap, ntile_bound = a.ntile_bound;
    constexpr bool runs = RUNS;
    if constexpr (RUNS) {
        const int run = blockIdx.z;
        b += run * a.slide_stride;
        slot0 = kernarg_at<int64_t>(offsetof(TileStepArgsRuns, slot0_r) + 8 * (size_t)run);
        cap = kernarg_at<int32_t>(offsetof(TileStepArgsRuns, cap_r) + 4 * (size_t)run);
        ntile_bound = kernarg_at<int32_t>(offsetof(TileStepArgsRuns, ntb_r) + 4 * (size_t)run);
    }
    const int C = a.C, K = a.K, D = a.D;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int cb = blockIdx.x, h = blockIdx.y;
    // The last column block is the TAIL workgroup of its hidden unit: it owns b1[h] and W2[:, h] (h = 0: b2 too) and no
    // column of W1 -- on a W1 workgroup those few elements' sums and second Adam update ran after the wave's own W1 work
    // (Batched runs of four or more keep the small elements on column block 0 instead -- tail_inside: R x 128 workgroups of
    // which a CU holds four, so that eight runs are ONE round of workgroups and not two; a few threads' second sum and
    // second Adam update behind their W1 work cost a single run 0.6 us, a launch of eight runs wins 8 us.)
    const bool tail_wg = !a.tail_inside && cb == (D >> 8); // (grid.x = D / 256 + 1; not gridDim: that pulls 256 bytes of hidden arguments into the segment)
    const bool tail_role = a.tail_inside ? cb == 0 : tail_wg;
    const bool wg0 = cb == 0 && h == 0;                                              // (of its run: publishes the slide's outputs)
    constexpr int CL = 64;                                                            // candidates per class
    MOC_STAMP(10);
    MOC_STAMP_MIN(33);
    MOC_STAMP_MAX(34);
    // ---- round trip 1, requested before anything else -- before the other argument lines are even touched: the keys of
    // this wave's first class (wave w: classes w, w + 4, ...)
    unsigned long long key[VQ];
    uint32_t rho[VQ / 4];
#define MOC_TILE_KEYS(c)                                                                                              \
    {                                                                                                                 \
        const int64_t s_c_ = slot0 + (int64_t)(c) * cap * TILE_R;                                                     \
        const int nrec_b_ = ntile_bound * TILE_R;                                                                     \
        _Pragma("unroll") for (int q = 0; q < VQ; ++q) {                                                              \
            const int j = q * 64 + lane;                                                                              \
            key[q] = a.tkey[s_c_ + (j < nrec_b_ ? j : nrec_b_ - 1)];                                                  \
        }                                                                                                             \
        _Pragma("unroll") for (int rq = 0; rq < VQ / 4; ++rq) {                                                       \
            const int x = rq * 64 + lane;                                                                             \
            rho[rq] = a.trho[slot0 / TILE_R + (int64_t)(c) * cap + (x < ntile_bound ? x : ntile_bound - 1)];          \
        }                                                                                                             \
    }
    if (wave < C) MOC_TILE_KEYS(wave)
    __builtin_amdgcn_sched_barrier(0);
    moc_kernarg_touch<sizeof(ArgsT)>();                  // every other line of the arguments, side by side, one wait
    const int PS_CAP = a.PS_CAP;
    base = a.base;
    if constexpr (RUNS) {
        const int run = blockIdx.z;
        base = kernarg_at<int64_t>(offsetof(TileStepArgsRuns, base_r) + 8 * (size_t)run);
        po = (int64_t)run * a.par_stride; w2o = (int64_t)run * a.w2_stride; w2outo = (int64_t)run * a.w2out_stride;
        imgo = (int64_t)run * a.img_stride;
    }
    // ... then what this thread steps (independent of the pooling; it queues behind the keys).  The small tensors' owners
    // sit in different waves: their sums over the pairs run side by side, not one after the other in one wave
    const int d = cb * 256 + t;                                                      // this thread's column of W1[h]
    float pw = 0.f, pm = 0.f, pv = 0.f;
    float pT = 0.f, pTm = 0.f, pTv = 0.f;                                            // this thread's element of W2 / b1 / b2
    int tail = -1;                                                                   // flat index past W1: b1 | W2 | b2
    if (tail_role) {
        if (t < 4) tail = H + t * H + h;                                             // W2[i = t][h]: wave 0
        else if (t == 64) tail = h;                                                  // b1[h]: wave 1
        else if (h == 0 && t >= 128 && t < 132) tail = H + 4 * H + (t - 128);        // b2[i]: wave 2
    }
    float w2v = 0.f;
    if (t < 4) w2v = a.W2[w2o + t * H + h];
    if (a.apply_adam) {
        if (!tail_wg) {
            const int64_t e = po + h * D + d;
            pw = a.W1[e]; pm = a.m_W1[e]; pv = a.v_W1[e];
        }
        if (tail >= H + 4 * H) { const int64_t i = po + tail - 5 * H; pT = a.b2[i]; pTm = a.m_b2[i]; pTv = a.v_b2[i]; }
        else if (tail >= H) { const int64_t i = po + tail - H; pTm = a.m_W2[i]; pTv = a.v_W2[i]; }
        else if (tail >= 0) { pT = a.b1[po + tail]; pTm = a.m_b1[po + tail]; pTv = a.v_b1[po + tail]; }
    }
    MOC_STAMP(19);
    // ---- the next slide's selected rows, its row ids and its candidate columns: one dword of every 128-byte line, by the
    // waves with no class to pool (C < 4), results never used.  The forward of the next step then finds them in the
    // Infinity Cache: with 492 MB of score pass streaming through that cache every pass, nothing of a slide is left in it
    // from the pass before (a 1-GB copy between two passes of steps with nothing else on the GPU: 16.8 -> 18.5 us per
    // step, scripts/diag_mall.py).  Inline asm: the compiler's wait insertion does not see these loads, nobody waits for
    // them (s_endpgm does).
    if (wave >= C && a.prefetch_next) {
        const int64_t nb = a.row_off[b + 1];
        const int S2 = a.n_sel[b + 1];
        const int esz_p = a.xdt == MOC_F32 ? 4 : 2;
        const int lpr = D * esz_p / 128;                                            // lines per row
        const int iw = 4 - C;                                                       // waves of a workgroup that come here
        const int gx = (D >> 8) + (a.tail_inside ? 0 : 1);
        const int g0 = ((h * gx + cb) * iw + (wave - C)) * 64 + lane, G = gx * H * iw * 64;
        // (LDS-DMA into 256 sacrificial bytes per wave: a load into a register that nobody reads leaves the compiler free to
        // re-use that register while the load is still in flight -- the first form of this did, and faulted)
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        float* sinkw = reinterpret_cast<float*>(smem + a.sink_off) + (wave - C) * 64;
        for (int L = g0; L < S2 * lpr; L += G) {
            const int r = L / lpr;
            const unsigned char* pl = a.X + a.sel_row[nb + r] * (int64_t)D * esz_p + (int64_t)(L - r * lpr) * 128;
            __builtin_amdgcn_global_load_lds((gptr_t)pl, (lptr_t)sinkw, 4, 0, 0);
        }
        const int ncol = 2 * C + 2, lpc = (S2 + 31) >> 5;                           // candidate columns, lines per column
        for (int L = g0; L < ncol * lpc; L += G) {
            const int c = L / lpc;
            const float* pl = a.cand + (int64_t)c * a.stride + nb + (int64_t)(L - c * lpc) * 32;
            __builtin_amdgcn_global_load_lds((gptr_t)pl, (lptr_t)sinkw, 4, 0, 0);
        }
    }
    AdamCoef ak = a.adam;
    if (a.adam_tab) ak = a.adam_tab[a.adam_ctr[0] + a.adam_pos];
    const int S = (!RUNS && a.S_host >= 0) ? a.S_host : a.n_sel[b];
    const int y = (int)a.labels[b];
    const int k = K < S ? K : S;
    // LDS carve: the 16-byte things first
    float4* plam = reinterpret_cast<float4*>(smem);                                  // [P] pairs' gates
    float4* psc = plam + (size_t)C * K;                                              // [P] ... candidate scores
    float* xs = reinterpret_cast<float*>(psc + (size_t)C * K);                       // [P][256] pairs' row pieces
    float* dhs = xs + (size_t)C * K * 256;                                           // [P (+3)] dh of unit h, 16-byte aligned
    float* dz = dhs + (((size_t)C * K + 3) & ~(size_t)3);                            // [P][4]
    unsigned long long* list = reinterpret_cast<unsigned long long*>(dz + (size_t)C * K * 4);   // [C][PS_CAP]
    unsigned long long* t0s = list + (size_t)C * PS_CAP;                             // [C] the bound T0 (as a key)
    int64_t* prid = reinterpret_cast<int64_t*>(t0s + C);                             // [P]
    int* lsrc = reinterpret_cast<int*>(prid + (size_t)C * K);                        // [C][CL] record of a candidate
    float* topv = reinterpret_cast<float*>(lsrc + (size_t)C * CL);                   // [C][16]
    float* pooled_s = topv + (size_t)C * 16;                                         // [C]
    float* dpool = pooled_s + C;                                                     // [C]
    int* ncand = reinterpret_cast<int*>(dpool + C);                                  // [C]
    int* flagc = ncand + C;                                                          // [C] 1: the records cannot prove exactness
    int* topk_s = flagc + C;                                                         // [C][K]
    float* h1s = reinterpret_cast<float*>(topk_s + C * K);                           // [P]
    float* w2s = h1s + (size_t)C * K;                                                // [4]
    if (t < 4) { w2s[t] = w2v; if (tail >= H) pT = w2v; }
    const int ntile = (S + 15) >> 4;
    for (int c = wave; c < C; c += 4) {                                              // ---- class c: this wave's, start to end
        const int64_t s_c = slot0 + (int64_t)c * cap * TILE_R;                       // the class's first record
        if (c != wave) MOC_TILE_KEYS(c)
        if (lane == 0) ncand[c] = 0;
        // the score halves of the keys decide bound and candidates (32-bit maxima and compares); absent records: 0
        uint32_t hi[VQ];
        uint32_t m = 0u;
        const int nrec = ntile * TILE_R;
#pragma unroll
        for (int q = 0; q < VQ; ++q) {
            hi[q] = q * 64 + lane < nrec ? (uint32_t)(key[q] >> 32) : 0u;
            m = hi[q] > m ? hi[q] : m;
        }
        MOC_STAMP(20);
        // group g = the tiles g, g + 16, ...: record j sits on lane j & 63, its tile is j >> 2 -- the four lanes 4g .. 4g + 3
        // hold group g, whatever q.  Quad maxima, then every row gathers the sixteen of them.
        {
            const uint32_t o1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
            m = o1 > m ? o1 : m;
            const uint32_t o2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
            m = o2 > m ? o2 : m;
        }
        const uint32_t gm = (uint32_t)__shfl((int)m, (lane & 15) * 4, 64);
        int grank = 0;
#define MOC_ROR(n) grank += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)gm, 0x120 + n, 0xf, 0xf, false) > gm ? 1 : 0;
        MOC_ROR(1) MOC_ROR(2) MOC_ROR(3) MOC_ROR(4) MOC_ROR(5) MOC_ROR(6) MOC_ROR(7) MOC_ROR(8)
        MOC_ROR(9) MOC_ROR(10) MOC_ROR(11) MOC_ROR(12) MOC_ROR(13) MOC_ROR(14) MOC_ROR(15)
#undef MOC_ROR
        // T0 = the K-th largest group maximum: K scores are >= it.  (0: fewer than K groups hold a record, or two group
        // maxima tie across the K-th place -- then every record is a candidate.)
        const unsigned long long hit = __ballot(lane < 16 && grank == k - 1 && gm != 0u);
        uint32_t T0 = 0u;
        if (hit != 0ull) T0 = (uint32_t)__builtin_amdgcn_readlane((int)gm, __ffsll((long long)hit) - 1);
        // the records >= T0: a lane with any takes its places in the list by ONE LDS add (few lanes have any)
        // (an absent record's key half is 0: with the bound raised to 1 one compare tells both; the sixty-fours of records
        // past the slide's last tile are skipped by uniform branches)
        const uint32_t T1 = T0 > 1u ? T0 : 1u;
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < VQ; ++q)
            if (q * 64 < nrec) cnt += hi[q] >= T1 ? 1 : 0;
        if (cnt > 0) {
            int pos = atomicAdd(&ncand[c], cnt);
#pragma unroll
            for (int q = 0; q < VQ; ++q) {
                if (q * 64 < nrec && hi[q] >= T1) {
                    if (pos < CL) {
                        list[(size_t)c * PS_CAP + pos] = key[q];
                        lsrc[c * CL + pos] = q * 64 + lane;
                    }
                    ++pos;
                }
            }
        }
        // ... and the proof that no score >= T0 is missing from the records: no tile's (TILE_R + 1)-th largest reaches T0
        bool bad = false;
#pragma unroll
        for (int rq = 0; rq < VQ / 4; ++rq)
            bad = bad || (rq * 64 + lane < ntile && rho[rq] != 0u && rho[rq] >= T0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int n = ncand[c];
        const bool fall = __ballot(bad) != 0ull || n > CL || n < k || k <= 0;
        if (lane == 0) { flagc[c] = fall ? 1 : 0; t0s[c] = (unsigned long long)T0 << 32; }
        MOC_STAMP(11);
        if (!fall) {                                                                 // (uniform over the wave)
            const unsigned long long mine = lane < n ? list[(size_t)c * PS_CAP + lane] : 0ull;
            const int jrec = lane < n ? lsrc[c * CL + lane] : 0;
            // the candidate's operands: one round trip, under the ranking loop
            const int64_t rid = a.trid[s_c + jrec];
            const float4 lam = a.tlam[s_c + jrec];
            const float4 scv = a.tsc[s_c + jrec];
            int rank = 0;
            for (int l = 0; l < n; ++l) rank += list[(size_t)c * PS_CAP + l] > mine ? 1 : 0;
            MOC_STAMP(12);
            if (lane < n && rank < k) {
                topk_s[c * K + rank] = (int)(~(unsigned)mine);
                topv[c * 16 + rank] = key_to_float((unsigned)(mine >> 32));
                const int p = c * k + rank;
                prid[p] = rid; plam[p] = lam; psc[p] = scv;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {                                                         // summed largest first
                float sum = 0.f;
                for (int r = 0; r < k; ++r) sum += topv[c * 16 + r];
                pooled_s[c] = sum / (float)k;
                if (wg0) {
                    a.pooled_out[(int64_t)b * C + c] = pooled_s[c];
                    if (a.topk_cnt_out) a.topk_cnt_out[(int64_t)b * C + c] = k;
                }
            }
            if (a.topk_idx_out && wg0)
                for (int r = lane; r < K; r += 64)
                    a.topk_idx_out[((int64_t)b * C + c) * K + r] = r < k ? topk_s[c * K + r] : -1;
        }
    }
    __syncthreads();
    MOC_STAMP(13);
    bool fast = true;
    for (int c = 0; c < C; ++c) fast = fast && flagc[c] == 0;
    if (wg0 && t == 0 && a.n_pair) {                     // (which path ran: tests, diagnostics; with runs: the largest code of any)
        if (a.n_runs > 0) atomicMax(a.n_pair, fast ? MOC_TILE_PATH_RECORDS : MOC_TILE_PATH_FULL);
        else *a.n_pair = fast ? MOC_TILE_PATH_RECORDS : MOC_TILE_PATH_FULL;
    }
    if (!fast) {
        // ---- fall-back: the same bound T0 (a valid one: K group maxima are K scores >= T0) over ALL mixed scores of the
        // slide.  Compact code, not fast code: it runs when a tile holds more than TILE_R of the candidates.
        if (t < C) ncand[t] = 0;
        __syncthreads();
        for (int c = 0; c < C; ++c) {
            const unsigned long long T0 = t0s[c];
            const float* col = a.mixed_in + (int64_t)c * a.stride + base;
            for (int i = t; i < S; i += 256) {
                const unsigned long long kk = ((unsigned long long)moc_key_desc(col[i]) << 32) | (unsigned)(~(unsigned)i);
                if (kk >= T0) {
                    const int pos = atomicAdd(&ncand[c], 1);
                    if (pos < PS_CAP) list[(size_t)c * PS_CAP + pos] = kk;
                }
            }
        }
        __syncthreads();
        for (int c = wave; c < C; c += 4) {                                          // k rounds of wave maximum
            const int n = ncand[c];
            const bool overflow = n > PS_CAP;                                        // pathological ties: straight from global
            const float* col = a.mixed_in + (int64_t)c * a.stride + base;
            unsigned long long prev = ~0ull;
            float sum = 0.f;
            for (int r = 0; r < k; ++r) {
                unsigned long long best = 0ull;
                if (!overflow) {
                    for (int i = lane; i < n; i += 64) {
                        const unsigned long long v = list[(size_t)c * PS_CAP + i];
                        if (v < prev && v > best) best = v;
                    }
                } else {
                    for (int i = lane; i < S; i += 64) {
                        const unsigned long long v = ps_key(col, i, S);
                        if (v < prev && v > best) best = v;
                    }
                }
                best = wave_max_u64(best);
                prev = best;
                sum += key_to_float((unsigned)(best >> 32));
                if (lane == 0) topk_s[c * K + r] = (int)(~(unsigned)best);
            }
            if (lane == 0) pooled_s[c] = k > 0 ? sum / (float)k : __uint_as_float(0x7FC00000u);   // mean over no rows = NaN, like torch
            __builtin_amdgcn_wave_barrier();
            if (lane == 0 && wg0) {
                a.pooled_out[(int64_t)b * C + c] = pooled_s[c];
                if (a.topk_cnt_out) a.topk_cnt_out[(int64_t)b * C + c] = k;
            }
            if (a.topk_idx_out && wg0)
                for (int r = lane; r < K; r += 64)
                    a.topk_idx_out[((int64_t)b * C + c) * K + r] = r < k ? topk_s[c * K + r] : -1;
        }
        __syncthreads();
        const int P0 = C * k;
        for (int p = t; p < P0; p += 256) {                                          // the pooled rows' operands, gathered
            const unsigned kinv0 = (65536u + (unsigned)k - 1u) / (unsigned)k;
            const int c = (int)(((unsigned)p * kinv0) >> 16);
            const int sidx = topk_s[c * K + (p - c * k)];
            prid[p] = a.sel_row[base + sidx];
            plam[p] = reinterpret_cast<const float4*>(a.gates)[base + sidx];
            float4 s4;
            s4.x = a.cand[(int64_t)c * a.stride + base + sidx];
            s4.y = a.cand[(int64_t)(C + c) * a.stride + base + sidx];
            s4.z = a.cand[(int64_t)(2 * C) * a.stride + base + sidx];
            s4.w = a.cand[(int64_t)(2 * C + 1) * a.stride + base + sidx];
            psc[p] = s4;
        }
        __syncthreads();
    }
    MOC_STAMP(14);
    MOC_STAMP(15);
    // ---- round trip 2: the pairs' row pieces and their hidden activation of unit h
    const int P = C * k;
    if (P > 0) {                                                                     // (uniform)
        const unsigned kinv = (65536u + (unsigned)k - 1u) / (unsigned)k;
        if (wave == 3) {                                                             // the pairs' hidden activation of unit h
            for (int p = lane; p < P; p += 64) {
                const int c = (int)(((unsigned)p * kinv) >> 16);
                h1s[p] = a.H1[(base + topk_s[c * K + (p - c * k)]) * H + h];
            }
        } else if (!tail_wg) {                                                       // row pieces -> fp32 in LDS: eight in flight per wave
            // (twenty-four rows per round over the three waves: K = 10, C = 2 is ONE round trip; with four per wave it was two)
            const int esz = a.xdt == MOC_F32 ? 4 : 2;
            const int64_t col0 = (int64_t)cb * 256 * esz;
            constexpr int RF = 8;
            for (int p0 = wave; p0 < P; p0 += 3 * RF) {
                const unsigned char* rp[RF];
#pragma unroll
                for (int u = 0; u < RF; ++u) {
                    const int pu = p0 + 3 * u;
                    rp[u] = a.X + prid[pu < P ? pu : p0] * (int64_t)D * esz + col0;
                }
                float4 v[RF];
                if (a.xdt == MOC_F32) {
#pragma unroll
                    for (int u = 0; u < RF; ++u) v[u] = reinterpret_cast<const float4*>(rp[u])[lane];
                } else {
                    const bool f16 = a.xdt == MOC_F16;
                    auto widen = [&](uint2 raw) {
                        float4 o;
                        if (f16) {
                            o.x = moc_f16_to_f32((uint16_t)raw.x); o.y = moc_f16_to_f32((uint16_t)(raw.x >> 16));
                            o.z = moc_f16_to_f32((uint16_t)raw.y); o.w = moc_f16_to_f32((uint16_t)(raw.y >> 16));
                        } else {
                            o.x = __uint_as_float(raw.x << 16); o.y = __uint_as_float(raw.x & 0xFFFF0000u);
                            o.z = __uint_as_float(raw.y << 16); o.w = __uint_as_float(raw.y & 0xFFFF0000u);
                        }
                        return o;
                    };
                    uint2 raw[RF];
#pragma unroll
                    for (int u = 0; u < RF; ++u) raw[u] = reinterpret_cast<const uint2*>(rp[u])[lane];
#pragma unroll
                    for (int u = 0; u < RF; ++u) v[u] = widen(raw[u]);
                }
#pragma unroll
                for (int u = 0; u < RF; ++u) {
                    const int pu = p0 + 3 * u;
                    if (pu < P) reinterpret_cast<float4*>(xs + (size_t)pu * 256)[lane] = v[u];
                }
            }
        }
    }
    // cross entropy of the pooled logits, argmax and d loss / d pooled: the fourth wave, beside the others' row requests
    if (wave == 3) {
        const float xv = lane < C ? pooled_s[lane] : -INFINITY;
        float mx = xv;
        int arg = lane < C ? lane : 0x7fffffff;
        for (int off = 8; off > 0; off >>= 1) {
            const float om = __shfl_xor(mx, off, 64);
            const int oa = __shfl_xor(arg, off, 64);
            if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
        }
        const float ex = lane < C ? expf(xv - mx) : 0.f;
        float se = ex;
        for (int off = 8; off > 0; off >>= 1) se += __shfl_xor(se, off, 64);
        const float lse = mx + logf(se);
        if (lane < C) dpool[lane] = expf(xv - lse) - (lane == y ? 1.f : 0.f);
        if (lane == 0 && wg0) {
            a.loss[b] = lse - pooled_s[y];
            a.pred[b] = arg;
        }
    }
    __syncthreads();
    MOC_STAMP(16);
    for (int p = t; p < P; p += 256) {                   // thread = pair: its dz, and dh of hidden unit h
        const unsigned kinv = (65536u + (unsigned)k - 1u) / (unsigned)k;
        const int c = (int)(((unsigned)p * kinv) >> 16);
        const float gk = dpool[c] / (float)k;
        const float4 l4 = plam[p], s4 = psc[p];
        const float lamv[4] = {l4.x, l4.y, l4.z, l4.w}, scs[4] = {s4.x, s4.y, s4.z, s4.w};
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float dlam = (a.use_bits >> i & 1u) ? gk * scs[i] : 0.f;
            const float dzi = dlam * lamv[i] * (1.f - lamv[i]);
            dz[p * 4 + i] = dzi;
            v = fmaf(dzi, w2s[i], v);
        }
        dhs[p] = h1s[p] > 0.f ? v : 0.f;
    }
    __syncthreads();
    MOC_STAMP(17);
    // ---- gradients of the owned elements, every sum over the pairs in ascending order (operands four pairs at a time:
    // the additions stay in order, the LDS reads do not wait for each other)
    float gr = 0.f;
    if (!tail_wg) {
        int p = 0;
        for (; p + 4 <= P; p += 4) {
            const float4 d4 = *reinterpret_cast<const float4*>(dhs + p);
            const float x0 = xs[(size_t)p * 256 + t], x1 = xs[(size_t)(p + 1) * 256 + t];
            const float x2 = xs[(size_t)(p + 2) * 256 + t], x3 = xs[(size_t)(p + 3) * 256 + t];
            gr = fmaf(d4.x, x0, gr); gr = fmaf(d4.y, x1, gr); gr = fmaf(d4.z, x2, gr); gr = fmaf(d4.w, x3, gr);
        }
        for (; p < P; ++p) gr = fmaf(dhs[p], xs[(size_t)p * 256 + t], gr);
    }
    float gt = 0.f;
    if (tail >= H + 4 * H) {
        const int i = tail - 5 * H;
        int p = 0;
        for (; p + 4 <= P; p += 4) {
            const float z0 = dz[p * 4 + i], z1 = dz[(p + 1) * 4 + i], z2 = dz[(p + 2) * 4 + i], z3 = dz[(p + 3) * 4 + i];
            gt += z0; gt += z1; gt += z2; gt += z3;
        }
        for (; p < P; ++p) gt += dz[p * 4 + i];
    } else if (tail >= H) {
        int p = 0;
        for (; p + 4 <= P; p += 4) {
            const float z0 = dz[p * 4 + t], z1 = dz[(p + 1) * 4 + t], z2 = dz[(p + 2) * 4 + t], z3 = dz[(p + 3) * 4 + t];
            const float4 h4 = {h1s[p], h1s[p + 1], h1s[p + 2], h1s[p + 3]};
            gt = fmaf(z0, h4.x, gt); gt = fmaf(z1, h4.y, gt); gt = fmaf(z2, h4.z, gt); gt = fmaf(z3, h4.w, gt);
        }
        for (; p < P; ++p) gt = fmaf(dz[p * 4 + t], h1s[p], gt);
    } else if (tail >= 0) {
        int p = 0;
        for (; p + 4 <= P; p += 4) {
            const float4 d4 = *reinterpret_cast<const float4*>(dhs + p);
            gt += d4.x; gt += d4.y; gt += d4.z; gt += d4.w;
        }
        for (; p < P; ++p) gt += dhs[p];
    }
    if (!a.apply_adam) {                                  // gradient out (data-parallel step with a collective)
        if (!tail_wg) a.g_W1[h * D + d] = gr;
        if (tail >= H + 4 * H) a.g_b2[tail - 5 * H] = gt;
        else if (tail >= H) a.g_W2[tail - H] = gt;
        else if (tail >= 0) a.g_b1[tail] = gt;
        return;
    }
    const float gs = ak.grad_scale;
    if (!tail_wg) {
        const int64_t e = po + h * D + d;
        adam_update(pw, pm, pv, gr * gs, ak);
        a.W1[e] = pw; a.m_W1[e] = pm; a.v_W1[e] = pv;
        w1_image_store(a.img_dt, a.W1img + imgo, D, h, d, pw);
    }
    if (tail >= 0) {
        adam_update(pT, pTm, pTv, gt * gs, ak);
        if (tail >= H + 4 * H) { const int64_t i = po + tail - 5 * H; a.b2[i] = pT; a.m_b2[i] = pTm; a.v_b2[i] = pTv; }
        else if (tail >= H) { const int i = tail - H; a.W2out[w2outo + i] = pT; a.m_W2[po + i] = pTm; a.v_W2[po + i] = pTv; }
        else { a.b1[po + tail] = pT; a.m_b1[po + tail] = pTm; a.v_b1[po + tail] = pTv; }
    }
    MOC_STAMP(18);
    MOC_STAMP_MAX(35);
}
#undef MOC_TILE_KEYS

// ------------------------------------------------------------------ one-launch step, wide shapes
// C <= 64, K <= 16, S <= 8192 (EBRAINS-30, the 64-way stress shape): hundreds of gradient pairs, whose bag
// rows (P x D) and hidden activations (P x 64) do not fit in LDS.  Same grid as pool_w1_step_kernel (16
// workgroups of 1024 threads, every one repeats the pooling), different split of the backward pass:
// workgroup g owns the D/16 COLUMNS [g*D/16, ...) of W1 for all 64 hidden units, so it needs only a
// D/16-wide piece of every pair's row (64 B at D = 512 bf16: one gather round, P x 64 B of LDS), and
// instead of dh[P][64] it keeps per pair the four gate derivatives dz[p][0..3] and the 64-bit ReLU mask of
// its hidden row: dh[p][h] = mask_p[h] ? sum_i dz[p][i] W2[i][h] : 0 is re-formed on the fly.  The small
// tensors are split by hidden unit: workgroup g also owns W2[:, 4g..4g+3] and b1[4g..4g+3] (16 wave-wide
// reductions over the pairs), workgroup 0 b2.
constexpr int WD_PCH_MIN = 128;  // pairs per chunk of the W1 gradient, at least (wide_pch: all pairs in one chunk when they fit)


__global__ __launch_bounds__(1024) void pool_w1_step_wide_kernel(FusedArgs g, float* pooled_out, int32_t* topk_idx_out,
                                                                 int32_t* topk_cnt_out, int PS_CAP, int region_bytes,
                                                                 int external_pool, int WD_PCH) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const FinishArgs& a = g.f;
    const int b = a.slide0, C = a.C, K = a.K, D = a.D;
    const int wg = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int DS = D / 16, d_lo = wg * DS;               // this workgroup's columns of W1
    moc_kernarg_touch<sizeof(FusedArgs)>();
    MOC_STAMP(40);
    const int esz = a.xdt == MOC_F32 ? 4 : 2;
    const int PMAX = C * K;
    // ---- LDS carve
    unsigned char* region = smem;                                                    // [C][PS_CAP] u64  |  [P][DS] raw
    unsigned long long* list = reinterpret_cast<unsigned long long*>(region);
    unsigned long long* wmax = reinterpret_cast<unsigned long long*>(smem + region_bytes);   // [C][16]
    float* pooled_s = reinterpret_cast<float*>(wmax + (size_t)C * 16);
    float* dpool = pooled_s + C;
    int* ncand = reinterpret_cast<int*>(dpool + C);
    int* topk_s = ncand + C;                                                         // [C][K]
    // (16-byte aligned; sixteen rows of slack behind the pairs: the product below walks them in groups of sixteen)
    // (the offset is rounded, not the address: a pointer rebuilt from an integer is a FLAT pointer to the compiler)
    float* dz = reinterpret_cast<float*>(smem + (((reinterpret_cast<unsigned char*>(topk_s + PMAX) - smem) + 15) & ~(ptrdiff_t)15));   // [P + 16][4]
    float* h1o = dz + (size_t)(PMAX + 16) * 4;                                       // [P][4]
    const int PM = (PMAX + 16 + 3) & ~3;                                             // pairs per plane of the ReLU masks
    unsigned* mask = reinterpret_cast<unsigned*>(h1o + (size_t)PMAX * 4);            // [2][PM]: bits of hidden units 0..31 | 32..63
    int* sidx_s = reinterpret_cast<int*>(mask + (size_t)PM * 2);                     // [P]
    float* W2s = reinterpret_cast<float*>(sidx_s + PMAX);                            // [4][H]
    float* red = W2s + 4 * H;                                                        // [32]
    int64_t* prow_s = reinterpret_cast<int64_t*>(smem + (((reinterpret_cast<unsigned char*>(red + 32) - smem) + 7) & ~(ptrdiff_t)7));   // [P]

    // ---- operands that do not depend on this slide: requested first, consumed last.
    // The W1 gradient of this workgroup is a [64 x DS] tile product on the fp32 matrix cores: wave w < 4*DS/16
    // owns the 16 x 16 tile (hidden units 16 (w / NTN) .., columns d_lo + 16 (w % NTN) ..); lane l ends with
    // elements (h0 + (l >> 4) * 4 + i, d0 + (l & 15)), i < 4.
    const int NTN = DS / 16;                              // n-tiles (2 or 4)
    const bool has_tile = wave < 4 * NTN;
    // D = 512: eight tiles, sixteen waves -- waves 8..15 take the SECOND HALF of every chunk's pairs for the same tiles
    // (twave) and the halves meet in LDS behind the chunk loop (first + second: one fixed association); the chunk's MFMA
    // phase was 6.1 of the step's 26 us with half of the waves idle
    const bool ksplit = 4 * NTN <= 8;
    const int twave = ksplit ? (wave & 7) : wave;
    const bool in_product = ksplit || has_tile;
    const int h0 = (twave / NTN) * 16, d0 = d_lo + (twave % NTN) * 16;
    float pw[4], pm[4], pv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        pw[i] = pm[i] = pv[i] = 0.f;
        if (has_tile && a.apply_adam) {
            const int e = (h0 + (lane >> 4) * 4 + i) * D + d0 + (lane & 15);
            pw[i] = g.W1[e]; pm[i] = g.m_W1[e]; pv[i] = g.v_W1[e];
        }
    }
    const AdamCoef ak = step_coef(a);
    // (W2 through a register: stored to LDS below, once the pooling's own first loads are under way -- a store here would
    // wait for everything requested above before the slide's data is even asked for)
    const float w2r = t < 4 * H ? a.W2[t] : 0.f;
    // small tensors: threads 0..15 -> W2[i][4 wg + jj] (i = t >> 2, jj = t & 3); 16..19 -> b1[4 wg + jj]; 20..23 -> b2 (wg 0)
    float pS = 0.f, pSm = 0.f, pSv = 0.f;
    int small = -1;                                       // flat tail index (b1 | W2 | b2), as in the narrow kernel
    if (t < 16) small = H + (t >> 2) * H + wg * 4 + (t & 3);
    else if (t < 20) small = wg * 4 + (t - 16);
    else if (t < 24 && wg == 0) small = H + 4 * H + (t - 20);
    if (small >= 0 && a.apply_adam) {
        if (small >= 5 * H) { pS = a.b2[small - 5 * H]; pSm = a.m_b2[small - 5 * H]; pSv = a.v_b2[small - 5 * H]; }
        else if (small >= H) { pS = a.W2[small - H]; pSm = a.m_W2[small - H]; pSv = a.v_W2[small - H]; }
        else { pS = a.b1[small]; pSm = a.m_b1[small]; pSv = a.v_b1[small]; }
    }
    int64_t base;
    int k, P;
    unsigned kinv;
    float4 hv_pre[5];
    // the pairs' hidden rows: requested as soon as the pooled rows' indices are in LDS, their ReLU bits ORed into the masks
    // further down
    auto request_hidden = [&]() {
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int e = t + q * 1024;
            hv_pre[q] = float4{0.f, 0.f, 0.f, 0.f};
            if (e < P * 16) {
                const int p = e >> 4, c = (int)(((unsigned)p * kinv) >> 16);
                hv_pre[q] = *reinterpret_cast<const float4*>(a.H1 + (base + topk_s[c * K + (p - c * k)]) * H + (e & 15) * 4);
            }
        }
    };
    auto zero_masks = [&]() {
        for (int p = t; p < 2 * P; p += 1024) mask[(p >= P ? PM - P : 0) + p] = 0u;
        if (t < 64) dz[P * 4 + t] = 0.f;                   // the slack rows: pairs P .. P + 15 contribute nothing
    };
    if (external_pool) {
        // more than 8192 selected rows possible: the top-K came from topk_mean_kernel (one workgroup per class); pick it up,
        // add loss / argmax / d loss.  Thread e < C K owns the pooled row (class e / K, place e % K): its operands -- row id,
        // four candidate scores, gates -- are requested from the REGISTER that holds its index, before the index has even
        // been to LDS, and are under way while wave 0 does the cross entropy; the hidden rows (which need other threads'
        // indices) follow behind the first barrier.  (Before: indices -> LDS -> barrier -> cross entropy -> barrier -> the
        // operands' round trip -> barrier: 5.4 of the step's 15.7 us by the stamps.)
        base = a.base_host >= 0 ? a.base_host : a.row_off[b];
        const int y = (int)a.labels[b];
        const int PK = C * K;
        const int own_idx = a.topk_idx[(int64_t)b * PK + (t < PK ? t : PK - 1)];
        k = a.topk_cnt[(int64_t)b * C];
        const float pooled_v = a.pooled[(int64_t)b * C + (t < C ? t : 0)];
        P = C * k;
        // pairs per class k <= 16, pair numbers < 1024: p / k = (p * ceil(2^16 / k)) >> 16 exactly (no integer division)
        kinv = k > 0 ? (65536u + (unsigned)k - 1u) / (unsigned)k : 0u;
        const unsigned kinvK = (65536u + (unsigned)K - 1u) / (unsigned)K;
        const int own_c = (int)(((unsigned)t * kinvK) >> 16), own_pos = t - own_c * K;
        const bool own = t < PK && own_pos < k;
        int64_t own_row = 0;
        float sc[4] = {0.f, 0.f, 0.f, 0.f};
        float4 lam = {0.f, 0.f, 0.f, 0.f};
        if (own) {
            own_row = a.sel_row[base + own_idx];
            const float* cd = a.cand + base + own_idx;
            sc[0] = cd[(int64_t)own_c * a.stride]; sc[1] = cd[(int64_t)(C + own_c) * a.stride];
            sc[2] = cd[(int64_t)(2 * C) * a.stride]; sc[3] = cd[(int64_t)(2 * C + 1) * a.stride];
            lam = *reinterpret_cast<const float4*>(a.gates + (base + own_idx) * 4);
        }
        if (t < PK) topk_s[t] = own_idx;
        if (t < C) pooled_s[t] = pooled_v;
        if (t < 4 * H) W2s[t] = w2r;
        zero_masks();
        __syncthreads();
        request_hidden();
        ce_wave0<32>(a, b, C, y, pooled_s, dpool, wg == 0);
        __syncthreads();
        MOC_STAMP(41);
        if (own) {
            const int pr = own_c * k + own_pos;
            sidx_s[pr] = own_idx;
            prow_s[pr] = own_row;
            const float lv[4] = {lam.x, lam.y, lam.z, lam.w};
            const float gk = dpool[own_c] / (float)k;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                dz[pr * 4 + i] = (a.use_bits >> i & 1u) ? gk * sc[i] * lv[i] * (1.f - lv[i]) : 0.f;
        }
    } else {
        const PoolLds L = {list, wmax, pooled_s, dpool, ncand, topk_s};
        k = pool_phase<8, 32, 2>(a, b, L, PS_CAP, wg == 0, pooled_out, topk_idx_out, topk_cnt_out, &base);
        if (t < 4 * H) W2s[t] = w2r;
        __syncthreads();
        MOC_STAMP(41);
        // ---- pairs
        P = C * k;
        kinv = k > 0 ? (65536u + (unsigned)k - 1u) / (unsigned)k : 0u;
        zero_masks();
        request_hidden();
        for (int p = t; p < P; p += 1024) {
            const int c = (int)(((unsigned)p * kinv) >> 16), sidx = topk_s[c * K + (p - c * k)];
            sidx_s[p] = sidx;
            prow_s[p] = a.sel_row[base + sidx];
            const float* cd = a.cand + base + sidx;
            const float sc[4] = {cd[(int64_t)c * a.stride], cd[(int64_t)(C + c) * a.stride],
                                 cd[(int64_t)(2 * C) * a.stride], cd[(int64_t)(2 * C + 1) * a.stride]};
            const float4 lam = *reinterpret_cast<const float4*>(a.gates + (base + sidx) * 4);
            const float lv[4] = {lam.x, lam.y, lam.z, lam.w};
            const float gk = dpool[c] / (float)k;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                dz[p * 4 + i] = (a.use_bits >> i & 1u) ? gk * sc[i] * lv[i] * (1.f - lv[i]) : 0.f;
        }
    }
    __syncthreads();
    MOC_STAMP(42);
    // the first row pieces of the W1 gradient are requested below, right behind the hidden rows' ReLU bits (whose registers
    // they take over: both at once spill) and in front of the barrier and the small gradients: they land in LDS in the chunk loop
    const int DSb = DS * esz;                              // bytes of a pair's piece
    const int ppr = DSb / 16;                              // 16-B pieces per pair: 4, 8 or 16 -- shifts, no division
    const int ppr_sh = ppr == 4 ? 2 : ppr == 8 ? 3 : ppr == 16 ? 4 : ppr == 2 ? 1 : 0;
    // (one piece per call, returned by value, the four of a batch in named variables: an array filled under a branch, or
    // handed to a lambda by reference, goes to scratch memory)
    auto request_piece = [&](int c0, int n, int e) -> uint4 {
        const int ec = e < n * ppr ? e : n * ppr - 1;
        const int pp = ec >> ppr_sh, v = ec - (pp << ppr_sh);
        return *reinterpret_cast<const uint4*>(a.X + (prow_s[c0 + pp] * D + d_lo) * esz + v * 16);
    };
    // hidden rows of the pairs: ReLU mask (64 bits) and the four values this workgroup owns
    auto take_hidden = [&](int e, const float4& hv) {
        const int p = e >> 4, v = e & 15;
        const unsigned bits = (hv.x > 0.f ? 1u : 0u) | (hv.y > 0.f ? 2u : 0u) | (hv.z > 0.f ? 4u : 0u) | (hv.w > 0.f ? 8u : 0u);
        if (bits) atomicOr(&mask[(v >> 3) * PM + p], bits << ((v & 7) * 4));
        if (v == wg) *reinterpret_cast<float4*>(h1o + p * 4) = hv;
    };
#pragma unroll
    for (int q = 0; q < 5; ++q)
        if (t + q * 1024 < P * 16) take_hidden(t + q * 1024, hv_pre[q]);
    for (int e = t + 5 * 1024; e < P * 16; e += 1024)           // more than 320 pairs
        take_hidden(e, *reinterpret_cast<const float4*>(a.H1 + (base + sidx_s[e >> 4]) * H + (e & 15) * 4));
    const int n_first = P < WD_PCH ? P : WD_PCH;
    uint4 pre0 = {}, pre1 = {}, pre2 = {}, pre3 = {};
    if (P > 0) {
        pre0 = request_piece(0, n_first, t); pre1 = request_piece(0, n_first, t + 1024);
        pre2 = request_piece(0, n_first, t + 2048); pre3 = request_piece(0, n_first, t + 3072);
    }
    __syncthreads();                                       // masks complete
    MOC_STAMP(43);
    // ---- small gradients: 16 + 4 (+ 4) sums over the pairs, one or two per wave, pairs strided over the lanes.  They run
    // inside the W1 gradient's first gather round trip (called below between the requests for the first row pieces and
    // their stores) and need no barrier of their own (dz, h1o and W2s are complete; `red` is read behind the product's
    // barriers) -- instead of two barriers and 1.4 us behind the product.
    auto small_gradients = [&]() {
        float part = 0.f;
        {   // wave w: W2[i][4 wg + jj] with (i, jj) = (w >> 2, w & 3)
            const int i = wave >> 2, jj = wave & 3;
            for (int p = lane; p < P; p += 64) part = fmaf(dz[p * 4 + i], h1o[p * 4 + jj], part);
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
            if (lane == 0) red[wave] = part;
        }
        part = 0.f;
        if (wave >= 8 && wave < 12) {                      // b1[4 wg + (wave - 8)] = sum_p dh[p][h]
            const int hj = wave - 8, hq = wg * 4 + hj;
            for (int p = lane; p < P; p += 64) {
                const float4 z = *reinterpret_cast<const float4*>(dz + p * 4);
                const float dh = fmaf(z.w, W2s[3 * H + hq], fmaf(z.z, W2s[2 * H + hq], fmaf(z.y, W2s[H + hq], z.x * W2s[hq])));
                part += h1o[p * 4 + hj] > 0.f ? dh : 0.f;
            }
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
            if (lane == 0) red[16 + hj] = part;
        } else if (wave >= 12) {                           // b2[i] = sum_p dz[p][i]
            const int i = wave - 12;
            for (int p = lane; p < P; p += 64) part += dz[p * 4 + i];
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
            if (lane == 0) red[20 + i] = part;
        }
    };
    // ---- W1 gradient: dW1[h][d] = sum_p dh[p][h] x[p][d], pairs in chunks of WD_PCH through `region`:
    // [PCH][DS] pieces of the pairs' rows as stored, then v_mfma_f32_16x16x4_f32 over p.  dh is never staged: per group of
    // sixteen pairs a tile wave forms dh[16 pairs][its 16 hidden units] = dz[16][4] W2[4][16] with ONE matrix instruction
    // (k = the four gates), whose result registers are, lane for lane, the A operands of the group's four gradient
    // instructions -- lane (li, kq) ends with dh[pair 4 kq + i][h0 + li] in register i, which is A[m = li][k = kq] of the
    // instruction that takes the pairs {i, 4 + i, 8 + i, 12 + i} of the group as its k.  The ReLU mask is one AND per
    // element (a sign-extended bit of the pair's mask word).  What this replaced: every workgroup writing all P x 64
    // values of dh to LDS first (19 rounds x ~25 instructions x 16 waves on one CU: 7.0 of the step's 25 us by the stamps,
    // instruction issue, not latency) and reading them back as A fragments.  (Forming dh in the loop with the VALU -- the
    // old fmaf chain per element -- was measured first: 18 us, sixteen waves x 18 vector instructions per MFMA.)
    f32x4_t gacc = {0.f, 0.f, 0.f, 0.f};
    if (P == 0) small_gradients();                         // (no pair at all: the sums are zeros, but they are written)
    {
        unsigned char* xraw = region;                                              // [PCH][DS * esz]
        for (int c0 = 0; c0 < P; c0 += WD_PCH) {
            const int n = P - c0 < WD_PCH ? P - c0 : WD_PCH;
            const int n16 = (n + 15) & ~15;                // (<= WD_PCH: a multiple of 32)
            if (c0 > 0) __syncthreads();                   // previous chunk consumed
            // (up to four pieces per thread requested before the first is stored: one round trip for a whole chunk of
            // up to 4096 pieces; the first batch of the first chunk was requested above and the small gradients run in front
            // of its stores.  Requesting EVERY chunk's pieces a chunk ahead needs registers this 1024-thread kernel does not
            // have: at its 128-register cap hipcc waits for them at once and parks them in scratch -- measured 3 % slower)
            unsigned char* xr = region;
            auto store_batch = [&](int e0, uint4 q0, uint4 q1, uint4 q2, uint4 q3) {
                const int e = e0 + t;
                if (e < n * ppr) *reinterpret_cast<uint4*>(xr + (size_t)e * 16) = q0;
                if (e + 1024 < n * ppr) *reinterpret_cast<uint4*>(xr + (size_t)(e + 1024) * 16) = q1;
                if (e + 2048 < n * ppr) *reinterpret_cast<uint4*>(xr + (size_t)(e + 2048) * 16) = q2;
                if (e + 3072 < n * ppr) *reinterpret_cast<uint4*>(xr + (size_t)(e + 3072) * 16) = q3;
            };
            int e0 = 0;
            if (c0 == 0) {                                 // the batch requested above
                small_gradients();
                store_batch(0, pre0, pre1, pre2, pre3);
                e0 = 4096;
            }
            for (; e0 < n * ppr; e0 += 4096) {
                const uint4 q0 = request_piece(c0, n, e0 + t), q1 = request_piece(c0, n, e0 + t + 1024);
                const uint4 q2 = request_piece(c0, n, e0 + t + 2048), q3 = request_piece(c0, n, e0 + t + 3072);
                store_batch(e0, q0, q1, q2, q3);
            }
            // the rows behind the last pair of a group of sixteen: zero (their dh is zero, but 0 x whatever bytes lie here is not)
            for (int e = n * ppr + t; e < n16 * ppr; e += 1024) *reinterpret_cast<uint4*>(xraw + (size_t)e * 16) = uint4{0u, 0u, 0u, 0u};
            if (c0 == 0) MOC_STAMP(50);
            __syncthreads();
            if (c0 == 0) MOC_STAMP(52);
            if (in_product) {
                const int kq = lane >> 4, li = lane & 15, cc = (d0 - d_lo) + li;
                const int hh = h0 + li;                                 // this lane's hidden unit
                const float w2b = W2s[kq * H + hh];                     // B of the dh product: [gate kq][hidden unit]
                const unsigned msh = (unsigned)(hh & 31);
                // this wave's share of the chunk's groups: all of them, or (ksplit) the first / second half
                const int half = (((n16 >> 4) + 1) >> 1) << 4;
                const int kb = (ksplit && wave >= 8) ? half : 0, ke = (ksplit && wave < 8) ? half : n16;
                const float* dzp = dz + (size_t)(c0 + li) * 4 + kq;                           // A of the dh product: [pair li][gate kq]
                const uint4* mkp = reinterpret_cast<const uint4*>(mask + (size_t)(hh >> 5) * PM + c0 + 4 * kq);   // words of the lane's four pairs
                auto product = [&](auto kind) {
                    constexpr int XK = decltype(kind)::value;           // 0 fp32, 1 bf16, 2 fp16
                    typedef typename std::conditional<XK == 0, float, uint16_t>::type xs_t;
                    const xs_t* xp = reinterpret_cast<const xs_t*>(xraw) + (size_t)(4 * kq) * DS + cc;   // row ks + 4 kq + i: + (ks + i) * DS
                    auto xcv = [&](xs_t v) -> float {
                        if constexpr (XK == 0) return v;
                        else if constexpr (XK == 2) return moc_f16_to_f32(v);
                        else return moc_bf16_to_f32(v);
                    };
                    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
                    for (int ks = kb; ks < ke; ks += 16) {
                        const float za = dzp[ks * 4];
                        const uint4 m = mkp[ks >> 2];
                        const xs_t x0 = xp[(size_t)(ks + 0) * DS], x1 = xp[(size_t)(ks + 1) * DS], x2 = xp[(size_t)(ks + 2) * DS],
                                   x3 = xp[(size_t)(ks + 3) * DS];
                        const f32x4_t dh = __builtin_amdgcn_mfma_f32_16x16x4f32(za, w2b, zero4, 0, 0, 0);
                        const unsigned mk[4] = {m.x, m.y, m.z, m.w};
                        const float xv[4] = {xcv(x0), xcv(x1), xcv(x2), xcv(x3)};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const unsigned on = (unsigned)__builtin_amdgcn_sbfe((int)mk[i], msh, 1u);      // 0 or all ones
                            gacc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(__float_as_uint(dh[i]) & on), xv[i], gacc, 0, 0, 0);
                        }
                    }
                };
                if (a.xdt == MOC_F32) product(std::integral_constant<int, 0>{});
                else if (a.xdt == MOC_F16) product(std::integral_constant<int, 2>{});
                else product(std::integral_constant<int, 1>{});
            }
            if (c0 == 0) MOC_STAMP(53);
        }
    }
    __syncthreads();
    if (ksplit) {                                          // second halves -> LDS (the chunk buffers are free), first halves add them
        float* part = reinterpret_cast<float*>(region);    // [8 tiles][4][64]
        if (wave >= 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) part[(twave * 4 + i) * 64 + lane] = gacc[i];
        }
        __syncthreads();
        if (wave < 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) gacc[i] = moc_fadd(gacc[i], part[(twave * 4 + i) * 64 + lane]);
        }
    }
    MOC_STAMP(44);
    MOC_STAMP(45);
    // ---- outputs
    const float gs = ak.grad_scale;
    float gv = 0.f;
    if (small >= 0) gv = t < 16 ? red[(t >> 2) * 4 + (t & 3)] : t < 20 ? red[16 + (t - 16)] : red[20 + (t - 20)];
    if (!a.apply_adam) {                                   // gradients out (data-parallel step with a collective)
        if (has_tile)
            for (int i = 0; i < 4; ++i) g.g_W1[(h0 + (lane >> 4) * 4 + i) * D + d0 + (lane & 15)] = gacc[i];
        if (small >= 5 * H) a.g_b2[small - 5 * H] = gv;
        else if (small >= H) a.g_W2[small - H] = gv;
        else if (small >= 0) a.g_b1[small] = gv;
        return;
    }
    if (g.x.world > 1) {                                   // one-shot exchange over xGMI (moc_p2p.h), as in the narrow kernel
        __shared__ int p2p_ok;
        const int64_t nW1 = (int64_t)H * D;
        if (has_tile)
#pragma unroll
            for (int i = 0; i < 4; ++i) p2p_push(g.x, (int64_t)(h0 + (lane >> 4) * 4 + i) * D + d0 + (lane & 15), gacc[i]);
        if (small >= 0) p2p_push(g.x, nW1 + small, gv);
        if (!p2p_signal_wait(g.x, wg, &p2p_ok)) return;   // time-out: reported through g.x.error, no update
        if (has_tile)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                gacc[i] = p2p_sum(g.x, (int64_t)(h0 + (lane >> 4) * 4 + i) * D + d0 + (lane & 15), gacc[i]);
        if (small >= 0) gv = p2p_sum(g.x, nW1 + small, gv);
    }
    if (has_tile) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int h = h0 + (lane >> 4) * 4 + i, d = d0 + (lane & 15), e = h * D + d;
            adam_update(pw[i], pm[i], pv[i], gacc[i] * gs, ak);
            g.W1[e] = pw[i]; g.m_W1[e] = pm[i]; g.v_W1[e] = pv[i];
            w1_image_store(g.img_dt, g.W1img, D, h, d, pw[i]);
        }
    }
    if (small >= 0) {
        adam_update(pS, pSm, pSv, gv * gs, ak);
        if (small >= 5 * H) { const int i = small - 5 * H; a.b2[i] = pS; a.m_b2[i] = pSm; a.v_b2[i] = pSv; }
        else if (small >= H) { const int i = small - H; g.W2out[i] = pS; a.m_W2[i] = pSm; a.v_W2[i] = pSv; }
        else { a.b1[small] = pS; a.m_b1[small] = pSm; a.v_b1[small] = pSv; }
    }
    MOC_STAMP(46);
}

// ------------------------------------------------------------------ W1 gradient (+ Adam)
struct W1Args {
    const unsigned char* X;
    const float* pair_dh;
    const int64_t* pair_row;
    const int32_t* n_pair;
    float *W1, *m_W1, *v_W1, *g_W1;
    unsigned char* W1img;   // kept in sync with W1 when the step is applied (nullable)
    int D, apply_adam;
    int P_static;           // >= 0: pair count known to the host (padded list), else read n_pair
    AdamCoef adam;
};

// dW1[h][d] = sum_p dh[p][h] * x_p[d] over the <= K*C gradient pairs, then Adam in place.
// grid (D/256, H/8): a workgroup owns 256 columns d and 8 hidden units.  Everything it needs is
// requested at once: its 8 x 256 parameters and moments, and the pairs' row segments (P x 256
// elements, via LDS) -- with `P_static` >= 0 (the fused step pads its pair list to C*K entries with
// zero dh) not even n_pair has to arrive first.
constexpr int W1_MAXP = 64;
template <bool BF16, bool F16 = false>
__global__ __launch_bounds__(256) void w1_update_kernel(W1Args a) {
    __shared__ __attribute__((aligned(16))) float xs[W1_MAXP][256];
    __shared__ float dh_s[W1_MAXP][8];
    __shared__ int64_t prow_s[W1_MAXP];
    const int d = blockIdx.x * 256 + threadIdx.x, h0 = blockIdx.y * 8;
    MOC_STAMP(20);
    float pw[8], pm[8], pv[8];
    if (a.apply_adam) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = (h0 + j) * a.D + d;
            pw[j] = a.W1[e]; pm[j] = a.m_W1[e]; pv[j] = a.v_W1[e];
        }
    }
    const int P = a.P_static >= 0 ? a.P_static : *a.n_pair;
    float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int p0 = 0; p0 < P; p0 += W1_MAXP) {
        const int np = P - p0 < W1_MAXP ? P - p0 : W1_MAXP;
        if (p0 > 0) __syncthreads();
        {
            // 16-B loads, all independent: LPR lanes cover one row's 256 columns, 256/LPR rows per round
            if (threadIdx.x < np) prow_s[threadIdx.x] = a.pair_row[p0 + threadIdx.x];
            __syncthreads();
            constexpr int EPL = BF16 ? 8 : 4, LPR = 256 / EPL, RPR = 256 / LPR;   // elements/lane, lanes/row, rows/round
            const int v = threadIdx.x % LPR, pr = threadIdx.x / LPR;
            const int64_t col = (int64_t)blockIdx.x * 256 + v * EPL;
#pragma unroll
            for (int r = 0; r < W1_MAXP / RPR; ++r) {
                const int p = r * RPR + pr;
                if (p < np) {
                    const int64_t row = prow_s[p];
                    float* dst = &xs[p][v * EPL];
                    if constexpr (BF16) {
                        const uint4 raw = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(a.X) + row * a.D + col);
                        float4 lo, hi;
                        lo.x = moc_half_to_f32<F16>((uint16_t)raw.x); lo.y = moc_half_to_f32<F16>((uint16_t)(raw.x >> 16));
                        lo.z = moc_half_to_f32<F16>((uint16_t)raw.y); lo.w = moc_half_to_f32<F16>((uint16_t)(raw.y >> 16));
                        hi.x = moc_half_to_f32<F16>((uint16_t)raw.z); hi.y = moc_half_to_f32<F16>((uint16_t)(raw.z >> 16));
                        hi.z = moc_half_to_f32<F16>((uint16_t)raw.w); hi.w = moc_half_to_f32<F16>((uint16_t)(raw.w >> 16));
                        reinterpret_cast<float4*>(dst)[0] = lo;
                        reinterpret_cast<float4*>(dst)[1] = hi;
                    } else {
                        *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.X) + row * a.D + col);
                    }
                }
            }
        }
        for (int e = threadIdx.x; e < np * 8; e += 256) dh_s[e >> 3][e & 7] = a.pair_dh[(p0 + (e >> 3)) * H + h0 + (e & 7)];
        __syncthreads();
        MOC_STAMP(21);
        for (int p = 0; p < np; ++p) {
            const float xv = xs[p][threadIdx.x];
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = fmaf(dh_s[p][j], xv, g[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int e = (h0 + j) * a.D + d;
        if (a.apply_adam) {
            adam_update(pw[j], pm[j], pv[j], g[j] * a.adam.grad_scale, a.adam);
            a.W1[e] = pw[j]; a.m_W1[e] = pm[j]; a.v_W1[e] = pv[j];
            if (a.W1img) {
                if constexpr (F16) w1_image_store_half<true>(a.W1img, a.D, h0 + j, d, pw[j]);
                else if constexpr (BF16) w1_image_store_half<false>(a.W1img, a.D, h0 + j, d, pw[j]);
                else w1_image_store_f32(a.W1img, a.D, h0 + j, d, pw[j]);
            }
        } else a.g_W1[e] = g[j];
    }
    MOC_STAMP(22);
}

// gradients already in g_* (e.g. after an all-reduce): plain Adam over all four tensors
// (+ the W1 image when `img` is given, so that the next forward needs no rebuild)
__global__ __launch_bounds__(256) void adam_all_kernel(moc_meta_t M, int D, AdamCoef k, unsigned char* img, int img_dt) {
    const int nW1 = H * D, n = nW1 + H + 4 * H + 4;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    float *p, *m, *v, *g;
    int o = e;
    if (o < nW1) { p = M.W1; m = M.m_W1; v = M.v_W1; g = M.g_W1; }
    else if ((o -= nW1) < H) { p = M.b1; m = M.m_b1; v = M.v_b1; g = M.g_b1; }
    else if ((o -= H) < 4 * H) { p = M.W2; m = M.m_W2; v = M.v_W2; g = M.g_W2; }
    else { o -= 4 * H; p = M.b2; m = M.m_b2; v = M.v_b2; g = M.g_b2; }
    adam_update(p[o], m[o], v[o], g[o] * k.grad_scale, k);
    if (img && e < nW1) {
        w1_image_store(img_dt, img, D, e / D, e % D, p[o]);
    }
}

// ------------------------------------------------------------------ senet backward for a caller-written loop
// `lambdas = model(selected_feat)` kept by a caller who wrote the reference's loop body himself (main_moc.py:390-410):
// autograd hands back d loss / d lambdas [S, 4] -- zero except on the <= K*C rows that reached a class's top-K.  One
// workgroup lists the rows that carry gradient (ascending), forms dz = g * lam * (1 - lam), dh = (dz W2) * [H1 > 0] and
// the gradients of b1 / W2 / b2; the W1 gradient is w1_update_kernel over that list, as in the three-launch step.
// A dense d lambdas (some other loss) is the same code with a list of S rows.
__global__ __launch_bounds__(1024) void senet_bwd_rows_kernel(const float* H1, const float* gates, const float* gg, const float* W2,
                                                              int64_t S, float* pair_dz, float* pair_dh, int64_t* pair_row,
                                                              int32_t* n_pair, float* g_b1, float* g_W2, float* g_b2) {
    __shared__ int wcnt[16];
    __shared__ int total_s;
    __shared__ float W2s[4 * H];
    __shared__ float red[16][5 * H];                         // per p-group partials: W2 gradient [4][H] | b1 gradient [H]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t < 4 * H) W2s[t] = W2[t];
    if (t == 0) total_s = 0;
    __syncthreads();
    for (int64_t r0 = 0; r0 < S; r0 += 1024) {
        const int64_t r = r0 + t;
        float4 g = {0.f, 0.f, 0.f, 0.f}, lam = {0.f, 0.f, 0.f, 0.f};
        if (r < S) {
            g = reinterpret_cast<const float4*>(gg)[r];
            lam = reinterpret_cast<const float4*>(gates)[r];
        }
        const bool act = g.x != 0.f || g.y != 0.f || g.z != 0.f || g.w != 0.f;
        const unsigned long long bal = __ballot(act);
        if (lane == 0) wcnt[wave] = __popcll(bal);
        __syncthreads();
        int before = total_s;
        for (int w = 0; w < wave; ++w) before += wcnt[w];
        if (act) {
            const int pos = before + __popcll(bal & ((1ull << lane) - 1ull));
            pair_row[pos] = r;
            float4 dz;
            dz.x = g.x * lam.x * (1.f - lam.x); dz.y = g.y * lam.y * (1.f - lam.y);
            dz.z = g.z * lam.z * (1.f - lam.z); dz.w = g.w * lam.w * (1.f - lam.w);
            reinterpret_cast<float4*>(pair_dz)[pos] = dz;
        }
        __syncthreads();
        if (t == 0) { int a = total_s; for (int w = 0; w < 16; ++w) a += wcnt[w]; total_s = a; }
        __syncthreads();
    }
    const int P = total_s;
    if (t == 0) *n_pair = P;
    __threadfence_block();
    __syncthreads();
    // thread (p-group t >> 6, hidden unit t & 63): pairs p = group, group + 16, ...
    const int h = t & 63, grp = t >> 6;
    float gb1 = 0.f, gw[4] = {0.f, 0.f, 0.f, 0.f};
    for (int p = grp; p < P; p += 16) {
        const float4 dz = reinterpret_cast<const float4*>(pair_dz)[p];
        const float hv = H1[pair_row[p] * H + h];
        float v = 0.f;
        v = fmaf(dz.x, W2s[0 * H + h], v); v = fmaf(dz.y, W2s[1 * H + h], v);
        v = fmaf(dz.z, W2s[2 * H + h], v); v = fmaf(dz.w, W2s[3 * H + h], v);
        v = hv > 0.f ? v : 0.f;
        pair_dh[(int64_t)p * H + h] = v;
        gb1 += v;
        gw[0] = fmaf(dz.x, hv, gw[0]); gw[1] = fmaf(dz.y, hv, gw[1]);
        gw[2] = fmaf(dz.z, hv, gw[2]); gw[3] = fmaf(dz.w, hv, gw[3]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) red[grp][i * H + h] = gw[i];
    red[grp][4 * H + h] = gb1;
    __syncthreads();
    if (t < 5 * H) {                                          // the sixteen partial sums, in a fixed order
        float a = 0.f;
        for (int g16 = 0; g16 < 16; ++g16) a += red[g16][t];
        if (t < 4 * H) g_W2[t] = a; else g_b1[t - 4 * H] = a;
    } else if (t < 5 * H + 4) {
        const int i = t - 5 * H;
        float a = 0.f;
        for (int p = 0; p < P; ++p) a += pair_dz[(int64_t)p * 4 + i];
        g_b2[i] = a;
    }
}


AdamCoef adam_coef(const moc_meta_t* M, int64_t step, float grad_scale) {
    AdamCoef k;
    k.wd = (float)M->weight_decay;
    k.one_minus_b1 = (float)(1.0 - M->beta1);
    k.beta2 = (float)M->beta2;
    k.one_minus_b2 = (float)(1.0 - M->beta2);
    k.eps = (float)M->eps;
    const double bc1 = 1.0 - pow(M->beta1, (double)step);
    const double bc2 = 1.0 - pow(M->beta2, (double)step);
    k.neg_step_size = (float)(-(M->lr / bc1));
    k.bc2_sqrt = (float)sqrt(bc2);
    k.grad_scale = grad_scale;
    return k;
}

int check_meta(const moc_batch_t* B, const moc_meta_t* M, const moc_meta_ws_t* ws, const char* who,
               bool need_adam, bool need_grad, bool need_h1 = true) {
    MOC_REQUIRE(M && ws, "%s: null meta/ws", who);
    MOC_REQUIRE(M->H == H, "%s: hidden width %d unsupported (must be %d)", who, M->H, H);
    MOC_REQUIRE(M->D == B->D, "%s: meta D=%d != batch D=%d", who, M->D, B->D);
    MOC_REQUIRE(M->W1 && M->b1 && M->W2 && M->b2, "%s: null parameter", who);
    MOC_REQUIRE((!need_h1 || (ws->H1 && ws->gates)) && ws->mixed && ws->pooled && ws->topk_idx && ws->topk_cnt && ws->loss && ws->pred,
                "%s: null work array", who);
    MOC_REQUIRE(B->sel_row && B->n_sel && B->cand, "%s: batch has no phase-A outputs", who);
    MOC_REQUIRE(!(B->flags & MOC_CAND_FROM_STATS) || !(need_adam || need_grad),
                "%s: a MOC_CAND_FROM_STATS batch has no materialised candidate scores (evaluation only)", who);
    MOC_REQUIRE(B->topk <= 256 && B->C <= 256, "%s: topk/C too large for the fused step (<= 256)", who);
    if (need_adam)
        MOC_REQUIRE(M->m_W1 && M->m_b1 && M->m_W2 && M->m_b2 && M->v_W1 && M->v_b1 && M->v_W2 && M->v_b2,
                    "%s: null Adam state", who);
    if (need_grad) MOC_REQUIRE(M->g_W1 && M->g_b1 && M->g_W2 && M->g_b2, "%s: null gradient output", who);
    if (need_adam || need_grad) MOC_REQUIRE(ws->pair_dh && ws->pair_row && ws->n_pair, "%s: null backward scratch", who);
    return MOC_OK;
}

int launch_w1_image(const moc_batch_t* B, const moc_meta_t* M, hipStream_t s) {
    MOC_REQUIRE(M->W1_image, "meta: W1_image buffer is null (moc_w1_image_bytes)");
    const int grid = H * B->D / 256;
    w1_image_kernel<<<grid, 256, 0, s>>>(M->W1, B->D, (unsigned char*)M->W1_image, B->dtype);
    MOC_CHECK_LAUNCH("moc_w1_image");
    return MOC_OK;
}

constexpr int F64_SINGLE_ROWS = 16384;   // one slide: the 64-row forward from this many selectable rows on (16-bit bags)
int s_bound(const moc_batch_t* B) {
    const int64_t by_sel = (int64_t)B->topj * (2 * B->C + 2);
    return (int)(by_sel < B->max_rows ? by_sel : B->max_rows);
}

bool tiles_ok(const moc_batch_t* B, const moc_meta_ws_t* ws);

// emit_tiles: the one-launch step over tile records follows (training step of one slide): leave the records.
// runs != nullptr: the training forward of runs->n_runs meta-learners in one launch (moc_train_steps_runs), `w2_cur` /
// `w2_stride` = where their current W2 lives.
int launch_forward(const moc_batch_t* B, const moc_meta_t* M, const moc_meta_ws_t* ws, int slide0, int n,
                   uint32_t use_bits, hipStream_t s, bool emit_tiles = false, const moc_runs_t* runs = nullptr,
                   int64_t w2_stride = 0) {
    FwdArgs a;
    a.tile_on = 0; a.tile_cap = 0; a.tile_slot0 = 0;
    a.tile = TileWs();
    a.n_runs = 0; a.slide_stride = 0; a.par_stride = a.w2_stride = a.img_stride = 0;
    a.S_host = -1;
    a.X = (const unsigned char*)B->X; a.row_off = B->row_off; a.sel_row = B->sel_row; a.n_sel = B->n_sel;
    a.cand = B->cand; a.W1 = M->W1; a.b1 = M->b1; a.W2 = M->W2; a.b2 = M->b2;
    a.W1img = (const unsigned char*)M->W1_image;
    a.H1 = ws->H1; a.gates = ws->gates; a.mixed = ws->mixed; a.stride = B->total_rows;
    a.D = B->D; a.C = B->C; a.slide0 = slide0; a.use_bits = use_bits;
#ifdef MOC_FWD_DIAG
    if (const char* dg = getenv("MOC_FWD_DIAG")) a.use_bits |= (uint32_t)atoi(dg) << 8;
#endif
    a.base_host = (n == 1 && B->row_off_host) ? B->row_off_host[slide0] : -1;
    a.cand_mode = 0; a.stats = nullptr; a.sel_idx = nullptr;
    if (emit_tiles && n == 1 && a.base_host >= 0 && tiles_ok(B, ws)) {
        a.tile_on = 1;
        a.tile = tile_carve(ws->tile_ws, tile_slots(B->total_rows, B->n_slides, B->C));
        a.tile_cap = moc_cdiv(B->row_off_host[slide0 + 1] - a.base_host, 16);
        a.tile_slot0 = ((a.base_host >> 4) + slide0) * (int64_t)B->C * TILE_R;
    }
    if (B->flags & MOC_CAND_FROM_STATS) {
        MOC_REQUIRE(B->stats && B->sel_idx, "moc_meta_forward: MOC_CAND_FROM_STATS needs the batch's stats and sel_idx");
        a.cand_mode = (B->flags & MOC_STATS_COMPACT) ? 2 : 1;
        a.stats = B->stats; a.sel_idx = B->sel_idx;
    }
    dim3 grid(moc_cdiv(s_bound(B), 16), n);
    // one slide (a training step) whose S the host knows: S as an argument, and exactly the workgroups that have rows
    const bool s_known = n == 1 && B->n_sel_host != nullptr;
    if (s_known && !runs) {
        a.S_host = B->n_sel_host[slide0];
        MOC_REQUIRE(a.S_host >= 0 && a.S_host <= s_bound(B), "moc_meta_forward: n_sel_host[%d] = %d is not a row count of this batch (bound %d)",
                    slide0, a.S_host, s_bound(B));
        grid.x = a.S_host > 0 ? moc_cdiv(a.S_host, 16) : 1;
    }
    if (runs) {                                            // one training slide per run, grid.y = run
        MOC_REQUIRE(n == 1 && a.tile_on, "moc_train_steps_runs: the batched forward needs the tile-record step");
        a.n_runs = runs->n_runs; a.slide_stride = runs->slide_stride;
        a.par_stride = runs->par_stride; a.img_stride = runs->image_stride; a.w2_stride = w2_stride;
        int s_max = 0;
        for (int r = 0; r < runs->n_runs; ++r) {
            const int sl = slide0 + r * runs->slide_stride;
            a.base_r[r] = B->row_off_host[sl];
            a.tile_cap_r[r] = moc_cdiv(B->row_off_host[sl + 1] - B->row_off_host[sl], 16);
            a.tile_slot0_r[r] = ((B->row_off_host[sl] >> 4) + sl) * (int64_t)B->C * TILE_R;
            if (s_known) {
                a.S_r[r] = B->n_sel_host[sl];
                MOC_REQUIRE(a.S_r[r] >= 0 && a.S_r[r] <= s_bound(B), "moc_train_steps_runs: n_sel_host[%d] = %d is not a row count of this batch", sl, a.S_r[r]);
                s_max = a.S_r[r] > s_max ? a.S_r[r] : s_max;
            }
        }
        if (s_known) { a.S_host = 0; grid.x = s_max > 0 ? moc_cdiv(s_max, 16) : 1; }
        grid.y = runs->n_runs;
        // fp32 bags: the sixteen-wave kernel is built for the latency of ONE tile per CU; with many runs there are more
        // tiles than the chip holds sixteen-wave workgroups (two per CU), and the four-wave kernel -- the same bits
        // (tests: MOC_FORWARD_FOUR_WAVES) -- packs seven to a CU and keeps the matrix cores fed
        static const int fwd4_env = getenv("MOC_RUNS_FWD4") ? atoi(getenv("MOC_RUNS_FWD4")) : -1;
        const bool four = fwd4_env >= 0 ? fwd4_env != 0 : (int64_t)grid.x * grid.y > 512;
        if (B->dtype == MOC_F32 && four && B->C <= 16) {
            meta_forward_kernel<false, false, true><<<grid, 256, 0, s>>>(a);
            MOC_CHECK_LAUNCH("moc_meta_forward(runs, four waves)");
            return MOC_OK;
        }
        if (B->dtype == MOC_F32) {
            MOC_REQUIRE(B->D <= 1024 && B->C <= 64, "moc_train_steps_runs: fp32 bags need D <= 1024, C <= 64");
            static bool attr_r = false;
            if (!attr_r) {
                (void)hipFuncSetAttribute((const void*)meta_forward_ksplit_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, fks_lds_bytes(768));
                (void)hipFuncSetAttribute((const void*)meta_forward_ksplit_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, fks_lds_bytes(1024));
                attr_r = true;
            }
            const int lds = fks_lds_bytes(B->D);
            switch (B->D / 256) {
                case 1: meta_forward_ksplit_kernel<1, false><<<grid, 1024, lds, s>>>(a); break;
                case 2: meta_forward_ksplit_kernel<2, false><<<grid, 1024, lds, s>>>(a); break;
                case 3: meta_forward_ksplit_kernel<3, false><<<grid, 1024, lds, s>>>(a); break;
                default: meta_forward_ksplit_kernel<4, false><<<grid, 1024, lds, s>>>(a); break;
            }
        } else if (B->dtype == MOC_F16) meta_forward_kernel<true, true, true><<<grid, 256, 0, s>>>(a);
        else meta_forward_kernel<true, false, true><<<grid, 256, 0, s>>>(a);
        MOC_CHECK_LAUNCH("moc_meta_forward(runs)");
        return MOC_OK;
    }
    static const int fwd_variant = getenv("MOC_FORWARD_EVAL") ? atoi(getenv("MOC_FORWARD_EVAL")) : 128;   // diagnostic: 64 = the 64-row kernel
    if (n >= 4 && (B->D * moc_elem_size(B->dtype)) % 512 == 0 && s_bound(B) >= 1024 && fwd_variant == 128 &&
        !(B->flags & MOC_FORWARD_ROWS64)) {
        // many slides of many selected rows (evaluation): 128 rows per workgroup, rows by LDS-DMA, two workgroups per CU
        static bool attr128 = false;
        if (!attr128) {
            (void)hipFuncSetAttribute((const void*)meta_forward128_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, f128_lds(f128_kc(0)));
            (void)hipFuncSetAttribute((const void*)meta_forward128_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, f128_lds(f128_kc(1)));
            (void)hipFuncSetAttribute((const void*)meta_forward128_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, f128_lds(f128_kc(2)));
            attr128 = true;
        }
        dim3 g128(moc_cdiv(s_bound(B), F128_ROWS), n);
        if (B->dtype == MOC_F16) meta_forward128_kernel<1><<<g128, 256, f128_lds(f128_kc(1)), s>>>(a);
        else if (B->dtype == MOC_BF16) meta_forward128_kernel<0><<<g128, 256, f128_lds(f128_kc(0)), s>>>(a);
        else meta_forward128_kernel<2><<<g128, 256, f128_lds(f128_kc(2)), s>>>(a);
        MOC_CHECK_LAUNCH("moc_meta_forward(128)");
        return MOC_OK;
    }
    // ... or ONE slide of F64_SINGLE_ROWS or more selectable rows (the training forward of the 64-way x 50 k shape: 25,000
    // rows were 1,580 sixteen-row workgroups, each reading the whole 390-KB W1 image: 42.7 us; 8.9 -> 10.1 k meta-steps/s.
    // EBRAINS-30's 7,000 rows are faster sixteen at a time, and the 128-row kernel loses on one slide at either size.)
    const bool one_big = n == 1 && s_bound(B) >= F64_SINGLE_ROWS && !a.tile_on && !(B->flags & MOC_FORWARD_ROWS16);
    if ((n >= 4 || one_big) && B->dtype != MOC_F32 && (B->D * 2) % 512 == 0) {       // many slides at once (evaluation)
        dim3 g64(moc_cdiv(s_bound(B), 64), n);
        if (B->dtype == MOC_F16) meta_forward64_kernel<true><<<g64, 256, 0, s>>>(a);
        else meta_forward64_kernel<false><<<g64, 256, 0, s>>>(a);
        MOC_CHECK_LAUNCH("moc_meta_forward(64)");
        return MOC_OK;
    }
    if (B->dtype == MOC_F32 && B->D <= 1024 && B->C <= 64 && !(B->flags & MOC_FORWARD_FOUR_WAVES)) {
        // fp32 bags: the columns split over four wave groups (same bits as the four-wave kernel)
        static bool attr_ks = false;
        if (!attr_ks) {
            (void)hipFuncSetAttribute((const void*)meta_forward_ksplit_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, fks_lds_bytes(768));
            (void)hipFuncSetAttribute((const void*)meta_forward_ksplit_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, fks_lds_bytes(1024));
            (void)hipFuncSetAttribute((const void*)meta_forward_ksplit_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, fks_lds_bytes(768));
            (void)hipFuncSetAttribute((const void*)meta_forward_ksplit_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, fks_lds_bytes(1024));
            attr_ks = true;
        }
        const int lds = fks_lds_bytes(B->D);
#define MOC_KS_LAUNCH(DQ)                                                                        \
    do {                                                                                         \
        if (a.cand_mode) meta_forward_ksplit_kernel<DQ, true><<<grid, 1024, lds, s>>>(a);         \
        else meta_forward_ksplit_kernel<DQ, false><<<grid, 1024, lds, s>>>(a);                    \
    } while (0)
        switch (B->D / 256) {
            case 1: MOC_KS_LAUNCH(1); break;
            case 2: MOC_KS_LAUNCH(2); break;
            case 3: MOC_KS_LAUNCH(3); break;
            default: MOC_KS_LAUNCH(4); break;
        }
#undef MOC_KS_LAUNCH
        MOC_CHECK_LAUNCH("moc_meta_forward(ksplit)");
        return MOC_OK;
    }
    if (a.tile_on) {
        if (B->dtype == MOC_F16) meta_forward_kernel<true, true, true><<<grid, 256, 0, s>>>(a);
        else if (B->dtype == MOC_BF16) meta_forward_kernel<true, false, true><<<grid, 256, 0, s>>>(a);
        else meta_forward_kernel<false, false, true><<<grid, 256, 0, s>>>(a);
    } else if (B->dtype == MOC_F16) meta_forward_kernel<true, true><<<grid, 256, 0, s>>>(a);
    else if (B->dtype == MOC_BF16) meta_forward_kernel<true><<<grid, 256, 0, s>>>(a);
    else meta_forward_kernel<false><<<grid, 256, 0, s>>>(a);
    MOC_CHECK_LAUNCH("moc_meta_forward");
    return MOC_OK;
}

int launch_pool(const moc_batch_t* B, const moc_meta_ws_t* ws, int slide0, int n, hipStream_t s) {
    return moc_launch_topk_mean(ws->mixed, B->total_rows, ws->mixed, B->total_rows, B->row_off, B->n_sel,
                                slide0, n, B->C, B->topk, 0, ws->pooled, ws->topk_idx, ws->topk_cnt, s);
}

int launch_finish(const moc_batch_t* B, const moc_meta_t* M, const moc_meta_ws_t* ws, const int64_t* labels,
                  int slide0, int n, int train, int apply_adam, uint32_t use_bits, const AdamCoef& k, hipStream_t s) {
    FinishArgs a = {};
    a.row_off = B->row_off; a.sel_row = B->sel_row; a.n_sel = B->n_sel; a.cand = B->cand;
    a.H1 = ws->H1; a.gates = ws->gates; a.pooled = ws->pooled; a.topk_idx = ws->topk_idx; a.topk_cnt = ws->topk_cnt;
    a.labels = labels; a.loss = ws->loss; a.pred = ws->pred;
    a.W2 = M->W2; a.b2 = M->b2; a.b1 = M->b1;
    a.m_W2 = M->m_W2; a.m_b2 = M->m_b2; a.m_b1 = M->m_b1; a.v_W2 = M->v_W2; a.v_b2 = M->v_b2; a.v_b1 = M->v_b1;
    a.g_W2 = M->g_W2; a.g_b2 = M->g_b2; a.g_b1 = M->g_b1;
    a.pair_dh = ws->pair_dh; a.pair_row = ws->pair_row; a.n_pair = ws->n_pair;
    a.stride = B->total_rows; a.C = B->C; a.K = B->topk; a.slide0 = slide0; a.train = train;
    a.apply_adam = apply_adam; a.use_bits = use_bits; a.adam = k;
    const size_t smem = train ? (size_t)B->C * B->topk * (4 * sizeof(float) + sizeof(int)) +
                                (size_t)(FIN_CHUNK + 4) * H * sizeof(float) : 0;
    MOC_REQUIRE(smem <= 159 * 1024, "finish: C*topk = %d too large", B->C * B->topk);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)finish_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        attr_set = true;
    }
    finish_kernel<<<n, 256, smem, s>>>(a);
    MOC_CHECK_LAUNCH("moc_finish");
    return MOC_OK;
}

bool fused_ok(const moc_batch_t* B, int train) {
    return B->topk <= 16 && B->C <= 16 && s_bound(B) <= 64 * 16 * PS_VPT && (!train || B->C * B->topk <= 64);
}

// pooling + loss (+ pair gradients and the small-parameter step) for slides [slide0, slide0+n)
int launch_pool_finish(const moc_batch_t* B, const moc_meta_t* M, const moc_meta_ws_t* ws, const int64_t* labels,
                       int slide0, int n, int train, int apply_adam, uint32_t use_bits, const AdamCoef& k,
                       hipStream_t s) {
    if (!fused_ok(B, train)) {
        if (int rc = launch_pool(B, ws, slide0, n, s)) return rc;
        return launch_finish(B, M, ws, labels, slide0, n, train, apply_adam, use_bits, k, s);
    }
    FinishArgs a = {};
    a.row_off = B->row_off; a.sel_row = B->sel_row; a.n_sel = B->n_sel; a.cand = B->cand;
    a.H1 = ws->H1; a.gates = ws->gates; a.pooled = ws->pooled; a.mixed_in = ws->mixed;
    a.labels = labels; a.loss = ws->loss; a.pred = ws->pred;
    a.W2 = M->W2; a.b2 = M->b2; a.b1 = M->b1;
    a.m_W2 = M->m_W2; a.m_b2 = M->m_b2; a.m_b1 = M->m_b1; a.v_W2 = M->v_W2; a.v_b2 = M->v_b2; a.v_b1 = M->v_b1;
    a.g_W2 = M->g_W2; a.g_b2 = M->g_b2; a.g_b1 = M->g_b1;
    a.pair_dh = ws->pair_dh; a.pair_row = ws->pair_row; a.n_pair = ws->n_pair;
    a.stride = B->total_rows; a.C = B->C; a.K = B->topk; a.slide0 = slide0; a.train = train;
    a.apply_adam = apply_adam; a.use_bits = use_bits; a.adam = k;
    a.X = (const unsigned char*)B->X; a.D = B->D; a.xdt = B->dtype;   /* storage code: 0 f32, 1 bf16, 2 f16 */
    a.base_host = -1; a.seg_host = 0;
    if (n == 1 && B->row_off_host) {
        a.base_host = B->row_off_host[slide0];
        a.seg_host = (int)(B->row_off_host[slide0 + 1] - a.base_host);
    }
    const size_t C = B->C, K = B->topk, PK = train ? C * K : 0;
    const int cap = C <= 8 ? PS_CAP_MAX : PS_CAP_MAX / 2;
    const size_t smem = C * cap * 8 + C * 16 * 8 + C * 4 * 3 + C * K * 4 + PK * (4 * 4 + 2 * H * 4) + 4 * H * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)pool_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    pool_step_kernel<<<n, 1024, smem, s>>>(a, ws->pooled, ws->topk_idx, ws->topk_cnt, cap);
    MOC_CHECK_LAUNCH("moc_pool_step");
    return MOC_OK;
}

size_t fused_step_smem(const moc_batch_t* B, int cap) {
    const size_t C = B->C, K = B->topk, PK = C * K;
    return C * cap * 8 + C * 16 * 8 + C * 4 * 3 + C * K * 4 + PK * (4 * 4 + 2 * H * 4) + 4 * H * 4 + PK * 8 + 16 +
           PK * (size_t)B->D * 4;
}

// the one-launch step (pool_w1_step_kernel) applies when the pairs' rows fit in LDS
bool fused_step_ok(const moc_batch_t* B, const moc_meta_ws_t* ws) {
    if (!fused_ok(B, 1) || !ws->W2_alt) return false;
    if (B->D > 1024 || (int64_t)B->C * B->topk * B->D > FS_MAX_XS) return false;
    const int cap = B->C <= 8 ? PS_CAP_MAX : PS_CAP_MAX / 2;
    return fused_step_smem(B, cap) <= FS_MAX_DYN_LDS;
}

// the one-launch step over the forward's tile records (pool_w1_step_tiles_kernel): the shapes of pool_w1_step_kernel, when
// the caller has given the record arrays
// (+ 1 KiB at the very end: the prefetch's sink)
constexpr size_t TILES_SINK = 1024;
size_t tiles_step_smem(const moc_batch_t* B, int cap) {
    const size_t C = B->C, K = B->topk, P = C * K, CL = 64;
    return TILES_SINK + P * 32 + P * 256 * 4 + (P + 4) * 4 + P * 16 + C * cap * 8 + C * 8 + P * 8 + C * CL * 4 + C * 16 * 4 + C * 4 * 4 + P * 4 + P * 4 + 16 + 64;
}
bool tiles_ok(const moc_batch_t* B, const moc_meta_ws_t* ws) {
    static const bool off = getenv("MOC_TILE_RECORDS") && atoi(getenv("MOC_TILE_RECORDS")) == 0;   // diagnostic: the round-3 step
    if (off || !ws->tile_ws || !B->row_off_host || !fused_step_ok(B, ws)) return false;
    if (moc_cdiv(s_bound(B), 16) * TILE_R > 16 * 64) return false;       // a class's records: at most sixteen keys per lane
    if (ws->tile_ws_bytes < (int64_t)tile_bytes(tile_slots(B->total_rows, B->n_slides, B->C))) return false;
    const int cap = B->C <= 8 ? PS_CAP_MAX : PS_CAP_MAX / 2;
    return tiles_step_smem(B, cap) <= (size_t)FS_MAX_DYN_LDS;
}

// the wide one-launch step (pool_w1_step_wide_kernel): C <= 64, K <= 16, S <= 8192, every pair's D/16-column
// piece in LDS
int wide_cap(const moc_batch_t* B) {
    int cap = 1024;
    while (cap > 64 && (size_t)B->C * cap * 8 > 64 * 1024) cap >>= 1;
    return cap;
}
size_t wide_region(const moc_batch_t* B, int pch) {
    const size_t lists = (size_t)B->C * wide_cap(B) * 8;
    const size_t rows = (size_t)pch * ((B->D / 16) * moc_elem_size(B->dtype));   // row pieces of one chunk (dh is formed in the product loop)
    return ((lists > rows ? lists : rows) + 15) & ~(size_t)15;
}
size_t wide_smem(const moc_batch_t* B, int pch) {
    const size_t C = B->C, PK = C * B->topk;
    return wide_region(B, pch) + C * 16 * 8 + C * 4 * 3 + PK * 4 + PK * 16 * 2 + PK * 8 + PK * 4 + (4 * H + 32) * 4 + 8 + PK * 8
           + 16 + 16 * 16 + 2 * 20 * 4;      // dz aligned to 16 bytes, its sixteen slack rows, the mask planes' padding
}
// pairs per chunk of the W1 gradient: all C * K of them in ONE chunk when its row pieces fit beside the rest
// (EBRAINS-30: 300 pairs, 38 KiB at fp32; the 64-way shape: 640 pairs, 80 KiB at fp16) -- one gather round trip and one
// barrier instead of one per chunk; else the largest multiple of 32 that fits, at least WD_PCH_MIN
int wide_pch(const moc_batch_t* B) {
    int pch = (B->C * B->topk + 31) & ~31;
    while (pch > WD_PCH_MIN && wide_smem(B, pch) > (size_t)FS_MAX_DYN_LDS) pch -= 32;
    return pch < WD_PCH_MIN ? WD_PCH_MIN : pch;
}
bool fused_wide_ok(const moc_batch_t* B, const moc_meta_ws_t* ws) {
    if (!ws->W2_alt || B->topk > 16 || B->C > 64) return false;      // (S > 8192: pooling by topk_mean_kernel first)
    if (B->D % 512 != 0 || B->D > 1024) return false;        // D/16 = 32 or 64 columns per workgroup
    return wide_smem(B, WD_PCH_MIN) <= (size_t)FS_MAX_DYN_LDS;
}
// 0: three launches, 1: pool_w1_step_kernel, 2: pool_w1_step_wide_kernel
int fused_step_mode(const moc_batch_t* B, const moc_meta_ws_t* ws) {
    return fused_step_ok(B, ws) ? 1 : fused_wide_ok(B, ws) ? 2 : 0;
}

// device-resident Adam coefficients for graph replay: the step's coefficients are tab[ctr[0] + pos]
struct StepTab { const AdamCoef* tab; const int32_t* ctr; int pos; };

void fused_step_attrs() {
    // (outside any stream capture: raise the dynamic LDS limits of the two one-launch step kernels once)
    static bool done = false;
    if (done) return;
    (void)hipFuncSetAttribute((const void*)pool_w1_step_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FS_MAX_DYN_LDS);
    (void)hipFuncSetAttribute((const void*)pool_w1_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FS_MAX_DYN_LDS);
#define MOC_TILES_ATTR(VQ)                                                                                                         \
    (void)hipFuncSetAttribute((const void*)pool_w1_step_tiles_kernel<VQ, TileStepArgs>, hipFuncAttributeMaxDynamicSharedMemorySize, FS_MAX_DYN_LDS); \
    (void)hipFuncSetAttribute((const void*)pool_w1_step_tiles_kernel<VQ, TileStepArgsRuns>, hipFuncAttributeMaxDynamicSharedMemorySize, FS_MAX_DYN_LDS)
    MOC_TILES_ATTR(4); MOC_TILES_ATTR(8); MOC_TILES_ATTR(12); MOC_TILES_ATTR(16);
#undef MOC_TILES_ATTR
    done = true;
}

// the argument block of pool_w1_step_tiles_kernel for slide `slide` of B (one run)
void tile_region(const moc_batch_t* B, int slide, int64_t* slot0, int* tcap, int* tb) {
    const int64_t base = B->row_off_host[slide], seg = B->row_off_host[slide + 1] - base;
    // the slide's region: cap tiles per class; the kernel's loads are issued before n_sel is known and clamped to the
    // tiles the slide can have at all (at most s_bound selected rows)
    *tcap = moc_cdiv(seg, 16) > 0 ? moc_cdiv(seg, 16) : 1;
    int tb_all = moc_cdiv(s_bound(B), 16);
    if (B->n_sel_host) {                                    // the slide's own tile count: fewer record keys per lane in the step
        const int sh = B->n_sel_host[slide];
        if (sh >= 0 && moc_cdiv(sh, 16) < tb_all) tb_all = sh > 0 ? moc_cdiv(sh, 16) : 1;
    }
    *tb = *tcap < tb_all ? *tcap : tb_all;
    *slot0 = ((base >> 4) + slide) * (int64_t)B->C * TILE_R;
}

TileStepArgs tile_step_args(const moc_batch_t* B, const moc_meta_t* M, const moc_meta_ws_t* ws, const int64_t* labels, int slide,
                            uint32_t use_bits, const AdamCoef& k, float* W2out, int apply_adam, const StepTab* tab) {
    const TileWs T = tile_carve(ws->tile_ws, tile_slots(B->total_rows, B->n_slides, B->C));
    TileStepArgs ta = {};
    ta.tkey = T.key; ta.trho = T.rho; ta.trid = T.rid; ta.tlam = T.lam; ta.tsc = T.sc;
    tile_region(B, slide, &ta.slot0, &ta.cap, &ta.ntile_bound);
    ta.C = B->C; ta.K = B->topk; ta.D = B->D; ta.slide0 = slide; ta.PS_CAP = B->C <= 8 ? PS_CAP_MAX : PS_CAP_MAX / 2;
    ta.xdt = B->dtype; ta.n_sel = B->n_sel; ta.labels = labels;
    ta.W1 = M->W1; ta.m_W1 = M->m_W1; ta.v_W1 = M->v_W1; ta.W2 = M->W2;
    ta.b1 = M->b1; ta.m_b1 = M->m_b1; ta.v_b1 = M->v_b1; ta.b2 = M->b2; ta.m_b2 = M->m_b2; ta.v_b2 = M->v_b2;
    ta.m_W2 = M->m_W2; ta.v_W2 = M->v_W2;
    ta.base = B->row_off_host[slide]; ta.adam = k;
    ta.row_off = B->row_off; ta.prefetch_next = 0;
    ta.S_host = B->n_sel_host ? B->n_sel_host[slide] : -1;
    ta.sink_off = (int)(tiles_step_smem(B, ta.PS_CAP) - TILES_SINK);
    if (tab) { ta.adam_tab = tab->tab; ta.adam_ctr = tab->ctr; ta.adam_pos = tab->pos; }
    ta.apply_adam = apply_adam; ta.use_bits = use_bits; ta.img_dt = B->dtype;
    ta.H1 = ws->H1; ta.X = (const unsigned char*)B->X;
    ta.pooled_out = ws->pooled; ta.topk_idx_out = ws->topk_idx; ta.topk_cnt_out = ws->topk_cnt;
    ta.loss = ws->loss; ta.pred = ws->pred; ta.n_pair = ws->n_pair;
    ta.W1img = (unsigned char*)M->W1_image; ta.W2out = W2out;
    ta.g_W1 = M->g_W1; ta.g_b1 = M->g_b1; ta.g_W2 = M->g_W2; ta.g_b2 = M->g_b2;
    ta.mixed_in = ws->mixed; ta.cand = B->cand; ta.gates = ws->gates; ta.sel_row = B->sel_row; ta.stride = B->total_rows;
    return ta;
}

// launches it -- for ONE meta-learner (the common argument block alone) or for tr.s.n_runs of them (grid.z; the per-run
// scalars behind the common block); the largest tile bound of any run picks the keys per lane
template <typename ArgsT>
int launch_tile_step_as(const moc_batch_t* B, const ArgsT& args, int runs, int tb, hipStream_t s) {
    const TileStepArgs& ta = tile_common(args);
    const size_t sm = tiles_step_smem(B, ta.PS_CAP);
    const dim3 grid(B->D / 256 + (ta.tail_inside ? 0 : 1), H, runs);   // D / 256 column blocks of W1 (+ the tail workgroup), per hidden unit
#define MOC_TILES_LAUNCH(VQ) pool_w1_step_tiles_kernel<VQ, ArgsT><<<grid, 256, sm, s>>>(args)
    const int vq = moc_cdiv((int64_t)tb * TILE_R, 64);                 // keys per lane of a class wave
    if (vq <= 4) MOC_TILES_LAUNCH(4);
    else if (vq <= 8) MOC_TILES_LAUNCH(8);
    else if (vq <= 12) MOC_TILES_LAUNCH(12);
    else MOC_TILES_LAUNCH(16);
#undef MOC_TILES_LAUNCH
    MOC_CHECK_LAUNCH("moc_fused_step(tiles)");
    return MOC_OK;
}
int launch_tile_step(const moc_batch_t* B, const TileStepArgs& ta, hipStream_t s) {
    return launch_tile_step_as(B, ta, 1, ta.ntile_bound, s);
}
int launch_tile_step_runs(const moc_batch_t* B, const TileStepArgsRuns& tr, hipStream_t s) {
    int tb = tr.s.ntile_bound;
    for (int r = 0; r < tr.s.n_runs; ++r) tb = tr.ntb_r[r] > tb ? tr.ntb_r[r] : tb;
    return launch_tile_step_as(B, tr, tr.s.n_runs, tb, s);
}

// next_slide: the slide the step after this one works on, in the same work arrays (-1: none / unknown) -- the tile-record
// step pulls its selected rows toward the Infinity Cache
int launch_fused_step(const moc_batch_t* B, const moc_meta_t* M, const moc_meta_ws_t* ws, const int64_t* labels,
                      int slide, uint32_t use_bits, const AdamCoef& k, float* W2out, hipStream_t s,
                      int apply_adam = 1, const P2pArgs* x = nullptr, const StepTab* tab = nullptr, int next_slide = -1) {
    FusedArgs g = {};
    if (x) g.x = *x;
    FinishArgs& a = g.f;
    if (tab) { a.adam_tab = tab->tab; a.adam_ctr = tab->ctr; a.adam_pos = tab->pos; }
    a.row_off = B->row_off; a.sel_row = B->sel_row; a.n_sel = B->n_sel; a.cand = B->cand;
    a.H1 = ws->H1; a.gates = ws->gates; a.pooled = ws->pooled; a.mixed_in = ws->mixed;
    a.labels = labels; a.loss = ws->loss; a.pred = ws->pred;
    a.W2 = M->W2; a.b2 = M->b2; a.b1 = M->b1;
    a.m_W2 = M->m_W2; a.m_b2 = M->m_b2; a.m_b1 = M->m_b1; a.v_W2 = M->v_W2; a.v_b2 = M->v_b2; a.v_b1 = M->v_b1;
    a.stride = B->total_rows; a.C = B->C; a.K = B->topk; a.slide0 = slide; a.train = 1;
    a.apply_adam = apply_adam; a.use_bits = use_bits; a.adam = k;
    a.g_W2 = M->g_W2; a.g_b2 = M->g_b2; a.g_b1 = M->g_b1; g.g_W1 = M->g_W1;
    a.X = (const unsigned char*)B->X; a.D = B->D; a.xdt = B->dtype;   /* storage code: 0 f32, 1 bf16, 2 f16 */
    a.base_host = -1; a.seg_host = 0;
    if (B->row_off_host) {
        a.base_host = B->row_off_host[slide];
        a.seg_host = (int)(B->row_off_host[slide + 1] - a.base_host);
    }
    g.W1 = M->W1; g.m_W1 = M->m_W1; g.v_W1 = M->v_W1; g.W1img = (unsigned char*)M->W1_image;
    g.W2out = W2out; g.img_dt = B->dtype;
    if (!fused_step_ok(B, ws)) {                           // wide shapes
        static bool wide_attr = false;
        if (!wide_attr) {
            if (hipFuncSetAttribute((const void*)pool_w1_step_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FS_MAX_DYN_LDS) != hipSuccess)
                MOC_FAIL(MOC_ELAUNCH, "moc_fused_step: cannot raise the dynamic LDS limit to %d bytes", FS_MAX_DYN_LDS);
            wide_attr = true;
        }
        // pooling inside the kernel: every one of the 16 workgroups ranks all C classes for itself -- fine for a few
        // ten thousand scores, not for EBRAINS-30's 30 x 7,500 (measured: 88 us against 11.4 + 28 with one workgroup per
        // class in topk_mean_kernel first); beyond 8,192 rows it does not fit the registers at all
        const int external = s_bound(B) > 8192 || (int64_t)B->C * s_bound(B) > 49152;
        if (external) {
            if (int rc = launch_pool(B, ws, slide, 1, s)) return rc;
            a.topk_idx = ws->topk_idx; a.topk_cnt = ws->topk_cnt;
        }
        const int pch = wide_pch(B);
        pool_w1_step_wide_kernel<<<H / 4, 1024, wide_smem(B, pch), s>>>(g, ws->pooled, ws->topk_idx, ws->topk_cnt, wide_cap(B),
                                                                         (int)wide_region(B, pch), external, pch);
        MOC_CHECK_LAUNCH("moc_fused_step(wide)");
        return MOC_OK;
    }
    const int cap = B->C <= 8 ? PS_CAP_MAX : PS_CAP_MAX / 2;
    if (g.x.world <= 1 && tiles_ok(B, ws)) {               // over the forward's tile records (the forward was told to leave them)
        fused_step_attrs();
        TileStepArgs ta = tile_step_args(B, M, ws, labels, slide, use_bits, k, W2out, apply_adam, tab);
        static const bool prefetch = !(getenv("MOC_STEP_PREFETCH") && atoi(getenv("MOC_STEP_PREFETCH")) == 0);   // diagnostic: off
        if (prefetch && next_slide == slide + 1 && next_slide < B->n_slides && B->C < 4) ta.prefetch_next = 1;
        return launch_tile_step(B, ta, s);
    }
    const size_t smem = fused_step_smem(B, cap);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)pool_w1_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FS_MAX_DYN_LDS) != hipSuccess)
            MOC_FAIL(MOC_ELAUNCH, "moc_fused_step: cannot raise the dynamic LDS limit to %d bytes", FS_MAX_DYN_LDS);
        attr_set = true;
    }
    // 32 workgroups of two hidden units each, unless the gradient is exchanged inside the kernel (16 channels)
    pool_w1_step_kernel<<<g.x.world > 1 ? H / 4 : H / 2, 1024, smem, s>>>(g, ws->pooled, ws->topk_idx, ws->topk_cnt, cap);
    MOC_CHECK_LAUNCH("moc_fused_step");
    return MOC_OK;
}

int launch_w1(const moc_batch_t* B, const moc_meta_t* M, const moc_meta_ws_t* ws, int apply_adam,
              const AdamCoef& k, hipStream_t s, bool padded_pairs) {
    W1Args a;
    a.P_static = padded_pairs ? B->C * B->topk : -1;
    // (measured: gathering the pairs' rows in the single-workgroup step kernel for this kernel costs
    // more there (+2 us) than it saves here (-1.4 us): the rows are gathered here)
    a.W1img = (unsigned char*)M->W1_image;
    a.X = (const unsigned char*)B->X; a.pair_dh = ws->pair_dh; a.pair_row = ws->pair_row; a.n_pair = ws->n_pair;
    a.W1 = M->W1; a.m_W1 = M->m_W1; a.v_W1 = M->v_W1; a.g_W1 = M->g_W1; a.D = B->D; a.apply_adam = apply_adam; a.adam = k;
    const dim3 grid(B->D / 256, H / 8);
    if (B->dtype == MOC_F16) w1_update_kernel<true, true><<<grid, 256, 0, s>>>(a);
    else if (B->dtype == MOC_BF16) w1_update_kernel<true><<<grid, 256, 0, s>>>(a);
    else w1_update_kernel<false><<<grid, 256, 0, s>>>(a);
    MOC_CHECK_LAUNCH("moc_w1_update");
    return MOC_OK;
}

}  // namespace

extern "C" int moc_meta_forward(const moc_batch_t* B, const moc_meta_t* M, const moc_meta_ws_t* ws,
                                int slide0, int n, uint32_t use_bits, moc_stream_t stream) {
    if (int rc = moc_check_batch(B, "moc_meta_forward")) return rc;
    if (int rc = check_meta(B, M, ws, "moc_meta_forward", false, false, false)) return rc;
    MOC_REQUIRE(slide0 >= 0 && n >= 1 && slide0 + n <= B->n_slides, "moc_meta_forward: bad slide range");
    // the parameters may have changed since the image was last written: rebuild it (H*D elements)
    if (int rc = launch_w1_image(B, M, (hipStream_t)stream)) return rc;
    return launch_forward(B, M, ws, slide0, n, use_bits, (hipStream_t)stream);
}

extern "C" int moc_mix_fixed(const moc_batch_t* B, const moc_meta_ws_t* ws, int slide0, int n, int mode,
                             moc_stream_t stream) {
    if (int rc = moc_check_batch(B, "moc_mix_fixed")) return rc;
    MOC_REQUIRE(ws && ws->mixed && B->n_sel && B->cand, "moc_mix_fixed: null work array");
    MOC_REQUIRE(!(B->flags & MOC_CAND_FROM_STATS), "moc_mix_fixed: a MOC_CAND_FROM_STATS batch has no materialised candidate scores");
    MOC_REQUIRE(slide0 >= 0 && n >= 1 && slide0 + n <= B->n_slides, "moc_mix_fixed: bad slide range");
    MOC_REQUIRE(mode >= 0 && mode <= 2, "moc_mix_fixed: mode %d not in {0 avg, 1 sum, 2 max}", mode);
    FwdArgs a = {};
    a.base_host = -1;
    a.row_off = B->row_off; a.n_sel = B->n_sel; a.cand = B->cand; a.mixed = ws->mixed;
    a.stride = B->total_rows; a.C = B->C; a.slide0 = slide0;
    fixed_mix_kernel<<<dim3(moc_cdiv(s_bound(B), 256), n), 256, 0, (hipStream_t)stream>>>(a, mode);
    MOC_CHECK_LAUNCH("moc_mix_fixed");
    return MOC_OK;
}

extern "C" int moc_pool_loss(const moc_batch_t* B, const moc_meta_ws_t* ws, const int64_t* labels,
                             int slide0, int n, moc_stream_t stream) {
    if (int rc = moc_check_batch(B, "moc_pool_loss")) return rc;
    MOC_REQUIRE(ws && labels, "moc_pool_loss: null ws/labels");
    MOC_REQUIRE(slide0 >= 0 && n >= 1 && slide0 + n <= B->n_slides, "moc_pool_loss: bad slide range");
    MOC_REQUIRE(B->C <= 256, "moc_pool_loss: C > 256");
    hipStream_t s = (hipStream_t)stream;
    moc_meta_t none = {};
    AdamCoef k = {};
    return launch_pool_finish(B, &none, ws, labels, slide0, n, 0, 0, 0, k, s);
}

extern "C" int moc_ce_loss(const float* pooled, const int64_t* labels, int n, int C, float* loss, int32_t* pred,
                           moc_stream_t stream) {
    MOC_REQUIRE(pooled && labels && loss && pred, "moc_ce_loss: null pointer");
    MOC_REQUIRE(n >= 1 && C >= 1 && C <= 256, "moc_ce_loss: bad n=%d C=%d (C <= 256)", n, C);
    FinishArgs a = {};
    a.pooled = pooled; a.labels = labels; a.loss = loss; a.pred = pred; a.C = C; a.slide0 = 0; a.train = 0;
    finish_kernel<<<n, 256, 0, (hipStream_t)stream>>>(a);
    MOC_CHECK_LAUNCH("moc_ce_loss");
    return MOC_OK;
}

extern "C" int moc_train_grad(const moc_batch_t* B, const moc_meta_t* M, const moc_meta_ws_t* ws,
                              const int64_t* labels, int slide, uint32_t use_bits, moc_stream_t stream) {
    if (int rc = moc_check_batch(B, "moc_train_grad")) return rc;
    if (int rc = check_meta(B, M, ws, "moc_train_grad", false, true)) return rc;
    MOC_REQUIRE(labels && slide >= 0 && slide < B->n_slides, "moc_train_grad: bad labels/slide");
    hipStream_t s = (hipStream_t)stream;
    AdamCoef k = {};
    k.grad_scale = 1.f;
    // forward (fresh W1 image: the parameters may have been stepped by moc_adam_step), pooling +
    // loss + pair gradients, W1 gradient -- everything one meta-step does short of the update
    if (int rc = launch_w1_image(B, M, s)) return rc;
    if (int rc = launch_forward(B, M, ws, slide, 1, use_bits, s, true)) return rc;
    if (fused_step_mode(B, ws)) return launch_fused_step(B, M, ws, labels, slide, use_bits, k, nullptr, s, 0);
    if (int rc = launch_pool_finish(B, M, ws, labels, slide, 1, 1, 0, use_bits, k, s)) return rc;
    return launch_w1(B, M, ws, 0, k, s, fused_ok(B, 1));
}

extern "C" int moc_senet_backward(const void* X, int dtype, int64_t S, int D, const float* H1, const float* gates,
                                  const float* grad_gates, const float* W2, float* g_W1, float* g_b1, float* g_W2, float* g_b2,
                                  float* pair_dz, float* pair_dh, int64_t* pair_row, int32_t* n_pair, moc_stream_t stream) {
    MOC_REQUIRE(X && H1 && gates && grad_gates && W2 && g_W1 && g_b1 && g_W2 && g_b2 && pair_dz && pair_dh && pair_row && n_pair,
                "moc_senet_backward: null pointer");
    MOC_REQUIRE(dtype == MOC_F32 || dtype == MOC_BF16 || dtype == MOC_F16, "moc_senet_backward: bad dtype %d", dtype);
    MOC_REQUIRE(S >= 1 && D > 0 && D % 256 == 0, "moc_senet_backward: bad shape S=%lld D=%d (D a multiple of 256)", (long long)S, D);
    hipStream_t s = (hipStream_t)stream;
    senet_bwd_rows_kernel<<<1, 1024, 0, s>>>(H1, gates, grad_gates, W2, S, pair_dz, pair_dh, pair_row, n_pair, g_b1, g_W2, g_b2);
    MOC_CHECK_LAUNCH("moc_senet_backward(rows)");
    W1Args a = {};
    a.P_static = -1;
    a.X = (const unsigned char*)X; a.pair_dh = pair_dh; a.pair_row = pair_row; a.n_pair = n_pair;
    a.g_W1 = g_W1; a.D = D; a.apply_adam = 0;
    const dim3 grid(D / 256, H / 8);
    if (dtype == MOC_F16) w1_update_kernel<true, true><<<grid, 256, 0, s>>>(a);
    else if (dtype == MOC_BF16) w1_update_kernel<true><<<grid, 256, 0, s>>>(a);
    else w1_update_kernel<false><<<grid, 256, 0, s>>>(a);
    MOC_CHECK_LAUNCH("moc_senet_backward(W1)");
    return MOC_OK;
}

extern "C" int moc_adam_step(const moc_meta_t* M, float grad_scale, moc_stream_t stream) {
    MOC_REQUIRE(M && M->H == H && M->D > 0, "moc_adam_step: bad meta");
    MOC_REQUIRE(M->W1 && M->m_W1 && M->v_W1 && M->g_W1 && M->b1 && M->m_b1 && M->v_b1 && M->g_b1 &&
                M->W2 && M->m_W2 && M->v_W2 && M->g_W2 && M->b2 && M->m_b2 && M->v_b2 && M->g_b2,
                "moc_adam_step: null tensor");
    const AdamCoef k = adam_coef(M, M->step + 1, grad_scale);
    const int n = H * M->D + H + 4 * H + 4;
    adam_all_kernel<<<moc_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(*M, M->D, k, nullptr, 0);
    MOC_CHECK_LAUNCH("moc_adam_step");
    return MOC_OK;
}

extern "C" int moc_train_steps_dp(const moc_batch_t* B, const moc_meta_t* M, const moc_meta_ws_t* ws,
                                  const int64_t* labels, int slide0, int n, uint32_t use_bits,
                                  float* grad_flat, int64_t grad_count, moc_allreduce_fn allreduce,
                                  void* comm, int world, moc_stream_t stream) {
    if (int rc = moc_check_batch(B, "moc_train_steps_dp")) return rc;
    if (int rc = check_meta(B, M, ws, "moc_train_steps_dp", true, true)) return rc;
    MOC_REQUIRE(labels && slide0 >= 0 && n >= 1 && slide0 + n <= B->n_slides, "moc_train_steps_dp: bad labels/slide range");
    const int64_t n_par = (int64_t)H * M->D + H + 4 * H + 4;
    MOC_REQUIRE(world >= 1 && (world == 1 || allreduce), "moc_train_steps_dp: world %d needs an all-reduce", world);
    MOC_REQUIRE(!allreduce || (grad_flat && grad_count >= n_par && M->g_W1 >= grad_flat &&
                               M->g_W1 + (int64_t)H * M->D <= grad_flat + grad_count),
                "moc_train_steps_dp: the gradient tensors must live in grad_flat[%lld]", (long long)grad_count);
    hipStream_t s = (hipStream_t)stream;
    if (int rc = launch_w1_image(B, M, s)) return rc;      // afterwards the Adam kernel keeps it in sync
    const bool fused = fused_step_mode(B, ws) != 0;
    const int img_dt = B->dtype;
    AdamCoef kg = {};
    kg.grad_scale = 1.f;
    for (int t = 0; t < n; ++t) {
        const int b = slide0 + t;
        if (int rc = launch_forward(B, M, ws, b, 1, use_bits, s, fused)) return rc;
        if (fused) {
            if (int rc = launch_fused_step(B, M, ws, labels, b, use_bits, kg, nullptr, s, 0)) return rc;
        } else {
            if (int rc = launch_pool_finish(B, M, ws, labels, b, 1, 1, 0, use_bits, kg, s)) return rc;
            if (int rc = launch_w1(B, M, ws, 0, kg, s, fused_ok(B, 1))) return rc;
        }
        if (allreduce) {   // the ONE collective of a step: sum of the flat gradient over the ranks
            const int rc = allreduce(grad_flat, grad_flat, (size_t)grad_count, 7 /* ncclFloat32 */, 0 /* ncclSum */, comm, stream);
            if (rc != 0) MOC_FAIL(MOC_ELAUNCH, "moc_train_steps_dp: all-reduce failed at step %d (rc=%d)", t, rc);
        }
        const AdamCoef k = adam_coef(M, M->step + 1 + t, 1.f / (float)world);
        adam_all_kernel<<<moc_cdiv(n_par, 256), 256, 0, s>>>(*M, M->D, k, (unsigned char*)M->W1_image, img_dt);
        MOC_CHECK_LAUNCH("moc_train_steps_dp");
    }
    return MOC_OK;
}

extern "C" int moc_p2p_step_supported(int C, int topk, int D, int topj) {
    // from the run's constants only (not the bag sizes): every rank must take the same path
    moc_batch_t B = {};
    B.C = C; B.topk = topk; B.D = D; B.topj = topj; B.max_rows = 0x7fffffff;
    moc_meta_ws_t ws = {};
    float dummy;
    ws.W2_alt = &dummy;
    return C >= 1 && topk >= 1 && D >= 1 && topj >= 1 && fused_step_mode(&B, &ws) != 0 ? 1 : 0;
}

extern "C" int moc_train_steps_p2p(const moc_batch_t* B, const moc_meta_t* M, const moc_meta_ws_t* ws,
                                   const int64_t* labels, int slide0, int n, uint32_t use_bits,
                                   moc_p2p_t* comm, moc_stream_t stream) {
    if (int rc = moc_check_batch(B, "moc_train_steps_p2p")) return rc;
    if (int rc = check_meta(B, M, ws, "moc_train_steps_p2p", true, false)) return rc;
    MOC_REQUIRE(labels && slide0 >= 0 && n >= 1 && slide0 + n <= B->n_slides, "moc_train_steps_p2p: bad labels/slide range");
    MOC_REQUIRE(comm, "moc_train_steps_p2p: null communicator");
    // Which step kernel runs is decided from the run's constants alone (as moc_p2p_step_supported does), never from
    // this rank's bag sizes: the narrow and the wide kernel assign W1 elements to workgroups differently, and the
    // per-workgroup arrival flags of the exchange only mean something when every rank runs the same one.
    moc_batch_t Bu = *B;
    Bu.max_rows = 0x7fffffff;
    MOC_REQUIRE(fused_step_mode(&Bu, ws) != 0, "moc_train_steps_p2p: shape outside the one-launch steps (K <= 16, C <= 64, "
                "D in {512, 1024} beyond C = 16); use moc_train_steps_dp");
    hipStream_t s = (hipStream_t)stream;
    if (int rc = launch_w1_image(B, M, s)) return rc;
    moc_meta_t Mt = *M;
    float* cur = M->W2;
    float* nxt = ws->W2_alt;
    for (int t = 0; t < n; ++t) {
        const int b = slide0 + t;
        P2pArgs x;
        if (int rc = moc_p2p_next_args(comm, &x)) return rc;
        MOC_REQUIRE(x.n_par == (int64_t)H * M->D + H + 4 * H + 4, "moc_train_steps_p2p: communicator made for %lld parameters",
                    (long long)x.n_par);
        const AdamCoef k = adam_coef(M, M->step + 1 + t, 1.f / (float)x.world);
        Mt.W2 = cur;
        if (int rc = launch_forward(B, &Mt, ws, b, 1, use_bits, s)) return rc;
        if (int rc = launch_fused_step(&Bu, &Mt, ws, labels, b, use_bits, k, nxt, s, 1, &x)) return rc;
        float* tmp = cur; cur = nxt; nxt = tmp;
    }
    if (cur != M->W2) {
        if (hipMemcpyAsync(M->W2, cur, sizeof(float) * 4 * H, hipMemcpyDeviceToDevice, s) != hipSuccess)
            MOC_FAIL(MOC_ELAUNCH, "moc_train_steps_p2p: copy-back of W2 failed");
    }
    return MOC_OK;
}

struct moc_step_graph;
namespace {
int issue_fused_pass(const moc_batch_t* B, const moc_meta_t* M, const moc_meta_ws_t* ws, const int64_t* labels,
                     int slide0, int n, uint32_t use_bits, hipStream_t s, const moc_step_graph* G);
}

extern "C" int moc_train_steps(const moc_batch_t* B, const moc_meta_t* M, const moc_meta_ws_t* ws,
                               const int64_t* labels, int slide0, int n, uint32_t use_bits,
                               moc_stream_t stream) {
    if (int rc = moc_check_batch(B, "moc_train_steps")) return rc;
    if (int rc = check_meta(B, M, ws, "moc_train_steps", true, false)) return rc;
    MOC_REQUIRE(labels && slide0 >= 0 && n >= 1 && slide0 + n <= B->n_slides, "moc_train_steps: bad labels/slide range");
    hipStream_t s = (hipStream_t)stream;
    if (fused_step_mode(B, ws)) return issue_fused_pass(B, M, ws, labels, slide0, n, use_bits, s, nullptr);
    if (int rc = launch_w1_image(B, M, s)) return rc;      // afterwards the W1 update keeps it in sync
    for (int t = 0; t < n; ++t) {
        const int b = slide0 + t;
        const AdamCoef k = adam_coef(M, M->step + 1 + t, 1.f);
        if (int rc = launch_forward(B, M, ws, b, 1, use_bits, s)) return rc;
        if (int rc = launch_pool_finish(B, M, ws, labels, b, 1, 1, 1, use_bits, k, s)) return rc;
        if (int rc = launch_w1(B, M, ws, 1, k, s, fused_ok(B, 1))) return rc;
    }
    return MOC_OK;
}

// ------------------------------------------------------------------ batched runs: R meta-learners, lockstep
// The reference's real workload is folds x shots of the SAME loop as independent processes (scripts/moc_train.sh:11-31), and
// one run is a chain of dependent 17-us steps that leaves most of the GPU idle.  Here ONE forward launch and ONE step
// launch serve R runs: grid.y / grid.z = run.  Every run stays the exact recurrence of moc_train_steps -- its own slides,
// parameters, Adam moments; the same kernels, the same operation order: bit-identical to running it alone.
extern "C" int moc_train_runs_mode(const moc_batch_t* B, const moc_meta_ws_t* ws) {
    if (!B || !ws) return 0;
    if (tiles_ok(B, ws)) return 1;
    return fused_step_mode(B, ws) != 0 ? 2 : 0;
}

extern "C" int moc_train_steps_runs(const moc_batch_t* B, const moc_meta_t* M, const moc_runs_t* R, const moc_meta_ws_t* ws,
                                    const int64_t* labels, int slide0, int n, uint32_t use_bits, moc_stream_t stream) {
    if (int rc = moc_check_batch(B, "moc_train_steps_runs")) return rc;
    if (int rc = check_meta(B, M, ws, "moc_train_steps_runs", true, false)) return rc;
    MOC_REQUIRE(R && R->n_runs >= 1 && R->n_runs <= MOC_MAX_RUNS, "moc_train_steps_runs: 1 .. %d runs", MOC_MAX_RUNS);
    MOC_REQUIRE(labels && slide0 >= 0 && n >= 1 && R->slide_stride >= n &&
                slide0 + n + (R->n_runs - 1) * R->slide_stride <= B->n_slides, "moc_train_steps_runs: bad labels/slide range");
    MOC_REQUIRE(R->par_stride >= (int64_t)H * B->D + H + 4 * H + 4 && R->image_stride >= (int64_t)moc_w1_image_bytes(B->D, B->dtype),
                "moc_train_steps_runs: parameter / image stride smaller than one meta-learner");
    hipStream_t s = (hipStream_t)stream;
    if (!tiles_ok(B, ws)) {
        // Shapes outside the tile-record step (wide banks: EBRAINS-30, the 64-way shape; C > 16, C K > 64, more than 4,096
        // selectable rows): every run's pass through moc_train_steps' own launches, one run after the other on this stream --
        // the same kernels on the same tensors as the run alone, so the same bits.  What the caller still gains is phase A
        // over all runs' slides in one pass and (moc_amd.runs: one group per run) the runs' chains side by side on streams
        // of their own, each of which keeps a few dozen CUs busy.  The one-launch steps only (W2_alt set): the three-launch
        // step's scratch is one per batch.
        MOC_REQUIRE(fused_step_mode(B, ws) != 0, "moc_train_steps_runs: this shape needs ws->W2_alt (the one-launch steps)");
        for (int r = 0; r < R->n_runs; ++r) {
            moc_meta_t Mr = *M;
            const int64_t po = (int64_t)r * R->par_stride;
            Mr.W1 += po; Mr.b1 += po; Mr.W2 += po; Mr.b2 += po;
            Mr.m_W1 += po; Mr.m_b1 += po; Mr.m_W2 += po; Mr.m_b2 += po;
            Mr.v_W1 += po; Mr.v_b1 += po; Mr.v_W2 += po; Mr.v_b2 += po;
            Mr.g_W1 = Mr.g_b1 = Mr.g_W2 = Mr.g_b2 = nullptr;
            Mr.W1_image = (unsigned char*)M->W1_image + (int64_t)r * R->image_stride;
            moc_meta_ws_t wr = *ws;
            wr.W2_alt = ws->W2_alt + (int64_t)r * 4 * H;
            if (int rc = issue_fused_pass(B, &Mr, &wr, labels, slide0 + r * R->slide_stride, n, use_bits, s, nullptr)) return rc;
        }
        return MOC_OK;
    }
    fused_step_attrs();
    w1_image_kernel<<<dim3(H * B->D / 256, R->n_runs), 256, 0, s>>>(M->W1, B->D, (unsigned char*)M->W1_image, B->dtype, R->par_stride,
                                                                    R->image_stride);
    MOC_CHECK_LAUNCH("moc_w1_image(runs)");
    moc_meta_t Mt = *M;
    float* cur = M->W2;
    float* nxt = ws->W2_alt;
    int64_t cur_stride = R->par_stride, nxt_stride = 4 * H;
    for (int t = 0; t < n; ++t) {
        const int b = slide0 + t;
        const AdamCoef k = adam_coef(M, M->step + 1 + t, 1.f);
        Mt.W2 = cur;
        if (int rc = launch_forward(B, &Mt, ws, b, 1, use_bits, s, true, R, cur_stride)) return rc;
        TileStepArgsRuns tr;
        tr.s = tile_step_args(B, &Mt, ws, labels, b, use_bits, k, nxt, 1, nullptr);
        TileStepArgs& ta = tr.s;
        ta.n_runs = R->n_runs; ta.slide_stride = R->slide_stride;
        // four runs or more: throughput, not one run's latency, is what a launch is about -- no tail workgroups
        static const int tail_env = getenv("MOC_RUNS_TAIL_INSIDE") ? atoi(getenv("MOC_RUNS_TAIL_INSIDE")) : -1;   // diagnostic override
        ta.tail_inside = tail_env >= 0 ? tail_env : (R->n_runs >= 4 ? 1 : 0);
        ta.par_stride = R->par_stride; ta.img_stride = R->image_stride; ta.w2_stride = cur_stride; ta.w2out_stride = nxt_stride;
        static const bool prefetch = !(getenv("MOC_STEP_PREFETCH") && atoi(getenv("MOC_STEP_PREFETCH")) == 0);   // diagnostic: off
        // (launches of four runs or more are bound by workgroup slots, not by one run's latency: the prefetch cost eight
        // batched runs 3.6 %)
        if (prefetch && t + 1 < n && B->C < 4 && R->n_runs < 4) ta.prefetch_next = 1;
        for (int r = 0; r < R->n_runs; ++r) {
            const int sl = b + r * R->slide_stride;
            tr.base_r[r] = B->row_off_host[sl];
            int cap_, tb_;
            tile_region(B, sl, &tr.slot0_r[r], &cap_, &tb_);
            tr.cap_r[r] = cap_; tr.ntb_r[r] = tb_;
        }
        for (int r = R->n_runs; r < MOC_MAX_RUNS; ++r) { tr.base_r[r] = 0; tr.slot0_r[r] = 0; tr.cap_r[r] = 0; tr.ntb_r[r] = 0; }
        if (int rc = launch_tile_step_runs(B, tr, s)) return rc;
        float* tmp = cur; cur = nxt; nxt = tmp;
        const int64_t ts = cur_stride; cur_stride = nxt_stride; nxt_stride = ts;
    }
    if (cur != M->W2) {   // odd number of steps: every run's current W2 lives in the scratch buffer
        if (hipMemcpy2DAsync(M->W2, sizeof(float) * (size_t)R->par_stride, cur, sizeof(float) * 4 * H, sizeof(float) * 4 * H,
                             (size_t)R->n_runs, hipMemcpyDeviceToDevice, s) != hipSuccess)
            MOC_FAIL(MOC_ELAUNCH, "moc_train_steps_runs: copy-back of W2 failed");
    }
    return MOC_OK;
}

// ------------------------------------------------------------------ a pass of meta-steps as ONE graph launch
// moc_train_steps issues 2 n + 1 launches per pass (0.2 ms of host time for 32 steps).  In a long run they are
// issued ahead of the GPU and cost nothing; at the START of a run, or of a short timed region, the queue is empty and
// the chain cannot begin before the host has got through its prelude.  The handle below keeps the pass as an
// instantiated hipGraph per (work arrays, meta-learner tensors, slide range): a replay is one hipGraphLaunch.
// What changes from replay to replay -- the Adam step count, i.e. the bias corrections -- is not a kernel argument
// there: the coefficients of steps tab_base + 1 ... tab_base + cap sit in a device table (computed by the host with
// the arithmetic of adam_coef, so the same floats as the eager path) and a device counter holds how many of them have
// been used; the kernel at position t of the pass reads entry ctr + t, the graph's last node advances ctr by n.
struct moc_step_graph {
    int32_t* ctr;            // device: steps taken since tab_base (caller's workspace, first 16 bytes)
    AdamCoef* tab;           // device [cap]: entry i = coefficients of step tab_base + i + 1
    int cap;
    // host mirror
    bool tab_valid;
    double lr, beta1, beta2, eps, wd;
    int64_t tab_base;
    int64_t dev_rel;         // the counter's value once everything issued so far has run (-1: unknown)
    AdamCoef* stage;         // pinned staging of the table upload
    hipEvent_t stage_done;   // the upload that read `stage` has run
    bool stage_busy;
    hipStream_t cap_stream;  // passes are captured here (nothing ever runs on it): the caller's stream may be the
                             // legacy default stream, which cannot be captured
    struct Entry {
        unsigned char key[512];
        hipGraph_t graph;
        hipGraphExec_t exec;
        uint64_t used;
    } e[8];
    int n_entries;
    uint64_t tick;
    int eager_only;          // capture or instantiation failed once: the handle stays on stream launches
    int captures, replays, eager_calls;
    // the counter and the table are shared by every call: a call on ANOTHER stream than the previous one waits for it
    hipStream_t last_stream;
    hipEvent_t last_done;
    bool has_last;
};

namespace {

__global__ void step_ctr_set_kernel(int32_t* ctr, int32_t v) { ctr[0] = v; }
__global__ void step_ctr_add_kernel(int32_t* ctr, int32_t n) { ctr[0] += n; }

struct GraphKey {            // everything a captured pass bakes into its kernel arguments
    const void *X, *row_off, *sel_row, *n_sel, *cand, *labels;
    const void *W1, *b1, *W2, *b2, *mW1, *mb1, *mW2, *mb2, *vW1, *vb1, *vW2, *vb2, *img;
    const void *H1, *gates, *mixed, *pooled, *topk_idx, *topk_cnt, *loss, *pred, *pair_dh, *W2_alt, *pair_row, *n_pair, *tile_ws;
    int64_t total_rows, tile_ws_bytes;
    uint64_t off_hash;       // FNV-1a of row_off_host[slide0 .. slide0 + n]
    int32_t dtype, D, n_slides, C, Ce, topj, topk, s_bound, mode, external, slide0, n;
    uint32_t use_bits, flags;  // flags: MOC_FORWARD_ROWS64 / MOC_FORWARD_FOUR_WAVES pick the forward kernel a capture bakes in
};
static_assert(sizeof(GraphKey) <= 512, "GraphKey outgrew moc_step_graph::Entry::key");

void graph_key(GraphKey* k, const moc_batch_t* B, const moc_meta_t* M, const moc_meta_ws_t* ws, const int64_t* labels,
               int slide0, int n, uint32_t use_bits, int mode) {
    memset(k, 0, sizeof(*k));
    k->X = B->X; k->row_off = B->row_off; k->sel_row = B->sel_row; k->n_sel = B->n_sel; k->cand = B->cand; k->labels = labels;
    k->W1 = M->W1; k->b1 = M->b1; k->W2 = M->W2; k->b2 = M->b2; k->mW1 = M->m_W1; k->mb1 = M->m_b1; k->mW2 = M->m_W2;
    k->mb2 = M->m_b2; k->vW1 = M->v_W1; k->vb1 = M->v_b1; k->vW2 = M->v_W2; k->vb2 = M->v_b2; k->img = M->W1_image;
    k->H1 = ws->H1; k->gates = ws->gates; k->mixed = ws->mixed; k->pooled = ws->pooled; k->topk_idx = ws->topk_idx;
    k->topk_cnt = ws->topk_cnt; k->loss = ws->loss; k->pred = ws->pred; k->pair_dh = ws->pair_dh; k->W2_alt = ws->W2_alt;
    k->pair_row = ws->pair_row; k->n_pair = ws->n_pair; k->tile_ws = ws->tile_ws; k->tile_ws_bytes = ws->tile_ws_bytes;
    k->total_rows = B->total_rows;
    uint64_t h = 1469598103934665603ull;
    if (B->row_off_host)
        for (int i = slide0; i <= slide0 + n; ++i) { h ^= (uint64_t)B->row_off_host[i]; h *= 1099511628211ull; }
    k->off_hash = B->row_off_host ? h : 0;
    k->dtype = B->dtype; k->D = B->D; k->n_slides = B->n_slides; k->C = B->C; k->topk = B->topk; k->s_bound = s_bound(B);
    k->mode = mode;
    k->external = mode == 2 && (s_bound(B) > 8192 || (int64_t)B->C * s_bound(B) > 49152);
    k->slide0 = slide0; k->n = n; k->use_bits = use_bits;
    k->Ce = B->Ce; k->topj = B->topj; k->flags = B->flags;
}

// the 2 n + 1 launches of a fused-step pass; `tab` != null: coefficients from the device table (capture)
int issue_fused_pass(const moc_batch_t* B, const moc_meta_t* M, const moc_meta_ws_t* ws, const int64_t* labels,
                     int slide0, int n, uint32_t use_bits, hipStream_t s, const moc_step_graph* G) {
    if (int rc = launch_w1_image(B, M, s)) return rc;      // afterwards the W1 update keeps it in sync
    // two launches per meta-step: forward, then pooling + loss + backward + the whole Adam step.
    // W2 is read by every workgroup of the second kernel while workgroup 0 steps it: ping-pong.
    moc_meta_t Mt = *M;
    float* cur = M->W2;
    float* nxt = ws->W2_alt;
    for (int t = 0; t < n; ++t) {
        const int b = slide0 + t;
        AdamCoef k = {};
        StepTab st = {};
        if (G) { st.tab = G->tab; st.ctr = G->ctr; st.pos = t; }
        else k = adam_coef(M, M->step + 1 + t, 1.f);
        Mt.W2 = cur;
        if (int rc = launch_forward(B, &Mt, ws, b, 1, use_bits, s, true)) return rc;
        if (int rc = launch_fused_step(B, &Mt, ws, labels, b, use_bits, k, nxt, s, 1, nullptr, G ? &st : nullptr, t + 1 < n ? b + 1 : -1)) return rc;
        float* tmp = cur; cur = nxt; nxt = tmp;
    }
    if (cur != M->W2) {   // odd number of steps: the current W2 lives in the scratch buffer
        if (hipMemcpyAsync(M->W2, cur, sizeof(float) * 4 * H, hipMemcpyDeviceToDevice, s) != hipSuccess)
            MOC_FAIL(MOC_ELAUNCH, "moc_train_steps: copy-back of W2 failed");
    }
    if (G) {
        step_ctr_add_kernel<<<1, 1, 0, s>>>(G->ctr, n);
        MOC_CHECK_LAUNCH("moc_step_ctr_add");
    }
    return MOC_OK;
}

}  // namespace

extern "C" size_t moc_step_graph_workspace_bytes(int max_steps) {
    return max_steps > 0 ? 16 + sizeof(AdamCoef) * (size_t)max_steps : 0;
}

extern "C" int moc_step_graph_create(void* device_ws, size_t ws_bytes, moc_step_graph_t** out) {
    MOC_REQUIRE(out && device_ws && ((uintptr_t)device_ws & 15) == 0, "moc_step_graph_create: null or unaligned workspace");
    MOC_REQUIRE(ws_bytes >= 16 + sizeof(AdamCoef) * 64, "moc_step_graph_create: workspace of %zu bytes holds fewer than 64 steps", ws_bytes);
    moc_step_graph* G = new (std::nothrow) moc_step_graph();
    MOC_REQUIRE(G, "moc_step_graph_create: out of host memory");
    memset(G, 0, sizeof(*G));
    G->ctr = (int32_t*)device_ws;
    G->tab = (AdamCoef*)((unsigned char*)device_ws + 16);
    G->cap = (int)((ws_bytes - 16) / sizeof(AdamCoef));
    G->dev_rel = -1;
    if (hipHostMalloc((void**)&G->stage, sizeof(AdamCoef) * (size_t)G->cap, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&G->stage_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&G->last_done, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&G->cap_stream, hipStreamNonBlocking) != hipSuccess) {
        if (G->stage) (void)hipHostFree(G->stage);
        delete G;
        MOC_FAIL(MOC_ELAUNCH, "moc_step_graph_create: cannot allocate the pinned staging buffer");
    }
    *out = G;
    return MOC_OK;
}

extern "C" int moc_step_graph_destroy(moc_step_graph_t* G) {
    if (!G) return MOC_OK;
    for (int i = 0; i < G->n_entries; ++i) {
        (void)hipGraphExecDestroy(G->e[i].exec);
        (void)hipGraphDestroy(G->e[i].graph);
    }
    if (G->stage_busy) (void)hipEventSynchronize(G->stage_done);
    (void)hipEventDestroy(G->stage_done);
    (void)hipEventDestroy(G->last_done);
    (void)hipStreamDestroy(G->cap_stream);
    (void)hipHostFree(G->stage);
    delete G;
    return MOC_OK;
}

extern "C" int moc_step_graph_stats(const moc_step_graph_t* G, int* captures, int* replays, int* eager_calls) {
    MOC_REQUIRE(G, "moc_step_graph_stats: null handle");
    if (captures) *captures = G->captures;
    if (replays) *replays = G->replays;
    if (eager_calls) *eager_calls = G->eager_calls;
    return MOC_OK;
}

extern "C" int moc_train_steps_graph(moc_step_graph_t* G, const moc_batch_t* B, const moc_meta_t* M, const moc_meta_ws_t* ws,
                                     const int64_t* labels, int slide0, int n, uint32_t use_bits, moc_stream_t stream) {
    MOC_REQUIRE(G, "moc_train_steps_graph: null handle");
    if (int rc = moc_check_batch(B, "moc_train_steps_graph")) return rc;
    if (int rc = check_meta(B, M, ws, "moc_train_steps_graph", true, false)) return rc;
    MOC_REQUIRE(labels && slide0 >= 0 && n >= 1 && slide0 + n <= B->n_slides, "moc_train_steps_graph: bad labels/slide range");
    hipStream_t s = (hipStream_t)stream;
    // Grids and kernel shapes follow B->max_rows, which a masked batch tightens to the largest KEPT-row count of the
    // pass (moc_host_max_kept): a different number every pass.  A captured pass must not depend on it, so the graph is
    // built for the largest SLIDE of the range -- an upper bound of every pass's kept rows (surplus workgroups leave at
    // once on n_sel).
    moc_batch_t Bg = *B;
    if (B->row_off_host) {
        int64_t mx = 1;
        for (int i = slide0; i < slide0 + n; ++i) {
            const int64_t r = B->row_off_host[i + 1] - B->row_off_host[i];
            if (r > mx) mx = r;
        }
        Bg.max_rows = (int32_t)(mx < 0x7fffffff ? mx : 0x7fffffff);
    }
    const moc_batch_t* Bo = B;     // as handed in: for the stream-launch fall-back
    B = &Bg;
    const int mode = fused_step_mode(B, ws);
    if (!mode || G->eager_only || n > G->cap) {            // shapes of the three-launch step, or a handle that gave up
        G->eager_calls++;
        G->dev_rel = -1;
        return moc_train_steps(Bo, M, ws, labels, slide0, n, use_bits, stream);
    }
    fused_step_attrs();
    if (G->has_last && G->last_stream != s) {              // the previous pass ran elsewhere: order the two on the device
        if (hipStreamWaitEvent(s, G->last_done, 0) != hipSuccess) MOC_FAIL(MOC_ELAUNCH, "moc_train_steps_graph: stream wait");
    }
    // ---- the coefficient table covers steps M->step + 1 ... M->step + n with these hyper-parameters?
    const bool same_hp = G->tab_valid && G->lr == M->lr && G->beta1 == M->beta1 && G->beta2 == M->beta2 &&
                         G->eps == M->eps && G->wd == M->weight_decay;
    if (!same_hp || M->step < G->tab_base || M->step + n > G->tab_base + G->cap) {
        if (G->stage_busy) {                               // the previous upload still owns the staging buffer
            if (hipEventSynchronize(G->stage_done) != hipSuccess) MOC_FAIL(MOC_ELAUNCH, "moc_train_steps_graph: staging event");
            G->stage_busy = false;
        }
        for (int i = 0; i < G->cap; ++i) G->stage[i] = adam_coef(M, M->step + 1 + i, 1.f);
        if (hipMemcpyAsync(G->tab, G->stage, sizeof(AdamCoef) * (size_t)G->cap, hipMemcpyHostToDevice, s) != hipSuccess ||
            hipEventRecord(G->stage_done, s) != hipSuccess)
            MOC_FAIL(MOC_ELAUNCH, "moc_train_steps_graph: upload of the coefficient table failed");
        G->stage_busy = true;
        G->tab_valid = true;
        G->lr = M->lr; G->beta1 = M->beta1; G->beta2 = M->beta2; G->eps = M->eps; G->wd = M->weight_decay;
        G->tab_base = M->step;
        G->dev_rel = -1;
    }
    // ---- the device counter stands where this optimizer's step count is?  (someone else may have stepped it, or a
    // checkpoint was loaded: one tiny launch puts it right, value by kernel argument)
    const int64_t rel = M->step - G->tab_base;
    if (G->dev_rel != rel) {
        step_ctr_set_kernel<<<1, 1, 0, s>>>(G->ctr, (int32_t)rel);
        MOC_CHECK_LAUNCH("moc_step_ctr_set");
        G->dev_rel = rel;
    }
    {   // diagnostic: the table-reading kernels as plain stream launches (separates what the table costs from what the graph costs)
        static const char* mode_env = getenv("MOC_STEP_GRAPH_MODE");
        if (mode_env && strcmp(mode_env, "table") == 0) {
            if (int rc = issue_fused_pass(B, M, ws, labels, slide0, n, use_bits, s, G)) return rc;
            G->eager_calls++;
            G->dev_rel = rel + n;
            G->has_last = hipEventRecord(G->last_done, s) == hipSuccess;
            G->last_stream = s;
            return MOC_OK;
        }
    }
    // ---- find or capture the pass
    GraphKey key;
    graph_key(&key, B, M, ws, labels, slide0, n, use_bits, mode);
    moc_step_graph::Entry* hit = nullptr;
    for (int i = 0; i < G->n_entries; ++i)
        if (memcmp(G->e[i].key, &key, sizeof(key)) == 0) { hit = &G->e[i]; break; }
    if (!hit) {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        int rc = MOC_OK;
        hipError_t he = hipStreamBeginCapture(G->cap_stream, hipStreamCaptureModeRelaxed);
        if (he != hipSuccess) rc = MOC_ELAUNCH;
        else {
            rc = issue_fused_pass(B, M, ws, labels, slide0, n, use_bits, G->cap_stream, G);
            he = hipStreamEndCapture(G->cap_stream, &graph);          // (always: leaves the stream out of capture mode)
            if (rc == MOC_OK && (he != hipSuccess || !graph)) rc = MOC_ELAUNCH;
        }
        if (rc == MOC_OK && (he = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0)) != hipSuccess) rc = MOC_ELAUNCH;
        if (rc != MOC_OK) {                                // no graph on this runtime: stream launches from now on
            moc_set_error("moc_train_steps_graph: capture / instantiation failed (%s); falling back to stream launches",
                          hipGetErrorString(he));
            if (graph) (void)hipGraphDestroy(graph);
            (void)hipGetLastError();
            G->eager_only = 1;
            G->eager_calls++;
            G->dev_rel = -1;
            return moc_train_steps(Bo, M, ws, labels, slide0, n, use_bits, stream);
        }
        if (G->n_entries < 8) hit = &G->e[G->n_entries++];
        else {                                             // replace the entry that was used longest ago
            hit = &G->e[0];
            for (int i = 1; i < 8; ++i) if (G->e[i].used < hit->used) hit = &G->e[i];
            (void)hipGraphExecDestroy(hit->exec);
            (void)hipGraphDestroy(hit->graph);
        }
        memset(hit->key, 0, sizeof(hit->key));
        memcpy(hit->key, &key, sizeof(key));
        hit->graph = graph;
        hit->exec = exec;
        G->captures++;
    }
    hit->used = ++G->tick;
    if (hipGraphLaunch(hit->exec, s) != hipSuccess) MOC_FAIL(MOC_ELAUNCH, "moc_train_steps_graph: hipGraphLaunch failed");
    G->replays++;
    G->dev_rel = rel + n;
    G->has_last = hipEventRecord(G->last_done, s) == hipSuccess;
    G->last_stream = s;
    return MOC_OK;
}

extern "C" size_t moc_tile_ws_bytes(int64_t total_rows, int n_slides, int C) {
    if (total_rows <= 0 || n_slides <= 0 || C <= 0) return 0;
    return tile_bytes(tile_slots(total_rows, n_slides, C));
}

extern "C" size_t moc_w1_image_bytes(int D, int dtype) {
    if (D <= 0) return 0;
    return dtype != MOC_F32 ? (size_t)D * H * 3 * 2 : (size_t)D * H * 4;
}
