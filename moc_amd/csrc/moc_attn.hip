// Gated-attention MIL pooling (reference models/model_clam.py:41-64 Attn_Net_Gated,
// :178-183 / :206 CLAM_SB.forward_single, :291-296 / :318 CLAM_MB.forward):
//
//   a = tanh(h Wa^T + ba),  b = sigmoid(h Wb^T + bb)                 [N, D]
//   A_raw[k][n] = Wc[k] . (a[n] * b[n]) + bc[k]                      [K, N]   (K = 1: CLAM_SB, n_classes: CLAM_MB)
//   M[k]        = sum_n softmax_n(A_raw[k])[n] * h[n]                [K, L]
//
// One pass over the bag.  A workgroup owns 64 rows; both projections run on the bf16 matrix cores with
// fp32-exact products: activations and weights are each held as three bf16 terms (8 + 8 + 8 mantissa bits,
// hi + mid + lo == the fp32 value exactly) and the six products down to 2^-16 of the leading one are
// accumulated in fp32 (v_mfma_f32_16x16x32_bf16: 6 instructions of 16 cycles replace 8 fp32-MFMAs of 32).
// Wave w owns ALL 64 rows x the D/4 gate columns [w D/4, (w+1) D/4) of both projections (so the gate is
// wave-local and every weight fragment is used by four row tiles): its weight fragments are private, they
// stream global -> LDS by LDS-DMA into a per-wave ring of five 6 KiB chunks (one a-tile + one b-tile x three
// terms x 32 columns of L) and are awaited with a counted vmcnt -- no barrier; only the 64 x 32 slab of h
// is shared (three raw fp32 slots, one barrier per slab) and split into bf16 terms in registers.  The gate,
// the K dot products with Wc, their 16-lane reductions and the sum over the four waves follow; then the
// workgroup forms its share of the softmax in the online form -- m = max score, l = sum exp(score - m),
// M' = sum exp(score - m) * h -- re-reading its 64 rows (still in L2).  A second, tiny launch merges the
// per-workgroup (m, l, M') triples.  Neither the [N, D] activations nor the softmax weights reach memory.
//
// Bound: the bf16 matrix pipe at 6 products per fp32 product: 4 * N * L * D * 6 flops against 2.5 PFLOP/s
// dense -- N = 15,000, L = 512, D = 384: 70.8 GFLOP -> 28 us (the fp32 matrix pipe would need 75 us); the
// bag is 30.7 MB (6 us of HBM time); each workgroup streams the 6 * D * L-byte weight image from L2 once.
#include "moc_common.h"
#include <type_traits>

namespace {

constexpr int AT_ROWS = 64;              // rows per workgroup
constexpr int AT_RING = 5;               // weight chunks per wave in LDS (four in flight behind the one in use)
constexpr int AT_CHUNK = 6 * 1024;       // (a-tile, b-tile) x 3 terms x 64 lanes x 16 B
constexpr int AT_ASLAB = 8 * 1024;       // 64 rows x 32 columns fp32: [row tile 4][piece 2][lane 64][16 B]
constexpr int AT_ASLOTS = 3;
constexpr int AT_SMEM = 4 * AT_RING * AT_CHUNK + AT_ASLOTS * AT_ASLAB;      // 144 KiB, whatever D and K

struct AttnArgs {
    const float* h;                  // [N, L]
    const unsigned char* img;        // weight image (attn_image_kernel)
    const float *ba, *bb, *Wc, *bc;  // [D] [D] [K, D] [K]
    float* A_raw;                    // [K, N]
    float *ws_m, *ws_l, *ws_M;       // [G, K] [G, K] [G, K, L]
    int64_t N;
    int L, D, K;
};
// row stride of the backward's `dab`: the two gradients, then the K softmax columns, padded to 16 floats
__host__ __device__ constexpr int attn_dab_stride(int D, int K) { return 2 * D + 16 * ((K + 15) / 16); }
struct AttnBwdArgs {                 // the recompute pass of the backward (kernel arguments live in scalar registers:
    const float* h;                  // the forward keeps its own, shorter list)
    const unsigned char* img;
    const float *ba, *bb, *Wc;
    const float* ds;                 // [K, N] gradient arriving at A_raw (through the softmax and directly)
    float* dab;                      // [N, attn_dab_stride] gradients at the two pre-activations, (a | b) side by side
    float* colpart;                  // [G][(2 + K) D] per-workgroup column sums: d_ba | d_bb | d_Wc[0..K)
    int64_t N;
    int L, D, K;
};

// image: [t = ceil(L/32)][wave 4][c = D/64][j = 3 s + term][lane][8 x bf16]; s = 0: Wa, 1: Wb; rows
// n = (wave * D/64 + c) * 16 + (lane & 15) of W, columns 32 t + (lane >> 4) * 8 + 0..7 (zero beyond L): the B
// operand of one 16x16x32 MFMA.  One thread writes the three terms of one fragment.
__global__ __launch_bounds__(256) void attn_image_kernel(const float* Wa, const float* Wb, int L, int D, uint4* img) {
    const int CH = D / 64, T = (L + 31) / 32;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)T * 4 * CH * 2 * 64) return;
    const int lane = (int)(idx & 63), s = (int)((idx >> 6) & 1);
    int64_t rest = idx >> 7;
    const int c = (int)(rest % CH);
    rest /= CH;
    const int wave = (int)(rest & 3), t = (int)(rest >> 2);
    const float* W = s ? Wb : Wa;
    const int n = (wave * CH + c) * 16 + (lane & 15), k0 = t * 32 + (lane >> 4) * 8;
    float w[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (k0 < L) {                                                     // L % 8 == 0: a fragment is all in or all out
        const float4 v0 = *reinterpret_cast<const float4*>(W + (int64_t)n * L + k0);
        const float4 v1 = *reinterpret_cast<const float4*>(W + (int64_t)n * L + k0 + 4);
        w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w; w[4] = v1.x; w[5] = v1.y; w[6] = v1.z; w[7] = v1.w;
    }
    uint16_t hi[8], mid[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) moc_split3<false>(w[e], 1.f, hi[e], mid[e], lo[e]);
    auto pack = [](const uint16_t (&v)[8]) {
        return uint4{(uint32_t)v[0] | ((uint32_t)v[1] << 16), (uint32_t)v[2] | ((uint32_t)v[3] << 16),
                     (uint32_t)v[4] | ((uint32_t)v[5] << 16), (uint32_t)v[6] | ((uint32_t)v[7] << 16)};
    };
    const int64_t base = ((((int64_t)t * 4 + wave) * CH + c) * 6 + s * 3) * 64 + lane;
    img[base] = pack(hi);
    img[base + 64] = pack(mid);
    img[base + 128] = pack(lo);
}

// Every LDS read of the main loop is issued by hand: an ordinary read would make hipcc wait vmcnt(0) first
// (an LDS-DMA in flight writes LDS and it cannot tell the slots apart) and the overlap would be gone.
typedef unsigned __attribute__((ext_vector_type(4))) au32x4_t;
template <int OFF>
__device__ __forceinline__ void at_lds16(au32x4_t& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void at_touch(au32x4_t& x) { asm volatile("" : "+v"(x)); }
template <int VM>
__device__ __forceinline__ void at_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory"); }
__device__ __forceinline__ void at_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <int VM>
__device__ __forceinline__ void at_barrier() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(VM) : "memory"); }

// three bf16 terms of eight fp32 values (truncating split: x == hi + mid + lo exactly), packed in MFMA
// operand order.  v_perm_b32 selector 0x07060302: the upper halves of (odd, even) side by side.
struct AtTerms { au32x4_t t[3]; };
// pair q (of four) of a fragment: elements 2q, 2q + 1 of the eight in (p0, p1) -- 11 vector instructions
template <int Q>
__device__ __forceinline__ void at_split_pair(const au32x4_t& p0, const au32x4_t& p1, AtTerms& o) {
    const unsigned e0 = Q < 2 ? p0[2 * Q] : p1[2 * Q - 4], e1 = Q < 2 ? p0[2 * Q + 1] : p1[2 * Q - 3];
    const float r0 = __uint_as_float(e0) - __uint_as_float(e0 & 0xFFFF0000u);
    const float r1 = __uint_as_float(e1) - __uint_as_float(e1 & 0xFFFF0000u);
    const unsigned r0b = __float_as_uint(r0), r1b = __float_as_uint(r1);
    const float l0 = r0 - __uint_as_float(r0b & 0xFFFF0000u);
    const float l1 = r1 - __uint_as_float(r1b & 0xFFFF0000u);
    o.t[0][Q] = __builtin_amdgcn_perm(e1, e0, 0x07060302u);
    o.t[1][Q] = __builtin_amdgcn_perm(r1b, r0b, 0x07060302u);
    o.t[2][Q] = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
}
__device__ __forceinline__ void at_split(const au32x4_t& p0, const au32x4_t& p1, AtTerms& o) {
    at_split_pair<0>(p0, p1, o); at_split_pair<1>(p0, p1, o); at_split_pair<2>(p0, p1, o); at_split_pair<3>(p0, p1, o);
}
// unit U of 16 = (row tile U / 4, pair U % 4) of a slab's raw rows xr[8]
template <int U>
__device__ __forceinline__ void at_split_unit(const au32x4_t (&xr)[8], AtTerms (&o)[4]) {
    at_split_pair<U % 4>(xr[2 * (U / 4)], xr[2 * (U / 4) + 1], o[U / 4]);
}
// the units of slot `slot` of `slots` (both constants once the chunk loop is unrolled): [16 slot / slots, 16 (slot + 1) / slots)
template <int U>
__device__ __forceinline__ void at_deal_from(int lo, int hi, const au32x4_t (&xr)[8], AtTerms (&o)[4]) {
    if constexpr (U < 16) {
        if (U >= lo && U < hi) at_split_unit<U>(xr, o);
        at_deal_from<U + 1>(lo, hi, xr, o);
    }
}
// The next slab's raw rows are read one row tile per chunk and split during the chunk after (32 raw
// registers would not fit beside 192 accumulators): chunk c reads at_tiles_n(c, CH) tiles starting at tile c.
__host__ __device__ constexpr int at_tiles_n(int c, int CH) {
    return c < 0 ? 0 : c < CH - 2 ? (c < 4 ? 1 : 0) : c == CH - 2 ? (4 - c > 0 ? 4 - c : 0) : 0;
}
// MFMA group g (of six) of chunk c splits its share of the units of the tiles read in chunk c - 1
__device__ __forceinline__ void at_deal(int c, int CH, int g, const au32x4_t (&xr)[8], AtTerms (&o)[4]) {
    const int n = 4 * at_tiles_n(c - 1, CH), u0 = 4 * (c - 1);
    at_deal_from<0>(u0 + n * g / 6, u0 + n * (g + 1) / 6, xr, o);
}

__device__ __forceinline__ f32x4_t at_mfma(const au32x4_t& x, const au32x4_t& w, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, x), __builtin_bit_cast(bf16x8_t, w), c, 0, 0, 0);
}

// gate nonlinearities on the hardware exp / rcp (v_exp_f32, v_rcp_f32: ~1e-7 absolute on outputs in [-1, 1])
__device__ __forceinline__ float fast_sigmoid(float x) { return __frcp_rn(1.f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * x)); }
// tanh(x) * sigmoid(y) = (e^2x - 1) / ((e^2x + 1)(1 + e^-y)) with one reciprocal; tanh(|x| > 15) is +-1 in fp32
__device__ __forceinline__ float gate_fn(float x, float y) {
    const float e = __expf(2.f * fminf(fmaxf(x, -15.f), 15.f)), f = __expf(-y);
    return (e - 1.f) * __frcp_rn((e + 1.f) * (1.f + f));
}

// vmcnt bookkeeping of one wave (loads complete in order).  Issue order: prologue A(0) A(1) B(0..3); then, slab t:
// A(t+2), and per chunk g: B(g+4).  A = 2 instructions (this wave's 16 rows of a slab), B = 6 (a chunk).
//   * chunk g waits for B(g+1), which it reads into registers one chunk ahead of its use: behind B(g+1) only
//     B(g+2..g+4) = 18 instructions may remain outstanding (+2 when an A issue lies between: vmcnt(18) then also
//     asks for the first third of B(g+2), issued two chunks ago -- stricter, never wrong);
//   * slab t+1 reads the raw rows A(t+2) of EVERY wave at its first chunk: behind A(t+2), issued at the start of
//     slab t, follow 6 CH instructions, so the barrier that ends slab t waits vmcnt(min(6 CH, 18)) first;
//   * the tail re-fetches chunks and slabs nobody reads (wrapped indices): the counts stay the same to the end,
//     and the final barrier drains them (vmcnt(0)) before the ring is reused by the epilogue.
//
// BWD = true is the recompute pass of the backward (moc_gated_attention_backward): the same main loop leaves the
// pre-activations in the accumulators; the epilogue then turns them, in registers, into the gradients at the two
// pre-activations -- dg = sum_k ds[k][n] Wc[k][d], da' = dg b (1 - a^2), db' = dg a b (1 - b) -- stores those
// ([N, 2 D], the operand of the three plain GEMMs that remain: dWa | dWb = dab^T h, dh = dab [Wa; Wb]) and writes this
// workgroup's column sums (d_ba, d_bb, d_Wc[k][d] = sum_n ds[k][n] a b) for a fixed-order merge.
template <int ND, bool BWD>
__global__ __launch_bounds__(256, 1) void gated_attention_kernel(std::conditional_t<BWD, AttnBwdArgs, AttnArgs> a) {
    constexpr int CH = ND / 4;                                        // chunks per slab = (a, b) tile pairs per wave
    static_assert(CH % 2 == 0, "the chunk loop alternates two register sets");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;                                       // [4 waves][AT_RING][AT_CHUNK]
    unsigned char* araw = smem + 4 * AT_RING * AT_CHUNK;              // [AT_ASLOTS][AT_ASLAB]
    const int lane = threadIdx.x & 63, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // scalar: LDS-DMA destinations go to M0
    const int L = a.L, K = a.K, T = (L + 31) / 32;
    const int64_t row0 = (int64_t)blockIdx.x * AT_ROWS;
    const int64_t my_row = row0 + wave * 16 + (lane & 15);
    const float* hp = a.h + (my_row < a.N ? my_row : a.N - 1) * L;    // clamp: loads stay in bounds
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    // accumulators start at the bias of their column (column = lane & 15 in every register of a tile)
    f32x4_t acc[CH][2][4];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int d = (wave * CH + c) * 16 + (lane & 15);
        const float wa = a.ba[d], wb = a.bb[d];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[c][0][r] = f32x4_t{wa, wa, wa, wa};
            acc[c][1][r] = f32x4_t{wb, wb, wb, wb};
        }
    }

    // weight chunks in image order, wrapping at the end (the last four issues re-fetch chunks nobody reads:
    // the counts stay uniform)
    const unsigned char* wsrc = a.img + (int64_t)wave * CH * AT_CHUNK + lane * 16;
    int it = 0, ic = 0, islot = 0;
    auto issue_B = [&]() {
        const unsigned char* src = wsrc + ((int64_t)it * 4 * CH + ic) * AT_CHUNK;
        unsigned char* dst = ring + (wave * AT_RING + islot) * AT_CHUNK;
#pragma unroll
        for (int i = 0; i < 6; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(src + i * 1024), (lptr_t)(dst + i * 1024), 16, 0, 0);
        if (++ic == CH) { ic = 0; if (++it == T) it = 0; }
        if (++islot == AT_RING) islot = 0;
    };
    // this wave's 16 rows of slab t: lane (row, kq) fetches h[row][32 t + 8 kq + 4 p ..+3], p = 0, 1 (columns
    // beyond L are re-aimed at column 0: finite data against zero weights)
    auto issue_A = [&](int t, int slot) {
        if (t > T - 1) t = T - 1;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            int col = t * 32 + kq * 8 + p * 4;
            if (col >= L) col = 0;
            __builtin_amdgcn_global_load_lds((gptr_t)(hp + col), (lptr_t)(araw + slot * AT_ASLAB + (wave * 2 + p) * 1024), 16, 0, 0);
        }
    };
    const unsigned ring_base = (unsigned)(uintptr_t)(ring + wave * AT_RING * AT_CHUNK + lane * 16);
    const unsigned araw_base = (unsigned)(uintptr_t)(araw + lane * 16);
    auto read_B = [&](au32x4_t (&b)[6], int slot) {
        const unsigned ad = ring_base + slot * AT_CHUNK;
        at_lds16<0>(b[0], ad); at_lds16<1024>(b[1], ad); at_lds16<2048>(b[2], ad);
        at_lds16<3072>(b[3], ad); at_lds16<4096>(b[4], ad); at_lds16<5120>(b[5], ad);
    };
    auto read_A_tiles = [&](au32x4_t (&x)[8], int slot, int r0, int n) {         // row tiles [r0, r0 + n), constants when unrolled
        const unsigned ad = araw_base + slot * AT_ASLAB;
        if (r0 <= 0 && 0 < r0 + n) { at_lds16<0>(x[0], ad); at_lds16<1024>(x[1], ad); }
        if (r0 <= 1 && 1 < r0 + n) { at_lds16<2048>(x[2], ad); at_lds16<3072>(x[3], ad); }
        if (r0 <= 2 && 2 < r0 + n) { at_lds16<4096>(x[4], ad); at_lds16<5120>(x[5], ad); }
        if (r0 <= 3 && 3 < r0 + n) { at_lds16<6144>(x[6], ad); at_lds16<7168>(x[7], ad); }
    };
    auto read_A = [&](au32x4_t (&x)[8], int slot) {
        const unsigned ad = araw_base + slot * AT_ASLAB;
        at_lds16<0>(x[0], ad); at_lds16<1024>(x[1], ad); at_lds16<2048>(x[2], ad); at_lds16<3072>(x[3], ad);
        at_lds16<4096>(x[4], ad); at_lds16<5120>(x[5], ad); at_lds16<6144>(x[6], ad); at_lds16<7168>(x[7], ad);
    };

    MOC_STAMP(30);
    issue_A(0, 0);
    issue_A(1, 1);
    issue_B(); issue_B(); issue_B(); issue_B();
    at_barrier<18>();                                                 // A(0), A(1), B(0) landed, everywhere
    AtTerms xa[4], xn[4];
    au32x4_t b0[6], b1[6], xr[8];
    read_A(xr, 0);
    read_B(b0, 0);
    at_wait_lds();
#pragma unroll
    for (int r = 0; r < 4; ++r) at_split(xr[2 * r], xr[2 * r + 1], xa[r]);
#pragma unroll
    for (int i = 0; i < 6; ++i) at_touch(b0[i]);

    MOC_STAMP(31);
    int rslot = 1;                                                    // ring slot of the chunk AFTER the current one
    // one slab: MFMAs on `xc` (this slab's terms) while the next slab's raw rows are read and split into `xnx`
    auto slab = [&](int t, AtTerms (&xc)[4], AtTerms (&xnx)[4]) {
        issue_A(t + 2, (t + 2) % AT_ASLOTS);                          // its slot was last read two slabs ago
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            au32x4_t (&bc_)[6] = (c & 1) ? b1 : b0;                   // current chunk (read one chunk ago)
            au32x4_t (&bn_)[6] = (c & 1) ? b0 : b1;                   // next chunk
            issue_B();
            at_wait_vm<18>();                                         // the next chunk has landed
            read_B(bn_, rslot);
            if (++rslot == AT_RING) rslot = 0;
            read_A_tiles(xr, (t + 1) % AT_ASLOTS, c, at_tiles_n(c, CH));     // next slab of h (published by the last barrier)
            // Six products per (tile, row tile), eight independent accumulators in turn.  The scheduling fence keeps
            // the block above the wait for the reads just issued (left alone, hipcc sinks most of the MFMAs
            // below the wait and the LDS latency is exposed; a second fence above the block costs registers: ND = 24
            // then spills, and a scratch access inside the loop would also break the vmcnt arithmetic) -- and the split of the
            // next slab's rows (16 units of 11 vector instructions) is dealt out between the MFMA groups: an MFMA
            // leaves 8 of its 16 issue cycles to vector work.
#define AT_GROUP(PR, TAV, TBV)                                                                                        \
            _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                             \
                _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                         \
                    acc[c][s][r] = at_mfma(xc[r].t[TAV], bc_[s * 3 + TBV], acc[c][s][r]);                             \
            at_deal(c, CH, PR, xr, xnx);
            AT_GROUP(0, 2, 0)                                         // smallest products first
            AT_GROUP(1, 0, 2)
            AT_GROUP(2, 1, 1)
            AT_GROUP(3, 1, 0)
            AT_GROUP(4, 0, 1)
            AT_GROUP(5, 0, 0)
#undef AT_GROUP
#pragma unroll
            for (int i = 0; i < 48; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one MFMA ...
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);    // ... two vector instructions in its shadow
            }
            __builtin_amdgcn_sched_barrier(0);
            at_wait_lds();
#pragma unroll
            for (int i = 0; i < 6; ++i) at_touch(bn_[i]);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i / 2 >= c && i / 2 < c + at_tiles_n(c, CH)) at_touch(xr[i]);
            // pin the terms split in this chunk to this chunk: their first use is a slab away, and hipcc otherwise
            // sinks the whole split there -- 180 vector instructions in a row in front of that slab's first MFMA
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r >= c - 1 && r < c - 1 + at_tiles_n(c - 1, CH)) { at_touch(xnx[r].t[0]); at_touch(xnx[r].t[1]); at_touch(xnx[r].t[2]); }
        }
        // A(t+2) must be in LDS on every wave before anyone reads it in slab t+1: 6 CH instructions follow it
        at_barrier<(6 * CH < 18 ? 6 * CH : 18)>();
    };
    for (int t = 0; t < T; t += 2) {
        slab(t, xa, xn);
        if (t + 1 < T) slab(t + 1, xn, xa);
    }
    at_barrier<0>();                                                  // the wrapped issues have landed: the ring is free
    MOC_STAMP(32);

    if constexpr (BWD) {
        // acc[c][s][r][i]: row r * 16 + (lane >> 4) * 4 + i, column (wave * CH + c) * 16 + (lane & 15)
        float* ds_s = reinterpret_cast<float*>(smem);                 // [K][AT_ROWS], zero beyond the bag
        for (int e = threadIdx.x; e < K * AT_ROWS; e += 256) {
            const int k = e / AT_ROWS, r = e % AT_ROWS;
            ds_s[e] = row0 + r < a.N ? a.ds[(int64_t)k * a.N + row0 + r] : 0.f;
        }
        __syncthreads();
        const int col = lane & 15, D = a.D, S = attn_dab_stride(D, K);
        float* cp = a.colpart + (int64_t)blockIdx.x * (2 + K) * D;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int d = (wave * CH + c) * 16 + col;
            float dg[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) dg[r][i] = 0.f;
            for (int k = 0; k < K; ++k) {                             // heads in order
                const float w = a.Wc[(int64_t)k * D + d];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 v = *reinterpret_cast<const float4*>(ds_s + k * AT_ROWS + r * 16 + kq * 4);
                    dg[r][0] = fmaf(v.x, w, dg[r][0]); dg[r][1] = fmaf(v.y, w, dg[r][1]);
                    dg[r][2] = fmaf(v.z, w, dg[r][2]); dg[r][3] = fmaf(v.w, w, dg[r][3]);
                }
            }
            float sa = 0.f, sb = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // tanh' = 4 e / (1 + e)^2 and sigmoid' = f / (1 + f)^2 in forms without cancellation
                    const float e = __expf(2.f * fminf(fmaxf(acc[c][0][r][i], -15.f), 15.f));
                    const float f = __expf(-fminf(fmaxf(acc[c][1][r][i], -30.f), 30.f));
                    const float ie = __frcp_rn(1.f + e), bv = __frcp_rn(1.f + f);
                    const float av = (e - 1.f) * ie, g = av * bv;
                    const float da = dg[r][i] * bv * (4.f * e * ie * ie), db = dg[r][i] * g * (f * bv);
                    acc[c][0][r][i] = g;                              // kept for d_Wc
                    sa += da; sb += db;
                    const int64_t row = row0 + r * 16 + kq * 4 + i;
                    if (row < a.N) {
                        a.dab[row * S + d] = da;
                        a.dab[row * S + D + d] = db;
                    }
                }
            sa += __shfl_xor(sa, 16, 64); sa += __shfl_xor(sa, 32, 64);
            sb += __shfl_xor(sb, 16, 64); sb += __shfl_xor(sb, 32, 64);
            if (kq == 0) { cp[d] = sa; cp[D + d] = sb; }
            for (int k = 0; k < K; ++k) {
                float sw = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 v = *reinterpret_cast<const float4*>(ds_s + k * AT_ROWS + r * 16 + kq * 4);
                    sw = fmaf(v.x, acc[c][0][r][0], sw); sw = fmaf(v.y, acc[c][0][r][1], sw);
                    sw = fmaf(v.z, acc[c][0][r][2], sw); sw = fmaf(v.w, acc[c][0][r][3], sw);
                }
                sw += __shfl_xor(sw, 16, 64); sw += __shfl_xor(sw, 32, 64);
                if (kq == 0) cp[(int64_t)(2 + k) * D + d] = sw;
            }
        }
    } else {

    // ---- gate and scores.  acc[c][s][r][i]: row r * 16 + (lane >> 4) * 4 + i, column (wave * CH + c) * 16 + (lane & 15)
    float* part_s = reinterpret_cast<float*>(smem);                  // [4 waves][K][AT_ROWS]
    float* score_s = part_s + (size_t)4 * K * AT_ROWS;                // [K][AT_ROWS]
    float* prob_s = score_s + (size_t)K * AT_ROWS;                    // [AT_ROWS]
    const int col = lane & 15;
    float wc[CH], wc_next[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) wc[c] = a.Wc[(wave * CH + c) * 16 + col];          // head 0, in flight behind the gate
#pragma unroll
    for (int c = 0; c < CH; ++c)                                      // the gate replaces the a-accumulators
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[c][0][r][i] = gate_fn(acc[c][0][r][i], acc[c][1][r][i]);
    for (int k = 0; k < K; ++k) {
        if (k + 1 < K) {
#pragma unroll
            for (int c = 0; c < CH; ++c) wc_next[c] = a.Wc[(int64_t)(k + 1) * a.D + (wave * CH + c) * 16 + col];
        }
        float part[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) part[r][i] = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) part[r][i] = fmaf(wc[c], acc[c][0][r][i], part[r][i]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) part[r][i] += __shfl_xor(part[r][i], off, 64);
            }
        if (col == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) part_s[((size_t)wave * K + k) * AT_ROWS + r * 16 + kq * 4 + i] = part[r][i];
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) wc[c] = wc_next[c];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < K * AT_ROWS; e += 256) {            // the four waves' column groups, in wave order
        const int k = e / AT_ROWS, r = e % AT_ROWS;
        const float s = (((part_s[((size_t)0 * K + k) * AT_ROWS + r] + part_s[((size_t)1 * K + k) * AT_ROWS + r]) +
                          part_s[((size_t)2 * K + k) * AT_ROWS + r]) + part_s[((size_t)3 * K + k) * AT_ROWS + r]) + a.bc[k];
        const bool valid = row0 + r < a.N;
        score_s[k * AT_ROWS + r] = valid ? s : -INFINITY;
        if (valid) a.A_raw[(int64_t)k * a.N + row0 + r] = s;
    }
    __syncthreads();

    MOC_STAMP(33);
    // ---- this workgroup's share of the softmax-weighted sum, per head: (m, l, M') in the online-softmax form
    const int nrow = a.N - row0 < AT_ROWS ? (int)(a.N - row0) : AT_ROWS;
    const int TL = L / 4;                                             // float4 columns of a row
    const int RS = TL >= 256 ? 1 : 256 / TL;                          // row slices that work side by side (RS * TL <= 256)
    const int slice = threadIdx.x / TL, cg = threadIdx.x % TL;
    float4* red_s = reinterpret_cast<float4*>(prob_s + AT_ROWS);      // [RS][TL] when RS > 1
    for (int k = 0; k < K; ++k) {
        if (wave == 0) {                                              // one lane per row
            const float sc = score_s[k * AT_ROWS + lane];             // -inf beyond the bag
            float m = sc;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
            const float pr = lane < nrow ? expf(sc - m) : 0.f;
            float l = pr;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) l += __shfl_xor(l, off, 64);
            prob_s[lane] = pr;
            if (lane == 0) {
                a.ws_m[(int64_t)blockIdx.x * K + k] = m;
                a.ws_l[(int64_t)blockIdx.x * K + k] = l;
            }
        }
        __syncthreads();
        float* out = a.ws_M + ((int64_t)blockIdx.x * K + k) * L;
        if (RS == 1) {
            for (int c = threadIdx.x * 4; c < L; c += 1024) {          // float4 columns, 8 rows in flight
                float4 sum = {0.f, 0.f, 0.f, 0.f};
                int r = 0;
                for (; r + 8 <= nrow; r += 8) {
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(a.h + (row0 + r + u) * L + c);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float p = prob_s[r + u];
                        sum.x = fmaf(p, v[u].x, sum.x); sum.y = fmaf(p, v[u].y, sum.y);
                        sum.z = fmaf(p, v[u].z, sum.z); sum.w = fmaf(p, v[u].w, sum.w);
                    }
                }
                for (; r < nrow; ++r) {
                    const float4 v = *reinterpret_cast<const float4*>(a.h + (row0 + r) * L + c);
                    const float p = prob_s[r];
                    sum.x = fmaf(p, v.x, sum.x); sum.y = fmaf(p, v.y, sum.y); sum.z = fmaf(p, v.z, sum.z); sum.w = fmaf(p, v.w, sum.w);
                }
                *reinterpret_cast<float4*>(out + c) = sum;
            }
        } else {
            // slice s takes rows s, s + RS, ...: all 256 threads load, 16 rows in flight each
            if (slice < RS) {
                float4 sum = {0.f, 0.f, 0.f, 0.f};
                const float* hc = a.h + row0 * L + cg * 4;
                int r = slice;
                for (; r + 15 * RS < nrow; r += 16 * RS) {
                    float4 v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const float4*>(hc + (int64_t)(r + u * RS) * L);
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const float p = prob_s[r + u * RS];
                        sum.x = fmaf(p, v[u].x, sum.x); sum.y = fmaf(p, v[u].y, sum.y);
                        sum.z = fmaf(p, v[u].z, sum.z); sum.w = fmaf(p, v[u].w, sum.w);
                    }
                }
                for (; r < nrow; r += RS) {
                    const float4 v = *reinterpret_cast<const float4*>(hc + (int64_t)r * L);
                    const float p = prob_s[r];
                    sum.x = fmaf(p, v.x, sum.x); sum.y = fmaf(p, v.y, sum.y); sum.z = fmaf(p, v.z, sum.z); sum.w = fmaf(p, v.w, sum.w);
                }
                red_s[slice * TL + cg] = sum;
            }
            __syncthreads();
            if (threadIdx.x < TL) {                                   // the slices, in slice order
                float4 sum = red_s[threadIdx.x];
                for (int q = 1; q < RS; ++q) {
                    const float4 v = red_s[q * TL + threadIdx.x];
                    sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
                }
                *reinterpret_cast<float4*>(out + threadIdx.x * 4) = sum;
            }
        }
        __syncthreads();
    }
    MOC_STAMP(34);
    }
}

// grid (K, ceil(L / 16)): merge the G per-workgroup triples of head k for 16 columns.  Thread = (column, one
// of sixteen interleaved slices of the workgroups), every load of a thread in flight at once (four at a time);
// the sixteen partial sums meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void attention_merge_kernel(const float* ws_m, const float* ws_l, const float* ws_M,
                                                              int G, int K, int L, float* M) {
    extern __shared__ float scale_s[];                               // [G] exp(m_g - m)
    __shared__ float red[256];
    const int k = blockIdx.x, c = blockIdx.y * 16 + (threadIdx.x & 15), slice = threadIdx.x >> 4;
    float m = -INFINITY;
    for (int g = threadIdx.x; g < G; g += 256) m = fmaxf(m, ws_m[(int64_t)g * K + k]);
    red[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
        __syncthreads();
    }
    m = red[0];
    __syncthreads();
    float l = 0.f;
    for (int g = threadIdx.x; g < G; g += 256) {
        const float sc = expf(ws_m[(int64_t)g * K + k] - m);
        scale_s[g] = sc;
        l += sc * ws_l[(int64_t)g * K + k];
    }
    red[threadIdx.x] = l;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    l = red[0];
    __syncthreads();
    float s = 0.f;
    if (c < L) {
        int g = slice;
        for (; g + 48 < G; g += 64) {                                // four independent loads in flight
            const float v0 = ws_M[((int64_t)g * K + k) * L + c], v1 = ws_M[((int64_t)(g + 16) * K + k) * L + c];
            const float v2 = ws_M[((int64_t)(g + 32) * K + k) * L + c], v3 = ws_M[((int64_t)(g + 48) * K + k) * L + c];
            s = fmaf(scale_s[g], v0, s); s = fmaf(scale_s[g + 16], v1, s);
            s = fmaf(scale_s[g + 32], v2, s); s = fmaf(scale_s[g + 48], v3, s);
        }
        for (; g < G; g += 16) s = fmaf(scale_s[g], ws_M[((int64_t)g * K + k) * L + c], s);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (slice == 0 && c < L) {
        float t = red[threadIdx.x];
#pragma unroll
        for (int q = 1; q < 16; ++q) t += red[threadIdx.x + 16 * q];
        M[(int64_t)k * L + c] = t / l;
    }
}

// ---- backward helpers -----------------------------------------------------------------------------------------

// grid G (64 rows per workgroup, as the main kernel).  dp[k][n] = h[n] . gM[k] -- the gradient arriving at the softmax
// weight p[k][n] through M = p h -- wave w on rows 16 w .. 16 w + 15, four rows in flight, lanes along the row (float4),
// butterfly sums in a fixed order.  Then the workgroup's share of the softmax statistics of every head, online form:
// part[g][k] = (m, l = sum e, c = sum e dp, sum gA) with e = exp(A_raw - m) over its rows.
__global__ __launch_bounds__(256) void attn_bwd_rows_kernel(const float* h, const float* gM, const float* A_raw, const float* gA,
                                                            int64_t N, int L, int K, float* dp, float4* part) {
    __shared__ float dp_s[AT_ROWS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * AT_ROWS;
    const int nrow = N - row0 < AT_ROWS ? (int)(N - row0) : AT_ROWS;
    const int L4 = L / 4;
    for (int k = 0; k < K; ++k) {
        if (gM) {
            const float4* gr = reinterpret_cast<const float4*>(gM + (int64_t)k * L);
            for (int r = wave * 16; r < wave * 16 + 16; r += 4) {
                float s[4] = {0.f, 0.f, 0.f, 0.f};
                for (int c = lane; c < L4; c += 64) {
                    const float4 g = gr[c];
                    float4 x[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int rr = r + u < nrow ? r + u : nrow - 1;                  // clamp: loads stay in the bag
                        x[u] = reinterpret_cast<const float4*>(h + (row0 + rr) * L)[c];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        s[u] = fmaf(x[u].x, g.x, s[u]); s[u] = fmaf(x[u].y, g.y, s[u]);
                        s[u] = fmaf(x[u].z, g.z, s[u]); s[u] = fmaf(x[u].w, g.w, s[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) s[u] += __shfl_xor(s[u], off, 64);
                }
                if (lane < 4) {
                    const float v = lane == 0 ? s[0] : lane == 1 ? s[1] : lane == 2 ? s[2] : s[3];
                    dp_s[r + lane] = v;
                    if (r + lane < nrow) dp[(int64_t)k * N + row0 + r + lane] = v;
                }
            }
        }
        __syncthreads();
        if (wave == 0) {                                              // one lane per row
            const bool valid = lane < nrow;
            const float sc = valid ? A_raw[(int64_t)k * N + row0 + lane] : -INFINITY;
            float m = sc;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
            const float e = valid ? expf(sc - m) : 0.f;
            float l = e, c = gM && valid ? e * dp_s[lane] : 0.f, sg = gA && valid ? gA[(int64_t)k * N + row0 + lane] : 0.f;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                l += __shfl_xor(l, off, 64); c += __shfl_xor(c, off, 64); sg += __shfl_xor(sg, off, 64);
            }
            if (lane == 0) part[(int64_t)blockIdx.x * K + k] = float4{m, l, c, sg};
        }
        __syncthreads();
    }
}

// grid (ceil(N / 256), K).  Every workgroup merges the G parts of its head in the same fixed order (thread t takes
// parts t, t + 256, ...; tree over the threads), then, for its 256 rows:  p = softmax_n(A_raw[k]),
// ds = p (dp - sum_n p dp) + gA  -- the gradient at A_raw from M through the softmax (dp null: none) plus the one
// arriving at A_raw itself (gA null: none).  p goes to column 2 D + k of dab (the columns up to the stride are zeroed: the
// caller's GEMMs run over the whole stride).  d_bc[k] = sum_n gA[k][n]: the softmax term sums to zero over n.
__global__ __launch_bounds__(256) void attn_bwd_ds_kernel(const float* A_raw, const float* dp, const float* gA, const float4* part,
                                                          int64_t N, int G, int D, int K, float* dab, float* ds, float* dbc) {
    __shared__ float red[256];
    const int k = blockIdx.y, S = attn_dab_stride(D, K);
    auto reduce = [&](float v, bool is_max) {
        red[threadIdx.x] = v;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) red[threadIdx.x] = is_max ? fmaxf(red[threadIdx.x], red[threadIdx.x + st]) : red[threadIdx.x] + red[threadIdx.x + st];
            __syncthreads();
        }
        const float r = red[0];
        __syncthreads();
        return r;
    };
    float m = -INFINITY;
    for (int g = threadIdx.x; g < G; g += 256) m = fmaxf(m, part[(int64_t)g * K + k].x);
    m = reduce(m, true);
    float l = 0.f, c = 0.f, sg = 0.f;
    for (int g = threadIdx.x; g < G; g += 256) {
        const float4 q = part[(int64_t)g * K + k];
        const float sc = expf(q.x - m);
        l = fmaf(sc, q.y, l); c = fmaf(sc, q.z, c); sg += q.w;
    }
    l = reduce(l, false);
    c = reduce(c, false) / l;
    sg = reduce(sg, false);
    if (blockIdx.x == 0 && threadIdx.x == 0) dbc[k] = sg;
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float pr = expf(A_raw[(int64_t)k * N + n] - m) / l;
    ds[(int64_t)k * N + n] = (dp ? pr * (dp[(int64_t)k * N + n] - c) : 0.f) + (gA ? gA[(int64_t)k * N + n] : 0.f);
    dab[n * S + 2 * D + k] = pr;
    if (k == 0)
        for (int j = 2 * D + K; j < S; ++j) dab[n * S + j] = 0.f;
}

// column sums of the per-workgroup parts, workgroups in a fixed order: out[j] = sum_g part[g][j].  Thread = (column,
// one of sixteen interleaved slices of the workgroups); the sixteen partial sums meet in LDS.
__global__ __launch_bounds__(256) void attn_bwd_colsum_kernel(const float* part, int G, int W, float* out) {
    __shared__ float red[256];
    const int j = blockIdx.x * 16 + (threadIdx.x & 15), slice = threadIdx.x >> 4;
    float s = 0.f;
    if (j < W) {
        int g = slice;
        for (; g + 48 < G; g += 64) {
            const float v0 = part[(int64_t)g * W + j], v1 = part[(int64_t)(g + 16) * W + j];
            const float v2 = part[(int64_t)(g + 32) * W + j], v3 = part[(int64_t)(g + 48) * W + j];
            s = (((s + v0) + v1) + v2) + v3;
        }
        for (; g < G; g += 16) s += part[(int64_t)g * W + j];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (slice == 0 && j < W) {
        float t = red[threadIdx.x];
#pragma unroll
        for (int q = 1; q < 16; ++q) t += red[threadIdx.x + 16 * q];
        out[j] = t;
    }
}

size_t attn_image_floats(int L, int D) { return (size_t)((L + 31) / 32) * D * 96; }      // 6 bytes per (padded) weight x 2 projections

size_t attn_ws_floats(int64_t N, int L, int D, int K) {
    const int64_t G = (N + AT_ROWS - 1) / AT_ROWS;
    return attn_image_floats(L, D) + (size_t)G * K * 2 + (size_t)G * K * L;
}

size_t attn_bwd_ws_floats(int64_t N, int L, int D, int K) {
    const int64_t G = (N + AT_ROWS - 1) / AT_ROWS;
    return attn_image_floats(L, D) + (size_t)G * (2 + K) * D + (size_t)K * N + (size_t)G * K * 4 + 4;
}

template <bool BWD>
int attn_launch(const std::conditional_t<BWD, AttnBwdArgs, AttnArgs>& a, int G, hipStream_t s, const char* what) {
#define MOC_LAUNCH_ATTN(NDV)                                                                                         \
    do {                                                                                                             \
        static bool attr_set = false;                                                                                \
        if (!attr_set) {                                                                                             \
            const hipError_t ea = hipFuncSetAttribute((const void*)gated_attention_kernel<NDV, BWD>,                 \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, AT_SMEM);          \
            if (ea != hipSuccess) MOC_FAIL(MOC_ELAUNCH, "%s: %s", what, hipGetErrorString(ea));                      \
            attr_set = true;                                                                                         \
        }                                                                                                            \
        gated_attention_kernel<NDV, BWD><<<G, 256, AT_SMEM, s>>>(a);                                                 \
    } while (0)
    if (a.D == 128) MOC_LAUNCH_ATTN(8);
    else if (a.D == 256) MOC_LAUNCH_ATTN(16);
    else MOC_LAUNCH_ATTN(24);
#undef MOC_LAUNCH_ATTN
    MOC_CHECK_LAUNCH(what);
    return MOC_OK;
}

}  // namespace

extern "C" size_t moc_gated_attention_workspace(int64_t N, int L, int D, int K) {
    if (N < 1 || L < 16 || D < 16 || K < 1) return 0;
    return attn_ws_floats(N, L, D, K) * sizeof(float);
}

extern "C" int moc_gated_attention_pool(const float* h, int64_t N, int L, const float* Wa, const float* ba,
                                        const float* Wb, const float* bb, int D, const float* Wc, const float* bc,
                                        int K, float* A_raw, float* M, void* workspace, size_t workspace_bytes,
                                        moc_stream_t stream) {
    MOC_REQUIRE(h && Wa && ba && Wb && bb && Wc && bc && A_raw && M && workspace, "moc_gated_attention_pool: null pointer");
    MOC_REQUIRE(N >= 1 && N < (1ll << 31), "moc_gated_attention_pool: bad N=%lld", (long long)N);
    MOC_REQUIRE(L >= 16 && L % 16 == 0 && L <= 4096, "moc_gated_attention_pool: L=%d must be a multiple of 16 (<= 4096)", L);
    MOC_REQUIRE(D == 128 || D == 256 || D == 384, "moc_gated_attention_pool: D=%d not in {128, 256, 384}", D);
    MOC_REQUIRE(K >= 1 && K <= 64, "moc_gated_attention_pool: K=%d outside [1, 64]", K);
    MOC_REQUIRE(((uintptr_t)h & 15) == 0 && ((uintptr_t)Wa & 15) == 0 && ((uintptr_t)Wb & 15) == 0 && ((uintptr_t)workspace & 15) == 0,
                "moc_gated_attention_pool: h, Wa, Wb and the workspace must be 16-byte aligned");
    MOC_REQUIRE(workspace_bytes >= attn_ws_floats(N, L, D, K) * sizeof(float),
                "moc_gated_attention_pool: workspace of %zu bytes, need %zu", workspace_bytes, attn_ws_floats(N, L, D, K) * sizeof(float));
    hipStream_t s = (hipStream_t)stream;
    const int G = (int)((N + AT_ROWS - 1) / AT_ROWS);
    float* img = (float*)workspace;
    AttnArgs a;
    a.h = h; a.img = (const unsigned char*)img; a.ba = ba; a.bb = bb; a.Wc = Wc; a.bc = bc; a.A_raw = A_raw;
    a.ws_m = img + attn_image_floats(L, D);
    a.ws_l = a.ws_m + (size_t)G * K;
    a.ws_M = a.ws_l + (size_t)G * K;
    a.N = N; a.L = L; a.D = D; a.K = K;
    const int64_t nfrag = (int64_t)((L + 31) / 32) * 4 * (D / 64) * 2 * 64;
    attn_image_kernel<<<moc_cdiv(nfrag, 256), 256, 0, s>>>(Wa, Wb, L, D, (uint4*)img);
    MOC_CHECK_LAUNCH("moc_gated_attention_pool(image)");
    static_assert((size_t)(5 * 64 + 1) * AT_ROWS * sizeof(float) + 4096 <= AT_SMEM, "the epilogue arrays (K <= 64) reuse the ring");
    { const int rc = attn_launch<false>(a, G, s, "moc_gated_attention_pool"); if (rc != MOC_OK) return rc; }
    attention_merge_kernel<<<dim3(K, moc_cdiv(L, 16)), 256, (size_t)G * sizeof(float), s>>>(a.ws_m, a.ws_l, a.ws_M, G, K, L, M);
    MOC_CHECK_LAUNCH("moc_gated_attention_pool(merge)");
    return MOC_OK;
}

extern "C" size_t moc_gated_attention_backward_workspace(int64_t N, int L, int D, int K) {
    if (N < 1 || L < 16 || D < 16 || K < 1) return 0;
    return attn_bwd_ws_floats(N, L, D, K) * sizeof(float);
}

extern "C" int moc_gated_attention_dab_stride(int D, int K) { return attn_dab_stride(D, K); }

extern "C" int moc_gated_attention_backward(const float* h, int64_t N, int L, const float* Wa, const float* ba,
                                            const float* Wb, const float* bb, int D, const float* Wc, int K,
                                            const float* A_raw, const float* gA, const float* gM, float* dab,
                                            float* ds, float* dcol, float* dbc, void* workspace, size_t workspace_bytes,
                                            moc_stream_t stream) {
    MOC_REQUIRE(h && Wa && ba && Wb && bb && Wc && A_raw && dab && ds && dcol && dbc && workspace,
                "moc_gated_attention_backward: null pointer");
    MOC_REQUIRE(N >= 1 && N < (1ll << 31), "moc_gated_attention_backward: bad N=%lld", (long long)N);
    MOC_REQUIRE(L >= 16 && L % 16 == 0 && L <= 4096, "moc_gated_attention_backward: L=%d must be a multiple of 16 (<= 4096)", L);
    MOC_REQUIRE(D == 128 || D == 256 || D == 384, "moc_gated_attention_backward: D=%d not in {128, 256, 384}", D);
    MOC_REQUIRE(K >= 1 && K <= 64, "moc_gated_attention_backward: K=%d outside [1, 64]", K);
    MOC_REQUIRE(((uintptr_t)h & 15) == 0 && ((uintptr_t)Wa & 15) == 0 && ((uintptr_t)Wb & 15) == 0 &&
                ((uintptr_t)workspace & 15) == 0 && (!gM || ((uintptr_t)gM & 15) == 0),
                "moc_gated_attention_backward: h, Wa, Wb, gM and the workspace must be 16-byte aligned");
    MOC_REQUIRE(workspace_bytes >= attn_bwd_ws_floats(N, L, D, K) * sizeof(float),
                "moc_gated_attention_backward: workspace of %zu bytes, need %zu", workspace_bytes, attn_bwd_ws_floats(N, L, D, K) * sizeof(float));
    hipStream_t s = (hipStream_t)stream;
    const int G = (int)((N + AT_ROWS - 1) / AT_ROWS);
    float* img = (float*)workspace;
    float* colpart = img + attn_image_floats(L, D);
    float* dp = colpart + (size_t)G * (2 + K) * D;
    float4* part = (float4*)(((uintptr_t)(dp + (size_t)K * N) + 15) & ~(uintptr_t)15);
    const int64_t nfrag = (int64_t)((L + 31) / 32) * 4 * (D / 64) * 2 * 64;
    attn_image_kernel<<<moc_cdiv(nfrag, 256), 256, 0, s>>>(Wa, Wb, L, D, (uint4*)img);
    MOC_CHECK_LAUNCH("moc_gated_attention_backward(image)");
    attn_bwd_rows_kernel<<<G, 256, 0, s>>>(h, gM, A_raw, gA, N, L, K, dp, part);
    MOC_CHECK_LAUNCH("moc_gated_attention_backward(rows)");
    attn_bwd_ds_kernel<<<dim3((unsigned)moc_cdiv(N, 256), K), 256, 0, s>>>(A_raw, gM ? dp : nullptr, gA, part, N, G, D, K, dab, ds, dbc);
    MOC_CHECK_LAUNCH("moc_gated_attention_backward(ds)");
    AttnBwdArgs a{};
    a.h = h; a.img = (const unsigned char*)img; a.ba = ba; a.bb = bb; a.Wc = Wc;
    a.ds = ds; a.dab = dab; a.colpart = colpart;
    a.N = N; a.L = L; a.D = D; a.K = K;
    { const int rc = attn_launch<true>(a, G, s, "moc_gated_attention_backward"); if (rc != MOC_OK) return rc; }
    const int W = (2 + K) * D;
    attn_bwd_colsum_kernel<<<moc_cdiv(W, 16), 256, 0, s>>>(colpart, G, W, dcol);
    MOC_CHECK_LAUNCH("moc_gated_attention_backward(colsum)");
    return MOC_OK;
}
