// Gated-attention MIL pooling (reference models/model_clam.py:41-64 Attn_Net_Gated,
// :178-183 / :206 CLAM_SB.forward_single, :291-296 / :318 CLAM_MB.forward):
//
//   a = tanh(h Wa^T + ba),  b = sigmoid(h Wb^T + bb)                 [N, D]
//   A_raw[k][n] = Wc[k] . (a[n] * b[n]) + bc[k]                      [K, N]   (K = 1: CLAM_SB, n_classes: CLAM_MB)
//   M[k]        = sum_n softmax_n(A_raw[k])[n] * h[n]                [K, L]
//
// One pass over the bag.  A workgroup owns 64 rows (a wave 16): both projections are accumulated side by
// side on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products, 2 * D/16 accumulator tiles
// per wave), the weights streamed through LDS in MFMA operand order (16 columns of L per stage, double
// buffered, every wave of the workgroup reads the same stage); the gate, the K dot products with Wc and
// their 16-lane reductions happen in registers; then the workgroup forms its share of the softmax in the
// online form -- m = max score, l = sum exp(score - m), M' = sum exp(score - m) * h -- re-reading its 64
// rows (still in L2).  A second, tiny launch merges the per-workgroup (m, l, M') triples.  Neither the
// [N, D] activations nor the softmax weights ever reach memory.
//
// Bound: the fp32 matrix pipe.  4 * N * L * D flops at 256 flop/clk/CU (157 TFLOP/s): N = 15,000, L = 512,
// D = 384 -> 11.8 GFLOP -> 75 us; the bag itself is 30.7 MB (6 us of HBM time), the weight stages come
// from L2 (48 KiB per stage per workgroup).
#include "moc_common.h"

namespace {

constexpr int AT_ROWS = 64;          // rows per workgroup

struct AttnArgs {
    const float* h;                  // [N, L]
    const float* img;                // weight image (attn_image_kernel)
    const float *ba, *bb, *Wc, *bc;  // [D] [D] [K, D] [K]
    float* A_raw;                    // [K, N]
    float *ws_m, *ws_l, *ws_M;       // [G, K] [G, K] [G, K, L]
    int64_t N;
    int L, D, K;
};

// image: [t = L/16][nt2 = 2 * D/16][lane][4] fp32; nt2 < D/16: Wa, else Wb;
// element m of lane l = W[nt*16 + (l & 15)][t*16 + (l >> 4)*4 + m]   (the B operand of four k = 4 MFMAs)
__global__ __launch_bounds__(256) void attn_image_kernel(const float* Wa, const float* Wb, int L, int D, float* img) {
    const int ND = D / 16, T = L / 16;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one float4 each
    if (idx >= (int64_t)T * 2 * ND * 64) return;
    const int lane = (int)(idx & 63), nt2 = (int)((idx >> 6) % (2 * ND)), t = (int)((idx >> 6) / (2 * ND));
    const float* W = nt2 < ND ? Wa : Wb;
    const int n = (nt2 < ND ? nt2 : nt2 - ND) * 16 + (lane & 15);
    const float4 v = *reinterpret_cast<const float4*>(W + (int64_t)n * L + t * 16 + (lane >> 4) * 4);
    reinterpret_cast<float4*>(img)[idx] = v;
}

// B fragments from LDS, issued by hand one pair of n-tiles ahead of the MFMAs that use them and awaited with a
// counted lgkmcnt (LDS returns in order; the LDS-DMA of the next stage counts on vmcnt, not here).  Left to
// hipcc the loop is "2 reads, wait, 8 MFMAs": the read latency is exposed once per 256 cycles of matrix work.
typedef unsigned __attribute__((ext_vector_type(4))) au32x4_t;
template <int OFF>
__device__ __forceinline__ void at_lds16(au32x4_t& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int KEEP>
__device__ __forceinline__ void at_wait(au32x4_t& x, au32x4_t& y) {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(KEEP) : "memory");
    asm volatile("" : "+v"(x));
    asm volatile("" : "+v"(y));
}
template <int Q, int NQ>
__device__ __forceinline__ void at_pairs(f32x4_t (&acc)[NQ], const float4& xa, unsigned base, au32x4_t& b0, au32x4_t& b1,
                                         au32x4_t& n0, au32x4_t& n1) {
    if constexpr (Q < NQ) {
        if constexpr (Q + 2 < NQ) {
            at_lds16<(Q + 2) * 1024>(n0, base);
            at_lds16<(Q + 3) * 1024>(n1, base);
            at_wait<2>(b0, b1);
        } else {
            at_wait<0>(b0, b1);
        }
        const float xs[4] = {xa.x, xa.y, xa.z, xa.w};
#pragma unroll
        for (int m = 0; m < 4; ++m) {                                 // two independent accumulators alternate
            acc[Q] = __builtin_amdgcn_mfma_f32_16x16x4f32(xs[m], __uint_as_float(b0[m]), acc[Q], 0, 0, 0);
            acc[Q + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xs[m], __uint_as_float(b1[m]), acc[Q + 1], 0, 0, 0);
        }
        at_pairs<Q + 2, NQ>(acc, xa, base, n0, n1, b0, b1);
    }
}

// gate nonlinearities on the hardware exp / rcp (v_exp_f32, v_rcp_f32: ~1e-7 absolute on outputs in [-1, 1])
__device__ __forceinline__ float fast_sigmoid(float x) { return __frcp_rn(1.f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * x)); }

template <int ND>
__global__ __launch_bounds__(256, 1) void gated_attention_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int STAGE_VEC = 2 * ND * 64;                            // float4 per stage (16 columns of L)
    float4* stage = reinterpret_cast<float4*>(smem);                 // [2][STAGE_VEC weights + 256 A fragments]
    float* score_s = reinterpret_cast<float*>(smem + 2 * (STAGE_VEC + 256) * 16);   // [K][AT_ROWS]
    float* prob_s = score_s + (size_t)a.K * AT_ROWS;                        // [AT_ROWS]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int L = a.L, K = a.K, T = L / 16;
    const int64_t row0 = (int64_t)blockIdx.x * AT_ROWS;
    const int64_t my_row = row0 + wave * 16 + (lane & 15);
    const int64_t ld_row = my_row < a.N ? my_row : a.N - 1;           // clamp: loads stay in bounds
    const float* hp = a.h + ld_row * L + (lane >> 4) * 4;
    const float4* img = reinterpret_cast<const float4*>(a.img);

    f32x4_t acc[2 * ND];
#pragma unroll
    for (int q = 0; q < 2 * ND; ++q) acc[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // Weight stages AND the workgroup's A fragments go global -> LDS directly (global_load_lds_dwordx4: no
    // VGPR in between -- with 2 * ND accumulator tiles per wave there are none to spare; register staging
    // spilled to scratch and ran 3.5x slower; an A fragment prefetched into registers was spilled too and its
    // reload's vmcnt(0) drained the weight DMA early).  One instruction moves 64 lanes x 16 B: the weight
    // image is copied as it lies; lane (row, kq) fetches h[row][16t + 4kq ..+3] into slot [wave][lane].
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    constexpr int PER = STAGE_VEC / 256;                              // weight instructions per wave per stage
    constexpr int BUF_VEC = STAGE_VEC + 256;                          // + 4 waves x 64 A fragments
    auto issue_stage = [&](int t, int buf) {
        float4* dst = stage + buf * BUF_VEC;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int v = q * 256 + wave * 64;                        // first float4 of this wave's piece
            __builtin_amdgcn_global_load_lds((gptr_t)(img + (int64_t)t * STAGE_VEC + v + lane), (lptr_t)(dst + v), 16, 0, 0);
        }
        __builtin_amdgcn_global_load_lds((gptr_t)(hp + t * 16), (lptr_t)(dst + STAGE_VEC + wave * 64), 16, 0, 0);
    };
    issue_stage(0, 0);
    __syncthreads();                                                  // (its fence waits for the LDS-DMA: vmcnt(0))
    for (int t = 0; t < T; ++t) {
        const float4* cur = stage + (t & 1) * BUF_VEC;
        if (t + 1 < T) issue_stage(t + 1, (t + 1) & 1);               // in flight behind this stage's MFMAs
        {
            // every LDS read of the loop is hand-issued: an ordinary read would make hipcc wait vmcnt(0) first
            // (the DMA just issued writes LDS; it cannot tell the two buffers apart) and the overlap is gone
            const unsigned base = (unsigned)(uintptr_t)(cur + lane);
            au32x4_t xr, b0, b1, n0, n1;
            at_lds16<0>(xr, (unsigned)(uintptr_t)(cur + STAGE_VEC + wave * 64 + lane));
            at_lds16<0>(b0, base);
            at_lds16<1024>(b1, base);
            at_wait<2>(xr, xr);
            const float4 xa = {__uint_as_float(xr[0]), __uint_as_float(xr[1]), __uint_as_float(xr[2]), __uint_as_float(xr[3])};
            at_pairs<0, 2 * ND>(acc, xa, base, b0, b1, n0, n1);
        }
        __syncthreads();                                              // next stage landed and visible; this one free
    }

    // ---- gate and scores.  acc[q][i]: row (lane >> 4) * 4 + i of the wave's 16, column q*16 + (lane & 15)
    const int col = lane & 15;
#pragma unroll
    for (int q = 0; q < ND; ++q) {                                    // the gate replaces the a-accumulators
        const int d = q * 16 + col;
        const float wa = a.ba[d], wb = a.bb[d];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            acc[q][i] = fast_tanh(acc[q][i] + wa) * fast_sigmoid(acc[ND + q][i] + wb);
    }
    for (int k = 0; k < K; ++k) {
        float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < ND; ++q) {
            const float wc = a.Wc[(int64_t)k * a.D + q * 16 + col];
#pragma unroll
            for (int i = 0; i < 4; ++i) part[i] = fmaf(wc, acc[q][i], part[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) part[i] += __shfl_xor(part[i], off, 64);
        }
        if (col == 0) {
            const float bk = a.bc[k];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = wave * 16 + (lane >> 4) * 4 + i;
                const float s = part[i] + bk;
                const bool valid = row0 + r < a.N;
                score_s[k * AT_ROWS + r] = valid ? s : -INFINITY;
                if (valid) a.A_raw[(int64_t)k * a.N + row0 + r] = s;
            }
        }
    }
    __syncthreads();

    // ---- this workgroup's share of the softmax-weighted sum, per head: (m, l, M') in the online-softmax form
    const int nrow = a.N - row0 < AT_ROWS ? (int)(a.N - row0) : AT_ROWS;
    for (int k = 0; k < K; ++k) {
        float m = -INFINITY;
        for (int r = 0; r < nrow; ++r) m = fmaxf(m, score_s[k * AT_ROWS + r]);
        if (threadIdx.x < AT_ROWS) prob_s[threadIdx.x] = threadIdx.x < nrow ? expf(score_s[k * AT_ROWS + threadIdx.x] - m) : 0.f;
        __syncthreads();
        if (threadIdx.x == 0) {
            float l = 0.f;
            for (int r = 0; r < nrow; ++r) l += prob_s[r];
            a.ws_m[(int64_t)blockIdx.x * K + k] = m;
            a.ws_l[(int64_t)blockIdx.x * K + k] = l;
        }
        for (int c = threadIdx.x * 4; c < L; c += 1024) {              // float4 columns, 8 rows in flight
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
            *reinterpret_cast<float4*>(a.ws_M + ((int64_t)blockIdx.x * K + k) * L + c) = sum;
        }
        __syncthreads();
    }
}

// grid (K, ceil(L / 64)): merge the G per-workgroup triples of head k for 64 columns.  Thread = (column,
// one of four interleaved slices of the workgroups); the four partial sums meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void attention_merge_kernel(const float* ws_m, const float* ws_l, const float* ws_M,
                                                              int G, int K, int L, float* M) {
    extern __shared__ float scale_s[];                               // [G] exp(m_g - m), then [4][64] partials
    __shared__ float red[256];
    const int k = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
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
        for (; g + 12 < G; g += 16) {                                // four independent loads in flight
            const float v0 = ws_M[((int64_t)g * K + k) * L + c], v1 = ws_M[((int64_t)(g + 4) * K + k) * L + c];
            const float v2 = ws_M[((int64_t)(g + 8) * K + k) * L + c], v3 = ws_M[((int64_t)(g + 12) * K + k) * L + c];
            s = fmaf(scale_s[g], v0, s); s = fmaf(scale_s[g + 4], v1, s);
            s = fmaf(scale_s[g + 8], v2, s); s = fmaf(scale_s[g + 12], v3, s);
        }
        for (; g < G; g += 4) s = fmaf(scale_s[g], ws_M[((int64_t)g * K + k) * L + c], s);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (slice == 0 && c < L)
        M[(int64_t)k * L + c] = (((red[threadIdx.x] + red[threadIdx.x + 64]) + red[threadIdx.x + 128]) + red[threadIdx.x + 192]) / l;
}

size_t attn_ws_floats(int64_t N, int L, int D, int K) {
    const int64_t G = (N + AT_ROWS - 1) / AT_ROWS;
    return (size_t)2 * D * L + (size_t)G * K * 2 + (size_t)G * K * L;
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
    a.h = h; a.img = img; a.ba = ba; a.bb = bb; a.Wc = Wc; a.bc = bc; a.A_raw = A_raw;
    a.ws_m = img + (size_t)2 * D * L;
    a.ws_l = a.ws_m + (size_t)G * K;
    a.ws_M = a.ws_l + (size_t)G * K;
    a.N = N; a.L = L; a.D = D; a.K = K;
    const int64_t nvec = (int64_t)(L / 16) * 2 * (D / 16) * 64;
    attn_image_kernel<<<moc_cdiv(nvec, 256), 256, 0, s>>>(Wa, Wb, L, D, img);
    MOC_CHECK_LAUNCH("moc_gated_attention_pool(image)");
    const size_t smem = (size_t)2 * (2 * (D / 16) * 64 + 256) * 16 + (size_t)(K + 1) * AT_ROWS * sizeof(float);
#define MOC_LAUNCH_ATTN(NDV)                                                                                         \
    do {                                                                                                             \
        static bool attr_set = false;                                                                                \
        if (!attr_set) {                                                                                             \
            (void)hipFuncSetAttribute((const void*)gated_attention_kernel<NDV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_set = true;                                                                                         \
        }                                                                                                            \
        gated_attention_kernel<NDV><<<G, 256, smem, s>>>(a);                                                         \
    } while (0)
    if (D == 128) MOC_LAUNCH_ATTN(8);
    else if (D == 256) MOC_LAUNCH_ATTN(16);
    else MOC_LAUNCH_ATTN(24);
#undef MOC_LAUNCH_ATTN
    MOC_CHECK_LAUNCH("moc_gated_attention_pool");
    attention_merge_kernel<<<dim3(K, moc_cdiv(L, 64)), 256, (size_t)G * sizeof(float), s>>>(a.ws_m, a.ws_l, a.ws_M, G, K, L, M);
    MOC_CHECK_LAUNCH("moc_gated_attention_pool(merge)");
    return MOC_OK;
}
