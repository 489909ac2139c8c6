// Shared host/device helpers for libmoc_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/moc_hip.h"

// No implicit fused multiply-adds anywhere in this library: a * b + c is a multiplication and an addition, each rounded,
// as in the reference's PyTorch-CPU elementwise kernels (the gated mix main_moc.py:391-403, Adam's lerp / addcmul /
// addcdiv); fused operations exist only where the source says fmaf.  hipcc's default (-ffp-contract=fast) fuses or not
// per call site -- __fmul_rn / __fadd_rn are plain operators in this toolchain -- so two kernels holding the same
// expression could differ in the last bit (the 256-row evaluation forward did, against the 16-row one, in `mixed`).
#pragma clang fp contract(off)
// The single operations the kernels spell out where the reference's rounding sequence matters.  NOT hip's __fmul_rn /
// __fadd_rn: those are inline functions of a system header, compiled under ITS contraction mode (fast), so that after
// inlining `__fadd_rn(v, __fmul_rn(a, b))` may still become one fma -- kernel by kernel, as the optimiser sees fit.
static __device__ __forceinline__ float moc_fadd(float a, float b) { return a + b; }
static __device__ __forceinline__ float moc_fsub(float a, float b) { return a - b; }
static __device__ __forceinline__ float moc_fmul(float a, float b) { return a * b; }
static __device__ __forceinline__ float moc_fdiv(float a, float b) { return a / b; }
static __device__ __forceinline__ float moc_fsqrt(float a) { return __builtin_sqrtf(a); }

// streaming score pass, ticketed walk: tiles per ticket (moc_scores.hip)
#define MOC_TILES_PER_TICKET 2
#define MOC_WAVE 64
#define MOC_HIDDEN 64

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// ---- host-side error plumbing ------------------------------------------------
void moc_set_error(const char* fmt, ...);
#define MOC_FAIL(code, ...)          \
    do {                             \
        moc_set_error(__VA_ARGS__);  \
        return (code);               \
    } while (0)
#define MOC_REQUIRE(cond, ...)                        \
    do {                                              \
        if (!(cond)) MOC_FAIL(MOC_EINVAL, __VA_ARGS__); \
    } while (0)
#define MOC_CHECK_LAUNCH(name)                                                       \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess)                                                       \
            MOC_FAIL(MOC_ELAUNCH, "%s: launch failed: %s", name, hipGetErrorString(e__)); \
    } while (0)

static inline int moc_elem_size(int dtype) { return dtype == MOC_F32 ? 4 : 2; }
static inline int moc_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- device helpers ------------------------------------------------------------
// Order-preserving map float -> uint32 (larger float => larger key); -0 == +0.
__device__ __forceinline__ uint32_t moc_key_desc(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u << 1) == 0) u = 0;  // canonical zero
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// softmax[c] of a row as the score pass forms it (row_stats_emit: e * rden with e = exp2((v - m1) log2 e)): the same
// instructions, hence the same bits -- the compact statistics (MOC_STATS_COMPACT) store (v, m1, 1/den) instead of the C
// softmax columns, and every consumer re-forms the value with this function
__device__ __forceinline__ float moc_softmax_from(float v, float m1, float rden) {
    return __builtin_amdgcn_exp2f((v - m1) * 1.44269504088896340736f) * rden;
}

__device__ __forceinline__ float moc_bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// round-to-nearest-even fp32 -> bf16 bits (finite inputs)
__device__ __forceinline__ uint16_t moc_f32_to_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// ---- 16-bit bag storage (bf16 or fp16), fp32-exact products -------------------------------------
// A weight is held as three 16-bit terms hi + mid + lo so that x * w is exact in the fp32 accumulator:
//   bf16: the three terms of w itself (8+8+8 mantissa bits);
//   fp16: the three terms of w * 2^S (11+11+2 bits) -- fp16 has 5 exponent bits, so the small terms of
//         an unscaled weight would be subnormal; the accumulated product is scaled back by 2^-S (exact).
//         S = 14 for the classifier bank (|w| < 2 is checked), 10 for the meta-learner's W1 (|w| < 32).
#define MOC_F16_BANK_SCALE 16384.0f
#define MOC_F16_W1_SCALE 1024.0f

__device__ __forceinline__ uint16_t moc_f16_bits(_Float16 h) { return __builtin_bit_cast(uint16_t, h); }
__device__ __forceinline__ float moc_f16_to_f32(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }

template <bool F16>
__device__ __forceinline__ float moc_half_to_f32(uint16_t b) {
    if constexpr (F16) return moc_f16_to_f32(b);
    else return moc_bf16_to_f32(b);
}

// `scale` is applied for fp16 only
template <bool F16>
__device__ __forceinline__ void moc_split3(float w, float scale, uint16_t& hi, uint16_t& mid, uint16_t& lo) {
    if constexpr (F16) {
        const float ws = w * scale;
        const _Float16 h = (_Float16)ws;
        const float r1 = ws - (float)h;
        const _Float16 m = (_Float16)r1;
        const _Float16 l = (_Float16)(r1 - (float)m);
        hi = moc_f16_bits(h); mid = moc_f16_bits(m); lo = moc_f16_bits(l);
    } else {
        hi = moc_f32_to_bf16_rne(w);
        const float r1 = w - moc_bf16_to_f32(hi);
        mid = moc_f32_to_bf16_rne(r1);
        lo = moc_f32_to_bf16_rne(r1 - moc_bf16_to_f32(mid));
    }
}

template <bool F16, typename V>
__device__ __forceinline__ f32x4_t moc_mfma_half(const V& a, const V& b, f32x4_t c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// Exclusive prefix over a block of 0/1 flags.  `wave_tot` is LDS scratch with one
// int per wave (+1).  Returns this thread's offset; *block_total gets the sum.
// All threads of the block must call it; it contains two __syncthreads().
__device__ __forceinline__ int moc_block_flag_scan(bool flag, int* wave_tot, int* block_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = (blockDim.x + 63) >> 6;
    const unsigned long long bal = __ballot(flag);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(bal);
    __syncthreads();
    int off = 0, tot = 0;
    for (int w = 0; w < nwave; ++w) {
        const int t = wave_tot[w];
        if (w < wave) off += t;
        tot += t;
    }
    __syncthreads();
    *block_total = tot;
    return off + before;
}

// Exclusive prefix over a block of small counts (same contract as moc_block_flag_scan).
__device__ __forceinline__ int moc_block_count_scan(int cnt, int* wave_tot, int* block_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = (blockDim.x + 63) >> 6;
    int inc = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    int off = 0, tot = 0;
    for (int w = 0; w < nwave; ++w) {
        const int t = wave_tot[w];
        if (w < wave) off += t;
        tot += t;
    }
    __syncthreads();
    *block_total = tot;
    return off + inc - cnt;
}

// inclusive max-scan by DPP row shifts, then the wave total from lane 63 (gfx9 DPP controls:
// row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143)
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
#define MOC_DPP_STEP(ctrl, rmask)                                                                  \
    {                                                                                              \
        const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, ctrl, rmask, 0xf, false); \
        const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, ctrl, rmask, 0xf, false); \
        const bool gt = ohi > hi || (ohi == hi && olo > lo);                                       \
        lo = gt ? olo : lo;                                                                        \
        hi = gt ? ohi : hi;                                                                        \
    }
    MOC_DPP_STEP(0x111, 0xf) MOC_DPP_STEP(0x112, 0xf) MOC_DPP_STEP(0x114, 0xf) MOC_DPP_STEP(0x118, 0xf)
    MOC_DPP_STEP(0x142, 0xa) MOC_DPP_STEP(0x143, 0xc)
#undef MOC_DPP_STEP
    lo = (unsigned)__builtin_amdgcn_readlane((int)lo, 63);
    hi = (unsigned)__builtin_amdgcn_readlane((int)hi, 63);
    return ((unsigned long long)hi << 32) | lo;
}

// Every 64-byte line of the first BYTES bytes of the kernel's argument segment, requested side by side and waited for
// ONCE.  A latency-critical kernel with a few hundred bytes of arguments otherwise meets them one scalar-cache miss after
// the other (each s_load of a new line in front of its first use: round 4 counted five serialised misses in front of the
// step kernel's first vector load).  BYTES must not exceed the kernel's explicit arguments (nothing past them is read).
template <size_t BYTES>
__device__ __forceinline__ void moc_kernarg_touch() {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(BYTES <= 1024, "moc_kernarg_touch: sixteen lines at most");
    auto ka = __builtin_amdgcn_kernarg_segment_ptr();
    unsigned d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0, d6 = 0, d7 = 0, d8 = 0, d9 = 0, d10 = 0, d11 = 0, d12 = 0, d13 = 0, d14 = 0, d15 = 0;
#define MOC_KA_LINE(i, off) if constexpr (BYTES > off) asm volatile("s_load_dword %0, %1, " #off : "=s"(d##i) : "s"(ka));
    MOC_KA_LINE(0, 0x0) MOC_KA_LINE(1, 0x40) MOC_KA_LINE(2, 0x80) MOC_KA_LINE(3, 0xc0) MOC_KA_LINE(4, 0x100) MOC_KA_LINE(5, 0x140)
    MOC_KA_LINE(6, 0x180) MOC_KA_LINE(7, 0x1c0) MOC_KA_LINE(8, 0x200) MOC_KA_LINE(9, 0x240) MOC_KA_LINE(10, 0x280) MOC_KA_LINE(11, 0x2c0)
    MOC_KA_LINE(12, 0x300) MOC_KA_LINE(13, 0x340) MOC_KA_LINE(14, 0x380) MOC_KA_LINE(15, 0x3c0)
#undef MOC_KA_LINE
    // (the destinations stay allocated until everything has landed: they are inputs of the wait)
    asm volatile("s_waitcnt lgkmcnt(0)" :: "s"(d0), "s"(d1), "s"(d2), "s"(d3), "s"(d4), "s"(d5), "s"(d6), "s"(d7), "s"(d8), "s"(d9),
                 "s"(d10), "s"(d11), "s"(d12), "s"(d13), "s"(d14), "s"(d15) : "memory");
#endif
}

// ---- diagnostic build only (-DMOC_STAMPS, make stamps): constant-clock (100 MHz) time stamps of
// kernel phases, written by thread 0 of workgroup (0,0) to a global array that nothing else reads.
#ifdef MOC_STAMPS
extern __device__ unsigned long long g_moc_stamps[128];
#define MOC_STAMP(id)                                                                          \
    do {                                                                                       \
        if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) {                            \
            g_moc_stamps[id] = wall_clock64();                                                 \
            g_moc_stamps[64 + (id)] = __builtin_amdgcn_s_memtime();   /* shader cycles */       \
        }                                                                                      \
    } while (0)
// the same after everything the stamping wave has in flight (vector and scalar memory, LDS) has arrived: the stamp then
// marks an ARRIVAL, not an issue (perturbs the kernel: the diagnostic build's business)
#define MOC_STAMP_DRAIN(id)                                                                     \
    do {                                                                                       \
        if (blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x >> 6) == 0)                       \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                        \
        MOC_STAMP(id);                                                                         \
    } while (0)
// the LATEST time any workgroup of the launch passes here (thread 0 of every workgroup; the host clears the slot)
#define MOC_STAMP_MAX(id)                                                                      \
    do {                                                                                       \
        if (threadIdx.x == 0) atomicMax(&g_moc_stamps[id], (unsigned long long)wall_clock64());  \
    } while (0)
#define MOC_STAMP_MIN(id)                                                                      \
    do {                                                                                       \
        if (threadIdx.x == 0) atomicMin(&g_moc_stamps[id], (unsigned long long)wall_clock64());  \
    } while (0)
#else
#define MOC_STAMP(id) do { } while (0)
#define MOC_STAMP_DRAIN(id) do { } while (0)
#define MOC_STAMP_MAX(id) do { } while (0)
#define MOC_STAMP_MIN(id) do { } while (0)
#endif
