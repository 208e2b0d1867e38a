// csrc/mst_common.h -- shared by the MST++ matrix-core kernels (mst_mfma.hip, mst_fused.hip).
#pragma once
#include <hip/hip_fp16.h>

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef float float2_t __attribute__((ext_vector_type(2)));

// GELU (exact-erf form, nn.GELU() of MST_Plus_Plus.py:48-50) for a float16 result, as x * sigmoid(x * P(x^2)):
// Phi(x) = sigmoid(logit Phi(x)) holds exactly and logit Phi is odd and smooth, so a degree-4 polynomial P in x^2 (a weighted
// minimax fit over |x| <= 9, the weight being the sensitivity x^2 Phi (1 - Phi) of the result to P) gives |error| <= 3.5e-6
// absolute over all finite float16 inputs in float32 evaluation -- 1/35 of half a float16 ulp at 0.25.  Cost per element: 6
// multiply/FMA-class operations, one v_exp_f32 and one v_rcp_f32 (the Abramowitz-Stegun 7.1.28 form it replaces: 13 + a
// v_rcp_f32 + sign and magnitude fix-ups; these kernels are bound by exactly this arithmetic).  The coefficients carry the
// factor -log2(e) so that the hardware's base-2 exponential applies directly; saturation is by IEEE arithmetic
// (exp2(-inf) = 0 -> x; exp2(+inf) = inf -> rcp = 0 -> -0), no branches, no clamps.
__device__ __forceinline__ float gelu_fast(float x) {
    constexpr float C0 = -0x1.26a96cp+1f, C1 = -0x1.af022ep-4f, C2 = 0x1.79c67ep-12f, C3 = 0x1.7218f2p-14f, C4 = -0x1.b16328p-19f;
    const float u = x * x;
    float p = __builtin_fmaf(u, C4, C3);
    p = __builtin_fmaf(p, u, C2);
    p = __builtin_fmaf(p, u, C1);
    p = __builtin_fmaf(p, u, C0);
    const float e = __builtin_amdgcn_exp2f(x * p);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// The same on a pair with packed float32 arithmetic: a wave64 vector instruction holds its SIMD for 4 cycles whether it is
// v_mul_f32 or v_pk_mul_f32 (measured: SQ_ACTIVE_INST_VALU ~ SQ_INSTS_VALU in quad-cycles), so the packed forms halve the cost
// of everything but the two transcendentals per element.
// The sigmoid form on a pair (round 2's first version; kept for A/B: -DAVX_GELU_SIGMOID).
__device__ __forceinline__ float2_t gelu_sig2(float2_t x) {
    constexpr float C0 = -0x1.26a96cp+1f, C1 = -0x1.af022ep-4f, C2 = 0x1.79c67ep-12f, C3 = 0x1.7218f2p-14f, C4 = -0x1.b16328p-19f;
    auto c2 = [](float v) { return float2_t{v, v}; };
    const float2_t u = x * x;
    float2_t p = __builtin_elementwise_fma(u, c2(C4), c2(C3));
    p = __builtin_elementwise_fma(p, u, c2(C2));
    p = __builtin_elementwise_fma(p, u, c2(C1));
    p = __builtin_elementwise_fma(p, u, c2(C0));
    const float2_t t = x * p;
    const float2_t d = float2_t{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + c2(1.0f);
    return x * float2_t{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}

// GELU without transcendentals: x * Phi(x), Phi(x) - 1/2 = xc * R(xc^2) with xc = clamp(x, -4, 4) and R a degree-7 minimax polynomial pinned
// so that Phi(+-4) = 1 / 0 exactly (the clamp continues the function flat: the error stays bounded however large |x| is).
// |Phi error| <= 3.2e-5 (at |x| = 4, where the true tail is 3.2e-5), |gelu error| <= 1.3e-4 absolute at x = -4 and <= 3.2e-5 relative for x > 0;
// end to end it moves the float32 network's output by 2.5e-6 on average (max 1.7e-5) -- 2 % of what float16 storage alone does.  Two v_med3_f32 +
// ten packed float32 operations per PAIR: on gfx950 a packed FMA costs ~1.95 ns per wave and SIMD, v_exp_f32 / v_rcp_f32 3.6 ns each
// (tools/experiments/valu_rate4.hip), so this is 23 ns per pair against the sigmoid form's 30.
// Degree and clamp are a build choice (-DAVX_GELU_DEG=7|6|5; tools/experiments/gelu_fit.py prints the tables: exact minimax fits, by linear programming,
// of max(|error|, |error| / gelu) with Phi(+-A) pinned to 1 / 0):
//   7 (A = 4):    |gelu error| <= 1.3e-4 absolute (at x = -4, the true tail), <= 3.2e-5 relative for x > 0   -- round 2's table
//   6 (A = 4):    <= 2.4e-4 absolute and relative: half of the worst-case float16 rounding (4.9e-4) that follows it
//   5 (A = 3.75): <= 6.7e-4
// The default is 5: two packed operations of twelve less per pair in kernels that are bound by exactly this arithmetic.  Measured against the reference's
// float32 outputs (tests/test_mstpp.py::test_forward_fp16_large_frames_vs_reference, 256 x 256 ... 3840 x 2160): the mean / rms error of the float16 forward
// pass is the same to within 1-4 % for degrees 7, 6 and 5 (worst crop: mean 1.755e-4 / 1.755e-4 / 1.758e-4) -- float16 storage decides it, not this fit --
// and every bound of the test holds unchanged; same-box A/B of the 4K frame: 44.7 / 44.2 / 43.2 ms per two-frame step (profiles/r03/ab_gelu_degree.txt).
#ifndef AVX_GELU_DEG
#define AVX_GELU_DEG 5
#endif
#if AVX_GELU_DEG == 7
#define AVX_GELU_A 4.0f
#define AVX_GELU_COEFFS {0x1.9860cap-2f, -0x1.0e9c60p-4f, 0x1.3a08cap-7f, -0x1.0bd734p-10f, 0x1.4005e4p-14f, -0x1.f322cap-19f, 0x1.c24602p-24f, -0x1.61225cp-30f}
#elif AVX_GELU_DEG == 6
#define AVX_GELU_A 4.0f
#define AVX_GELU_COEFFS {0x1.97e942p-2f, -0x1.0a57bcp-4f, 0x1.249574p-7f, -0x1.b85e3cp-11f, 0x1.a4e7c2p-15f, -0x1.c69c8ep-20f, 0x1.a2f00ap-26f}
#elif AVX_GELU_DEG == 5
#define AVX_GELU_A 3.75f
#define AVX_GELU_COEFFS {0x1.96ec78p-2f, -0x1.032bf4p-4f, 0x1.070972p-7f, -0x1.4ea816p-11f, 0x1.d5c108p-16f, -0x1.12e0acp-21f}
#else
#error "AVX_GELU_DEG must be 5, 6 or 7"
#endif
__device__ __forceinline__ float2_t gelu_poly2(float2_t x) {
    constexpr float R[AVX_GELU_DEG + 1] = AVX_GELU_COEFFS;
    auto c2 = [](float v) { return float2_t{v, v}; };
    const float2_t xc = float2_t{__builtin_amdgcn_fmed3f(x.x, -AVX_GELU_A, AVX_GELU_A), __builtin_amdgcn_fmed3f(x.y, -AVX_GELU_A, AVX_GELU_A)};
    const float2_t t = xc * xc;
    float2_t r = __builtin_elementwise_fma(t, c2(R[AVX_GELU_DEG]), c2(R[AVX_GELU_DEG - 1]));
#pragma unroll
    for (int k = AVX_GELU_DEG - 2; k >= 0; --k) r = __builtin_elementwise_fma(r, t, c2(R[k]));
    return x * __builtin_elementwise_fma(xc, r, c2(0.5f));
}

// NP pairs at once, STEP-major: every Horner step is issued for all pairs before the next one, so a wave has NP independent instructions between
// an instruction and its consumer (pair-major code -- one whole chain after the other -- leaves the vector unit waiting out each result's latency
// whenever fewer than ~4 waves share the SIMD, and the compiler does not interleave the chains by itself).  The asm barriers pin the order.
template <int NP>
__device__ __forceinline__ void gelu_multi(float2_t (&x)[NP]) {
#ifdef AVX_GELU_SIGMOID
#pragma unroll
    for (int i = 0; i < NP; ++i) x[i] = gelu_sig2(x[i]);
#else
    constexpr float R[AVX_GELU_DEG + 1] = AVX_GELU_COEFFS;
    auto c2 = [](float v) { return float2_t{v, v}; };
    float2_t xc[NP], t[NP], r[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) xc[i] = float2_t{__builtin_amdgcn_fmed3f(x[i].x, -AVX_GELU_A, AVX_GELU_A), __builtin_amdgcn_fmed3f(x[i].y, -AVX_GELU_A, AVX_GELU_A)};
#pragma unroll
    for (int i = 0; i < NP; ++i) t[i] = xc[i] * xc[i];
#pragma unroll
    for (int i = 0; i < NP; ++i) r[i] = __builtin_elementwise_fma(t[i], c2(R[AVX_GELU_DEG]), c2(R[AVX_GELU_DEG - 1]));
#pragma unroll
    for (int k = AVX_GELU_DEG - 2; k >= 0; --k) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            r[i] = __builtin_elementwise_fma(r[i], t[i], c2(R[k]));
            asm volatile("" : "+v"(r[i]));  // keeps the step-major order through the scheduler
        }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) r[i] = __builtin_elementwise_fma(xc[i], r[i], c2(0.5f));
#pragma unroll
    for (int i = 0; i < NP; ++i) x[i] = x[i] * r[i];
#endif
}

// ---- prescaled form (round 3): six packed + four full-rate instructions per pair instead of ten packed-rate ones -----------------------------------------------------------------
// The two v_med3_f32 of gelu_poly2 exist to clamp x; float32 instructions clamp their RESULT to [0, 1] for free (the `clamp` bit), so the
// polynomial is arranged for that to be the only clamp needed: the operator in front of the GELU delivers x' = x / A (its weights are scaled by 1 / A on the
// host -- A = 4, a power of two: exact in float16), and
//     t   = clamp01(x' x')                 v_mul_f32 ... clamp         (= (xc / A)^2)
//     r   = R^(t)                          Horner, R^(t) = A R(A^2 t)  (the same minimax fit, coefficients rescaled)
//     Phi = clamp01(x' r + 1/2)            v_fma_f32 ... clamp         (|x| > A: R^(1) = 1/2, so the sum leaves [0, 1] and the clamp pins Phi to 0 / 1)
//     out = x' Phi                         = gelu(x) / A: the operator behind takes its weights times A (depthwise conv -> the second GELU sees x2 / A again; the
//                                            last 1x1 conv / second depthwise conv restores the scale)
// Inside |x| <= A this is gelu_poly2's value; outside it is x or 0 exactly.  Degree 5 at A = 4: |error| <= 1.1e-3 relative (x > 0) / absolute (x < 0); degree 6:
// 2.4e-4 (tools/experiments/gelu_fit.py; -DAVX_GELU_PRE_DEG).  -DAVX_GELU_PRE=0 builds the kernels with gelu_poly2 and unscaled weights (avx_mst_gelu_prescale()
// tells the host which).
#ifndef AVX_GELU_PRE
#define AVX_GELU_PRE 1
#endif
#ifndef AVX_GELU_PRE_DEG
#define AVX_GELU_PRE_DEG 5
#endif
#if AVX_GELU_PRE_DEG == 5
#define AVX_GELU_PRE_COEFFS {0x1.960306p+0f, -0x1.fc1ff8p+1f, 0x1.ee409cp+2f, -0x1.262110p+3f, 0x1.7b5900p+2f, -0x1.932106p+0f}
#elif AVX_GELU_PRE_DEG == 6
#define AVX_GELU_PRE_COEFFS {0x1.97e942p+0f, -0x1.0a57bcp+2f, 0x1.249574p+3f, -0x1.b85e3cp+3f, 0x1.a4e7c2p+3f, -0x1.c69c8ep+2f, 0x1.a2f00ap+0f}
#else
#error "AVX_GELU_PRE_DEG must be 5 or 6"
#endif
constexpr float kGeluPrescale = AVX_GELU_PRE ? 4.0f : 1.0f;
// Written as fmed3(v, 0, 1) of the product / FMA: the compiler folds that into the instruction's clamp bit (v_mul_f32 / v_fma_f32 ... clamp) and keeps its own
// hazard bookkeeping -- as inline assembly the instructions sat directly behind the packed FMAs whose results they read, without the wait states the compiler
// inserts between such pairs on gfx950, and read stale registers (measured: wrong values in 60 % of the outputs).
__device__ __forceinline__ float2_t pk_sq_clamp(float2_t a) {
    return float2_t{__builtin_amdgcn_fmed3f(a.x * a.x, 0.0f, 1.0f), __builtin_amdgcn_fmed3f(a.y * a.y, 0.0f, 1.0f)};
}
__device__ __forceinline__ float2_t pk_fma_half_clamp(float2_t a, float2_t b) {  // clamp01(a b + 1/2)
    return float2_t{__builtin_amdgcn_fmed3f(__builtin_fmaf(a.x, b.x, 0.5f), 0.0f, 1.0f), __builtin_amdgcn_fmed3f(__builtin_fmaf(a.y, b.y, 0.5f), 0.0f, 1.0f)};
}
// NP pairs step-major (as gelu_multi); x holds x / A on entry, gelu(x) / A on return
template <int NP>
__device__ __forceinline__ void gelu_pre_multi(float2_t (&x)[NP]) {
    constexpr int D = AVX_GELU_PRE_DEG;
    constexpr float R[D + 1] = AVX_GELU_PRE_COEFFS;
    auto c2 = [](float v) { return float2_t{v, v}; };
    float2_t t[NP], r[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) t[i] = pk_sq_clamp(x[i]);
#pragma unroll
    for (int i = 0; i < NP; ++i) r[i] = __builtin_elementwise_fma(t[i], c2(R[D]), c2(R[D - 1]));
#pragma unroll
    for (int k = D - 2; k >= 0; --k) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            r[i] = __builtin_elementwise_fma(r[i], t[i], c2(R[k]));
            if (NP > 1) asm volatile("" : "+v"(r[i]));  // keeps the step-major order through the scheduler
        }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) r[i] = pk_fma_half_clamp(x[i], r[i]);
#pragma unroll
    for (int i = 0; i < NP; ++i) x[i] = x[i] * r[i];
}
// the GELU of the matrix-pipe kernels (k_mst_ffn_fused<DWM>, k_mst_attn_tail_mx): prescaled operands when built so
template <int NP>
__device__ __forceinline__ void gelu_mx(float2_t (&x)[NP]) {
#if AVX_GELU_PRE
    gelu_pre_multi<NP>(x);
#else
    gelu_multi<NP>(x);
#endif
}

__device__ __forceinline__ float2_t gelu_fast2(float2_t x) {
#ifdef AVX_GELU_SIGMOID
    return gelu_sig2(x);
#else
    return gelu_poly2(x);
#endif
}

// a.f16[half] * b.f16[half] + c in float32, c a separate operand
__device__ __forceinline__ float fma_mix_lo_c(unsigned a, unsigned b, float c) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float fma_mix_hi_c(unsigned a, unsigned b, float c) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// two float32 -> packed float16 pair, round to nearest even (v_cvt_pk_f16_f32)
__device__ __forceinline__ unsigned pack_f16(float a, float b) {
    typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
    const half2_t v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, v);
}

// acc += a.f16[half] * b.f16[half] in float32 (v_fma_mix_f32: both float16 operands are converted inside the FMA; the
// compiler only folds the conversions when float32 denormals are flushed, which these translation units do not ask for)
__device__ __forceinline__ void fma_mix_lo(float& acc, unsigned a, unsigned b) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void fma_mix_hi(float& acc, unsigned a, unsigned b) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(a), "v"(b));
}
