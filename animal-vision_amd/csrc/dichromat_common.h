// csrc/dichromat_common.h -- device helpers and argument blocks shared by the dichromat kernels
// (dichromat.hip: reference + 2-D tiled variants; dichromat_march.hip: marching strip variant).
#pragma once
#include "avx_internal.h"

namespace avxk {

constexpr int kThreads = 256;  // reference kernel workgroup

template <typename T>
struct Taps {
    T k[AVX_MAX_KSIZE];
};

struct DichromatArgs {
    const uint8_t* in;
    uint8_t* out;
    int n_frames, H, W;
    int tiles_x, tiles_y;
    int TW, TH;          // output tile
    int r;               // blur radius (0 when no Gaussian)
    float M[9];          // AVX_COLOR_MATRIX, or RGB->LMS for cat
    double Bk[9];        // cat: LMS->RGB (float64)
    float alpha, one_minus_alpha;
    int post_mode;
    const float* row_gain;  // device, H entries
    int row_gain_clamp;
    int chroma_enable;
    float chroma_keep;   // float32(1 - strength)
    const float* decode_lut;
    const void* enc_thr;  // T[256]
    uint32_t* flags;      // per frame: set when any byte > 1 was seen
    unsigned long long* stamps;  // diagnostic builds only (8 accumulators), else NULL
    int ablate;                  // AVX_ABLATE (tuning only): bit mask of phases to skip; 0 in production
    int in_f32;                  // input is a float32 HWC frame in [0,1] (reference kernel only)
};

__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

// "Am I the last workgroup to arrive?" for grids of hundreds to thousands of workgroups.  One counter serialises every arrival at one address (agent-scope atomics
// are performed at the memory side: ~10-25 ns each back to back, i.e. 10-25 us for 1,024 workgroups -- most of the duration of the small reduction kernels that end
// with it); here a workgroup arrives at counter 1 + (block % kTicketFan) and only the one that completes its group goes on to counter 0: <= nblocks / 32 + 32 arrivals
// deep.  Called by ONE thread of each workgroup after its own stores have completed (s_waitcnt vmcnt(0)); tk: 1 + kTicketFan words, zero at rest and left zero.
constexpr unsigned kTicketFan = 32;
// reuse: the counters are used again INSIDE the same launch (k_sel_all): a group's reset is then waited for before its finisher moves on (between launches the kernel
// boundary orders it).
__device__ __forceinline__ bool ticket_is_last(uint32_t* tk, unsigned block, unsigned nblocks, bool reuse = false) {
    const unsigned fan = nblocks < kTicketFan ? nblocks : kTicketFan, grp = block % kTicketFan;
    const unsigned gsize = (nblocks - grp + kTicketFan - 1) / kTicketFan;  // blocks b < nblocks with b % kTicketFan == grp
    if (__hip_atomic_fetch_add(&tk[1 + grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gsize - 1) return false;
    __hip_atomic_store(&tk[1 + grp], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (reuse) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (__hip_atomic_fetch_add(&tk[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != fan - 1) return false;
    __hip_atomic_store(&tk[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// count of thresholds <= v (thr ascending, 255 real entries + 1 huge pad)
template <typename T>
__device__ __forceinline__ uint32_t quantize(T v, const T* __restrict__ thr) {
    v = v < (T)0 ? (T)0 : (v > (T)1 ? (T)1 : v);
    uint32_t lo = 0;
#pragma unroll
    for (int step = 128; step >= 1; step >>= 1) {
        // invariant: thr[lo-1] <= v (or lo == 0); test whether lo+step thresholds are all <= v
        if (thr[lo + step - 1] <= v) lo += step;
    }
    return lo;  // 0..255
}

struct QuantCoarse {
    const uint8_t* table;  // device: count of thresholds <= bucket start, per bucket
    uint32_t lo_key;       // key just BEFORE the first bucket: table index = max(key, lo_key) - lo_key, entry 0 quantises to 0
    uint32_t n_keys;
    int n_fix;             // max thresholds strictly inside one bucket (refinement steps)
};

// Bucket key = sign/exponent + the top 7 mantissa bits: 1,634 buckets between the first threshold and 1.0 (a 2 KiB table in
// LDS), at most ONE encode threshold strictly inside a bucket, so one refinement compare suffices (6 bits: 818 buckets, two).
constexpr int kCoarseShiftF32 = 16, kCoarseShiftF64 = 45, kCoarseTableBytes = 2048, kCoarseNFix = 1;
__device__ __forceinline__ uint32_t key_of(float v) { return __float_as_uint(v) >> kCoarseShiftF32; }
__device__ __forceinline__ uint32_t key_of(double v) { return (uint32_t)(__double_as_longlong(v) >> kCoarseShiftF64); }

// out = #{k : thr[k] <= clip(v,0,1)} via a bucket table on the float's top bits + <= NFIX refinements.
template <typename T, int NFIX>
__device__ __forceinline__ uint32_t quantize_coarse(T v, const T* __restrict__ thr, const uint8_t* __restrict__ coarse, uint32_t lo_key) {
    v = v < (T)0 ? (T)0 : (v > (T)1 ? (T)1 : v);
    // table[0] is the "below the first bucket" entry (0) and lo_key the key just before the first bucket: one max + one
    // subtract, no branch around the lookup (a conditional load compiled to a divergent branch per sample)
    const uint32_t key = key_of(v);
    uint32_t k = (uint32_t)coarse[(key > lo_key ? key : lo_key) - lo_key];
#pragma unroll
    for (int i = 0; i < NFIX; ++i) k += (thr[k] <= v) ? 1u : 0u;  // thr[255] is a huge pad: never passes
    return k;
}

template <typename T> struct Vec16;
template <> struct Vec16<float> { using type = float4; };
template <> struct Vec16<double> { using type = double2; };
template <typename T> struct Vec8;
template <> struct Vec8<float> { using type = float2; };
template <> struct Vec8<double> { using type = double; };


}  // namespace avxk

// variant 2 (dichromat_march.hip): returns AVX_ERR_UNSUPPORTED when no instantiation covers (type, radius).
int avx_launch_dichromat_march(avx_ctx* ctx, avxk::DichromatArgs& a, const avx_dichromat_desc* d, bool f64_cat, hipStream_t s);

// AVX_POST_STREAK (dichromat_streak.hip): float32 colour matrix only, like every streak species.
int avx_launch_dichromat_streak(avx_ctx* ctx, avx_ws* ws, avxk::DichromatArgs& a, const avx_dichromat_desc* d, hipStream_t s);
