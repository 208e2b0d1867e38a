// csrc/uv.hip -- UV / spectral path kernels for gfx950.
//
// Replaces, per frame, the NumPy bodies of (paths relative to the reference):
//   classic_rgb_to_hsi analytic lobes   ml/classic_rgb_to_hsi/classic_rgb_to_hsi.py:47-82   (a13)
//   hsi * E, tensordot cone catches      animals/honeybee.py:125-135                          (a15, a16)
//   integrate_band / bandpass weights    uv_helpers.py:142-146                                (a17)
//   von_kries_white_patch / gray_world   uv_helpers.py:195-206, safe_norm :47-53              (a18)
//   gaussian_blur                        uv_helpers.py:67-73 (cv2 semantics, shared contract) (a19)
//   np.percentile (linear interpolation) uv_mappers.py:32,61-62,73,104,142                    (a20, a21)
//   map_opponent / falsecolor / ...      uv_mappers.py:29-144, then honeybee.py:166-173        (a20, a21, a11)
//
// Float contract: within 1e-4 relative of the reference for the spectral math (the transcendental
// functions - atan2f, powf - are the device's); exact order statistics for the percentiles; the
// final uint8 encode uses the same threshold table as the dichromat path.
// Layout: K float32 planes of H*W (struct-of-arrays: every pass streams planes with 16-byte accesses).
#include <hip/hip_fp16.h>

#include "dichromat_common.h"
#include "stack_up.h"

using namespace avxk;

namespace {

constexpr int kT = 256;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Per-block partial {min, max, sum} of up to 16 planes -> partials[block][k][3] (sum kept in double).
struct Stat3 { float mn, mx; double sum; };

template <int KMAX>
__device__ __forceinline__ void block_stats_store(const float (&mn)[KMAX], const float (&mx)[KMAX], const double (&sm)[KMAX], int K,
                                                  Stat3* partials) {
    __shared__ float s_mn[kT / 64][KMAX], s_mx[kT / 64][KMAX];
    __shared__ double s_sm[kT / 64][KMAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < K; ++k) {
        const float a = wave_min(mn[k]), b = wave_max(mx[k]);
        const double c = wave_sum(sm[k]);
        if (lane == 0) { s_mn[wave][k] = a; s_mx[wave][k] = b; s_sm[wave][k] = c; }
    }
    __syncthreads();
    if (threadIdx.x < K) {
        float a = s_mn[0][threadIdx.x], b = s_mx[0][threadIdx.x];
        double c = s_sm[0][threadIdx.x];
        for (int w = 1; w < kT / 64; ++w) { a = fminf(a, s_mn[w][threadIdx.x]); b = fmaxf(b, s_mx[w][threadIdx.x]); c += s_sm[w][threadIdx.x]; }
        partials[(size_t)blockIdx.x * K + threadIdx.x] = Stat3{a, b, c};
    }
}

// ---- RGB uint8 -> K planes through the decode table and a K x 3 matrix (folded lobes x illuminant x
//      receptor curves).  4 pixels (12 bytes = 3 aligned dwords) per thread. --------------------------
template <int KMAX>
__global__ __launch_bounds__(kT) void k_rgb_to_planes(const uint8_t* __restrict__ in, size_t n, const float* __restrict__ lut_g,
                                                      const float* __restrict__ mat /*K x 3 device*/, int K, float* __restrict__ out /*NULL: statistics only*/,
                                                      Stat3* partials) {
    // blockIdx.y = frame of a batch (frames, planes and partials back to back)
    in += (size_t)blockIdx.y * n * 3;
    if (out) out += (size_t)blockIdx.y * n * K;
    partials += (size_t)blockIdx.y * gridDim.x * K;
    __shared__ float lut[256];
    __shared__ float m[KMAX * 3];
    for (int i = threadIdx.x; i < 256; i += kT) lut[i] = lut_g[i];
    for (int i = threadIdx.x; i < K * 3; i += kT) m[i] = mat[i];
    __syncthreads();
    float mn[KMAX], mx[KMAX];
    double sm[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { mn[k] = 3.4e38f; mx[k] = -3.4e38f; sm[k] = 0.0; }
    const size_t nq = n / 4;
    const bool aligned = ((uintptr_t)in & 3u) == 0 && ((uintptr_t)out & 15u) == 0 && (n % 4 == 0) && ((n * K) % 4 == 0);
    for (size_t q = (size_t)blockIdx.x * kT + threadIdx.x; q < nq + 1; q += (size_t)gridDim.x * kT) {
        const size_t p0 = q * 4;
        const int cnt = q < nq ? 4 : (int)(n - p0);
        if (cnt <= 0) break;
        uint32_t code[4][3];
        if (cnt == 4 && ((uintptr_t)in & 3u) == 0) {
            const uint32_t* d = reinterpret_cast<const uint32_t*>(in + p0 * 3);
            const uint32_t u0 = d[0], u1 = d[1], u2 = d[2];
            code[0][0] = u0 & 255u; code[0][1] = (u0 >> 8) & 255u; code[0][2] = (u0 >> 16) & 255u;
            code[1][0] = u0 >> 24;  code[1][1] = u1 & 255u;        code[1][2] = (u1 >> 8) & 255u;
            code[2][0] = (u1 >> 16) & 255u; code[2][1] = u1 >> 24; code[2][2] = u2 & 255u;
            code[3][0] = (u2 >> 8) & 255u; code[3][1] = (u2 >> 16) & 255u; code[3][2] = u2 >> 24;
        } else {
            for (int p = 0; p < 4; ++p)
                for (int c = 0; c < 3; ++c) code[p][c] = p < cnt ? in[(p0 + p) * 3 + c] : 0u;
        }
        float c0[4], c1[4], c2[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) { c0[p] = lut[code[p][0]]; c1[p] = lut[code[p][1]]; c2[p] = lut[code[p][2]]; }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k < K) {
                float v[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    v[p] = fma_t(c2[p], m[3 * k + 2], fma_t(c1[p], m[3 * k + 1], c0[p] * m[3 * k]));
                    if (p < cnt) { mn[k] = fminf(mn[k], v[p]); mx[k] = fmaxf(mx[k], v[p]); sm[k] += (double)v[p]; }
                }
                if (out) {
                    float* o = out + (size_t)k * n + p0;
                    if (cnt == 4 && aligned) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
                    else for (int p = 0; p < cnt; ++p) o[p] = v[p];
                }
            }
        }
    }
    block_stats_store<KMAX>(mn, mx, sm, K, partials);
}

// ---- HSI cube -> K planes: out_k = sum_b cube_b * w[k][b] (sequential over b, f32 accumulate) -------
// layout 0: NHWC (pixel-major, bands contiguous)   layout 1: NCHW (band planes).  dtype 0: f32, 1: f16.
template <int KMAX>
__global__ __launch_bounds__(kT) void k_spectral_integrate(const void* __restrict__ cube, int layout, int dtype, size_t n, int B,
                                                           const float* __restrict__ w_g /*K x B*/, int K, float* __restrict__ out,
                                                           Stat3* partials) {
    extern __shared__ float w[];  // K*B
    for (int i = threadIdx.x; i < K * B; i += kT) w[i] = w_g[i];
    __syncthreads();
    float mn[KMAX], mx[KMAX];
    double sm[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { mn[k] = 3.4e38f; mx[k] = -3.4e38f; sm[k] = 0.0; }
    for (size_t p = (size_t)blockIdx.x * kT + threadIdx.x; p < n; p += (size_t)gridDim.x * kT) {
        float acc[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) acc[k] = 0.0f;
        for (int b = 0; b < B; ++b) {
            const size_t idx = layout == 0 ? p * B + b : (size_t)b * n + p;
            const float x = dtype == 0 ? reinterpret_cast<const float*>(cube)[idx]
                                       : __half2float(reinterpret_cast<const __half*>(cube)[idx]);
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) acc[k] = fma_t(x, w[k * B + b], acc[k]);
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) {
                out[(size_t)k * n + p] = acc[k];
                mn[k] = fminf(mn[k], acc[k]); mx[k] = fmaxf(mx[k], acc[k]); sm[k] += (double)acc[k];
            }
    }
    block_stats_store<KMAX>(mn, mx, sm, K, partials);
}

// The same for a float16 NHWC cube (what an MST++-style model emits; BASELINE config 5), at HBM rate: pixels are
// B*2 bytes (62 for 31 bands) -- unaligned for any vector load done per pixel -- so a workgroup stages the
// contiguous span of its 256 pixels in LDS with 16-byte loads and each thread then walks its own pixel there.
// Weights arrive band-major and zero-padded to KP ([B][KP]): for one band the KP weights are wave-uniform and
// contiguous, i.e. scalar loads feeding the FMAs as SGPR operands.  Same FMA order over b as the generic kernel.
template <int KP, int NV>
__global__ __launch_bounds__(kT, NV > 4 ? 2 : 1) void k_spectral_nhwc_h(const __half* __restrict__ cube, size_t n, int B, const float* __restrict__ wT /*[B][KP]*/, int K,
                                                        float* __restrict__ out, Stat3* partials) {
    extern __shared__ __align__(16) unsigned char tile_raw[];
    const __half* tile = reinterpret_cast<const __half*>(tile_raw);
    float mn[KP], mx[KP];
    double sm[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) { mn[k] = 3.4e38f; mx[k] = -3.4e38f; sm[k] = 0.0; }
    const int t = threadIdx.x;
    // Software pipeline (NV > 0): the next tile's 16-byte vectors are fetched into registers while this tile is integrated from
    // LDS (a tile is 256 * B * 2 bytes = NV vectors per thread: 4 for B <= 32, 11 for B <= 88; wider cubes take the plain path,
    // NV = 0), so the HBM round trip of a tile is covered by the arithmetic of the previous one, not by other workgroups alone.
    // Loads past the tile's end are clamped to its last vector rather than predicated: the array stays in registers.
    constexpr bool piped = NV > 0;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 pre[NV > 0 ? NV : 1];
#define AVX_FETCH(P0)                                                                                                              \
    {                                                                                                                              \
        const size_t q0_ = (P0);                                                                                                   \
        const int np_ = (int)(n - q0_ < (size_t)kT ? n - q0_ : kT);                                                                \
        const int last_ = (int)(((size_t)np_ * B * 2) / 16) - 1;                                                                   \
        const u32x4* src_ = reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(cube) + q0_ * B * 2);            \
        if (last_ >= 0) {                                                                                                          \
            _Pragma("unroll") for (int k = 0; k < NV; ++k) {                                                                       \
                const int i_ = t + k * kT;                                                                                         \
                pre[k] = __builtin_nontemporal_load(src_ + (i_ < last_ ? i_ : last_));                                             \
            }                                                                                                                      \
        }                                                                                                                          \
    }
    float acc[KP];
    bool have = false;
    size_t put_at = 0;
#define AVX_PUT()                                                                                                                  \
    {                                                                                                                              \
        _Pragma("unroll") for (int k = 0; k < KP; ++k)                                                                             \
            if (k < K) {                                                                                                           \
                __builtin_nontemporal_store(acc[k], out + (size_t)k * n + put_at);                                                 \
                mn[k] = fminf(mn[k], acc[k]); mx[k] = fmaxf(mx[k], acc[k]); sm[k] += (double)acc[k];                               \
            }                                                                                                                      \
    }
    if (piped && (size_t)blockIdx.x * kT < n) AVX_FETCH((size_t)blockIdx.x * kT)
    for (size_t p0 = (size_t)blockIdx.x * kT; p0 < n; p0 += (size_t)gridDim.x * kT) {
        const int np = (int)(n - p0 < (size_t)kT ? n - p0 : kT);
        const size_t bytes = (size_t)np * B * 2;
        const unsigned char* src = reinterpret_cast<const unsigned char*>(cube) + p0 * B * 2;  // 16-byte aligned: p0 % 256 == 0
        __syncthreads();  // the previous tile is no longer read
        const int nvec = (int)(bytes / 16);
        if (piped) {
            if (nvec > 0) {  // clamped like the fetch: the surplus lanes rewrite the tile's last vector with its own value
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const int i = t + k * kT;
                    reinterpret_cast<u32x4*>(tile_raw)[i < nvec - 1 ? i : nvec - 1] = pre[k];
                }
            }
        } else {
            for (int i = t; i < nvec; i += kT) reinterpret_cast<uint4*>(tile_raw)[i] = reinterpret_cast<const uint4*>(src)[i];
        }
        for (int i = nvec * 16 + t * 2; i < (int)bytes; i += kT * 2) *reinterpret_cast<uint16_t*>(tile_raw + i) = *reinterpret_cast<const uint16_t*>(src + i);
        __syncthreads();
        // the previous tile's catches are stored here, ahead of the next fetch: the wait for that fetch at the top of the loop is then
        // not also a wait for stores issued after it (one in-order counter covers both), and the stores get a whole tile of arithmetic to drain
        if (have) AVX_PUT()
        if (piped && p0 + (size_t)gridDim.x * kT < n) AVX_FETCH(p0 + (size_t)gridDim.x * kT)
        have = t < np;
        put_at = p0 + t;
        if (have) {
#pragma unroll
            for (int k = 0; k < KP; ++k) acc[k] = 0.0f;
            const __half* px = tile + (size_t)t * B;
#pragma unroll 4
            for (int b = 0; b < B; ++b) {
                const float x = __half2float(px[b]);
                const float* wb = wT + b * KP;  // wave-uniform
#pragma unroll
                for (int k = 0; k < KP; ++k) acc[k] = fma_t(x, wb[k], acc[k]);
            }
        }
    }
    if (have) AVX_PUT()
#undef AVX_PUT
#undef AVX_FETCH
    block_stats_store<KP>(mn, mx, sm, K, partials);
}

// ---- plain statistics of existing planes ----------------------------------------------------------
template <int KMAX>
__global__ __launch_bounds__(kT) void k_plane_stats(const float* __restrict__ planes, size_t n, int K, Stat3* partials) {
    float mn[KMAX], mx[KMAX];
    double sm[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { mn[k] = 3.4e38f; mx[k] = -3.4e38f; sm[k] = 0.0; }
    for (size_t p = (size_t)blockIdx.x * kT + threadIdx.x; p < n; p += (size_t)gridDim.x * kT)
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) { const float v = planes[(size_t)k * n + p]; mn[k] = fminf(mn[k], v); mx[k] = fmaxf(mx[k], v); sm[k] += (double)v; }
    block_stats_store<KMAX>(mn, mx, sm, K, partials);
}

// Final reduction (one block) -> stats[k] = {min, max, mean, denominator}; the denominator is what the
// next pass divides by: adapt 0: 1 | 1 white_patch: max(max, eps) | 2 gray_world: max(mean, eps)
// | 3 safe_norm: (max - min), with stats.x = min subtracted first (uv_helpers.py:47-53).
__global__ void k_finalize_stats(const Stat3* partials, int nblocks, int K, size_t n, int adapt, float eps, float4* stats) {
    // one wave per plane; lanes stride over the block partials, fixed-shape shuffle tree (deterministic); blockIdx.x = frame of a batch
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (k >= K) return;
    partials += (size_t)blockIdx.x * nblocks * K;
    stats += (size_t)blockIdx.x * 4;
    float a = 3.4e38f, b = -3.4e38f;
    double c = 0.0;
    for (int i = lane; i < nblocks; i += 64) { const Stat3 s = partials[(size_t)i * K + k]; a = fminf(a, s.mn); b = fmaxf(b, s.mx); c += s.sum; }
    a = wave_min(a); b = wave_max(b); c = wave_sum(c);
    if (lane != 0) return;
    const float mean = (float)(c / (double)n);
    float den = 1.0f;
    if (adapt == 1) den = fmaxf(b, eps);
    else if (adapt == 2) den = fmaxf(mean, eps);
    else if (adapt == 3) den = b - a;
    stats[k] = make_float4(a, b, mean, den);
}

// ---- plane blur: out = GaussianBlur((in - sub) / den), cv2 semantics, shared arithmetic contract ----
// sub/den come from stats (mode 0: none, 1: divide by stats.w, 3: safe_norm -> (x - min)/(max - min), or
// zeros when the range is < 1e-9).  Runtime radius (k <= 33); r == 0 is the pure rescale.
struct BlurArgs {
    const float* in; float* out; int K, H, W, r; const float4* stats; int scale_mode;
    float taps[AVX_MAX_KSIZE];
};

__device__ __forceinline__ float rescale(float v, int mode, float4 st) {
    if (mode == 1) return v / st.w;
    if (mode == 3) return (st.y - st.x) < 1e-9f ? 0.0f : (v - st.x) / (st.y - st.x);
    return v;
}

__global__ __launch_bounds__(kT) void k_plane_blur(BlurArgs a) {
    constexpr int TW = 64, TH = 32;
    extern __shared__ float sm[];
    const int r = a.r, AW = TW + 2 * r, AH = TH + 2 * r;
    float* A = sm;             // AH x AW
    float* Bm = sm + AH * AW;  // AH x TW
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int total = tiles_x * tiles_y * a.K;
    const size_t n = (size_t)a.H * a.W;
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int k = tile / (tiles_x * tiles_y), t2 = tile - k * tiles_x * tiles_y;
        const int ty = t2 / tiles_x, tx = t2 - ty * tiles_x;
        const int x0 = tx * TW, y0 = ty * TH;
        const float* src = a.in + (size_t)k * n;
        float* dst = a.out + (size_t)k * n;
        const float4 st = a.scale_mode ? a.stats[k] : make_float4(0.f, 0.f, 0.f, 1.f);
        for (int i = threadIdx.x; i < AH * AW; i += kT) {
            const int ly = i / AW, lx = i - ly * AW;
            const int gy = reflect101(y0 - r + ly, a.H), gx = reflect101(x0 - r + lx, a.W);
            A[i] = rescale(src[(size_t)gy * a.W + gx], a.scale_mode, st);
        }
        __syncthreads();
        const float* colsrc = A;
        int sw = AW;
        if (r > 0) {
            const int nt = 2 * r + 1;
            for (int i = threadIdx.x; i < AH * TW; i += kT) {
                const int ly = i / TW, x = i - ly * TW;
                const float* row = A + ly * AW + x;
                float s = row[0] * a.taps[0];
                for (int j = 1; j < nt; ++j) s = fma_t(row[j], a.taps[j], s);
                Bm[i] = s;
            }
            __syncthreads();
            colsrc = Bm;
            sw = TW;
        }
        for (int i = threadIdx.x; i < TH * TW; i += kT) {
            const int y = i / TW, x = i - y * TW;
            if (y0 + y >= a.H || x0 + x >= a.W) continue;
            const float* col = colsrc + (y + r) * sw + x;
            float s;
            if (r > 0) {
                s = col[0] * a.taps[r];
                for (int j = 1; j <= r; ++j) s = fma_t(col[j * sw] + col[-j * sw], a.taps[r + j], s);
            } else {
                s = col[0];
            }
            dst[(size_t)(y0 + y) * a.W + x0 + x] = s;
        }
        __syncthreads();
    }
}

// Compile-time radius form of k_plane_blur (same arithmetic: row pass = sequential FMA left to right, column pass = centre tap
// then fma(x[+j] + x[-j], k[r+j], s)).  The generic kernel above does one output per thread per trip with run-time tap loops
// (23 us for a 1080p plane, 0.7 TB/s); here the taps are registers, a thread produces FOUR outputs from a register window (the
// row pass reads its 4 + 2R samples with 16-byte LDS loads), and tiles that do not touch the image border skip the reflection.
template <int R>
__global__ __launch_bounds__(kT) void k_plane_blur_t(BlurArgs a) {
    constexpr int TW = 64, TH = 32, AH = TH + 2 * R, AWP = (TW + 2 * R + 3) & ~3, NT = 2 * R + 1;
    extern __shared__ __align__(16) float sm[];
    float* A = sm;               // AH x AWP (columns x0 - R .. x0 + TW + R - 1 at 0 ..)
    float* Bm = sm + AH * AWP;   // AH x TW
    float tp[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) tp[j] = a.taps[j];
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int total = tiles_x * tiles_y * a.K;
    const size_t n = (size_t)a.H * a.W;
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int k = tile / (tiles_x * tiles_y), t2 = tile - k * tiles_x * tiles_y;
        const int ty = t2 / tiles_x, tx = t2 - ty * tiles_x;
        const int x0 = tx * TW, y0 = ty * TH;
        const float* src = a.in + (size_t)k * n;
        float* dst = a.out + (size_t)k * n;
        const float4 st = a.scale_mode ? a.stats[k] : make_float4(0.f, 0.f, 0.f, 1.f);
        const bool interior = x0 - R >= 0 && x0 + TW + R <= a.W && y0 - R >= 0 && y0 + TH + R <= a.H;  // uniform
        constexpr int AW = TW + 2 * R, NLD = (AH * AW + kT - 1) / kT;
        // every load of the tile in flight together (round 3): written as one loop with the LDS store behind each load, a thread's ten loads were ten dependent
        // memory round trips -- most of a launch's 16-20 us for a 1080p plane
        float ld[NLD];
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int i = threadIdx.x + q * kT, ic = i < AH * AW ? i : AH * AW - 1;
            const int ly = ic / AW, lx = ic - ly * AW;
            int gy = y0 - R + ly, gx = x0 - R + lx;
            if (!interior) { gy = reflect101(gy, a.H); gx = reflect101(gx, a.W); }
            ld[q] = src[(size_t)gy * a.W + gx];
        }
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int i = threadIdx.x + q * kT;
            if (i < AH * AW) {
                const int ly = i / AW, lx = i - ly * AW;
                A[ly * AWP + lx] = rescale(ld[q], a.scale_mode, st);
            }
        }
        __syncthreads();
        // row pass: (row, group of 4 columns) items; window = columns 4g .. 4g + 3 + 2R of the row, 16-byte aligned
        for (int i = threadIdx.x; i < AH * (TW / 4); i += kT) {
            const int ly = i / (TW / 4), g = i - ly * (TW / 4);
            const float4* wp = reinterpret_cast<const float4*>(A + ly * AWP + 4 * g);
            float w[(4 + 2 * R + 3) & ~3];
#pragma unroll
            for (int q = 0; q < (4 + 2 * R + 3) / 4; ++q) {
                float4 v = wp[q];
                // opaque to the compiler: otherwise it re-reads one-sample-shifted copies of the window with ds_read2_b32 / _b64 to
                // feed odd-aligned packed FMAs -- 4-way / 2-way bank conflicts at this 16-byte lane stride (see dichromat_streak.hip)
                asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
                w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
            }
            float o[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                float sacc = w[x] * tp[0];
#pragma unroll
                for (int j = 1; j < NT; ++j) sacc = fma_t(w[x + j], tp[j], sacc);
                o[x] = sacc;
            }
            *reinterpret_cast<float4*>(Bm + ly * TW + 4 * g) = make_float4(o[0], o[1], o[2], o[3]);
        }
        __syncthreads();
        // column pass: (column, group of 4 rows) items; window = rows 4g .. 4g + 3 + 2R of the column
        for (int i = threadIdx.x; i < (TH / 4) * TW; i += kT) {
            const int g = i / TW, x = i - g * TW;
            float w[4 + 2 * R];
#pragma unroll
            for (int q = 0; q < 4 + 2 * R; ++q) w[q] = Bm[(4 * g + q) * TW + x];
#pragma unroll
            for (int yy = 0; yy < 4; ++yy) {
                const int y = 4 * g + yy;
                float sacc = w[yy + R] * tp[R];
#pragma unroll
                for (int j = 1; j <= R; ++j) sacc = fma_t(w[yy + R + j] + w[yy + R - j], tp[R + j], sacc);
                if (y0 + y < a.H && x0 + x < a.W) dst[(size_t)(y0 + y) * a.W + x0 + x] = sacc;
            }
        }
        __syncthreads();
    }
}

// one place that launches a plane blur: the compile-time-radius kernel where an instantiation exists, else the generic one
static int launch_plane_blur(avx_ctx* ctx, const BlurArgs& a, hipStream_t s) {
    const long tiles = (long)((a.W + 63) / 64) * ((a.H + 31) / 32) * a.K;
    const int g = (int)(tiles < (long)ctx->num_cus * 8 ? tiles : (long)ctx->num_cus * 8);  // measured: 8 per CU beats 4 by 3-4 %
#define AVX_BLUR_T(RR)                                                                                                       \
    case RR: {                                                                                                               \
        const size_t lds = sizeof(float) * ((size_t)(32 + 2 * RR) * ((64 + 2 * RR + 3) & ~3) + (size_t)(32 + 2 * RR) * 64);    \
        AVX_HIP(ctx, hipFuncSetAttribute((const void*)k_plane_blur_t<RR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(k_plane_blur_t<RR>, dim3(g), dim3(kT), lds, s, a);                                                 \
        break;                                                                                                               \
    }
    switch (a.r) {
        AVX_BLUR_T(1) AVX_BLUR_T(2) AVX_BLUR_T(3) AVX_BLUR_T(4) AVX_BLUR_T(5) AVX_BLUR_T(6) AVX_BLUR_T(7) AVX_BLUR_T(8) AVX_BLUR_T(9) AVX_BLUR_T(12)
        default: {
            const int AW = 64 + 2 * a.r, AH = 32 + 2 * a.r;
            const size_t lds = sizeof(float) * ((size_t)AH * AW + (size_t)AH * 64);
            AVX_HIP(ctx, hipFuncSetAttribute((const void*)k_plane_blur, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_plane_blur, dim3(g), dim3(kT), lds, s, a);
        }
    }
#undef AVX_BLUR_T
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

// ---- exact order statistic: 3-pass radix select on order-preserving keys ---------------------------
struct SelState {
    uint32_t prefix, mask;      // key bits fixed so far
    unsigned long long rank;    // remaining 0-based rank inside the selected prefix
    uint32_t key_lo, key_hi;    // results: key of x[k] and of x[k+1]
    uint32_t cnt_in_bin;        // size of the finally selected bin (duplicates of x[k])
    uint32_t next_key;          // smallest key > key_lo
    uint32_t next_above;        // smallest key above the selected 22-bit prefix range (last histogram pass)
};

__device__ __forceinline__ uint32_t f2key(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// One launch per radix pass.  Every workgroup histograms its share in LDS and flushes the non-empty bins with global
// atomics; the LAST workgroup to finish (ticket counter behind an agent-scope fence) picks the bin that holds the rank,
// narrows the prefix, re-zeroes the histogram and the ticket, and -- on the last pass -- resolves the successor key
// and performs NumPy's float32 lerp.  Pass 0 takes its state from the arguments: no init launch, no memset.
constexpr int kSelMax = 4;  // order statistics resolved together by one set of passes
struct SelJob { const float* x; size_t n; unsigned long long rank0; float gamma; int has_next; double* out; };
struct SelPass { int n_jobs, pass, shift, bits; SelState* st; uint32_t* hist; uint32_t* ticket; SelJob job[kSelMax];
                 size_t x_frame_stride; int out_frame_stride; };  // blockIdx.y = frame: job arrays x_frame_stride floats apart, results out_frame_stride doubles apart, kSelMax states / histograms and one ticket per frame

// NJ percentiles per launch: the data pass walks the NJ arrays one after the other into NJ LDS histograms, and the
// last workgroup picks for each in turn -- the per-launch fixed costs (LDS clear / flush, ticket, tail latency)
// are paid once, which is most of the time of a pass at 1080p.
template <bool AG>
__device__ void sel_pick_t(uint32_t* hist, SelState* st, int pass, int shift, int bits, unsigned long long rank0, float gamma, int has_next, double* out,
                           uint32_t prefix0, uint32_t mask0);
__device__ __forceinline__ void sel_pick(uint32_t* hist, SelState* st, int pass, int shift, int bits, unsigned long long rank0, float gamma, int has_next, double* out,
                                         uint32_t prefix0 = 0u, uint32_t mask0 = 0u) {
    sel_pick_t<false>(hist, st, pass, shift, bits, rank0, gamma, has_next, out, prefix0, mask0);
}

template <int NJ>
__global__ __launch_bounds__(kT) void k_sel_pass(const SelPass a_in) {
    SelPass a = a_in;
    {   // frame of this workgroup (gridDim.y == 1: the plain single-array form)
        const int f = blockIdx.y;
        a.st += (size_t)f * kSelMax; a.hist += (size_t)f * kSelMax * 2048; a.ticket += f;
        for (int j = 0; j < kSelMax; ++j) { a.job[j].x += (size_t)f * a.x_frame_stride; a.job[j].out += (size_t)f * a.out_frame_stride; }
    }
    __shared__ uint32_t h[NJ][2048];
    __shared__ uint32_t wmin[kT / 64];
    __shared__ int is_last;
    const int nb = 1 << a.bits, t = threadIdx.x;
    const bool last_pass = a.pass == 2;
    // The first eight 16-byte vectors of a thread's share of job j (all of it for a 1080p plane at one workgroup per CU): fetched a
    // job AHEAD -- job 0's while the histograms are cleared, job j + 1's before job j is visited -- so the jobs' memory round trips
    // overlap instead of adding up (a pass over four planes was four dependent round trips).
    const size_t stride = (size_t)gridDim.x * kT, i0 = (size_t)blockIdx.x * kT + t;
    auto job_vec = [&](int j, const float4*& xv, size_t& nvec, size_t& head, size_t& tail0) {
        const SelJob jb = a.job[j];
        const size_t mis = ((16 - ((uintptr_t)jb.x & 15)) & 15) / 4;
        head = mis < jb.n ? mis : jb.n;
        nvec = (jb.n - head) / 4; tail0 = head + nvec * 4;
        xv = reinterpret_cast<const float4*>(jb.x + head);
    };
    auto fetch8 = [&](int j, float4 (&v)[8]) {
        const float4* xv; size_t nvec, head, tail0;
        job_vec(j, xv, nvec, head, tail0);
        if (nvec == 0) return;
#pragma unroll
        for (int q = 0; q < 8; ++q) { const size_t idx = i0 + q * stride; v[q] = xv[idx < nvec ? idx : nvec - 1]; }
    };
    float4 cur[8], nxt[8];
    fetch8(0, cur);
    for (int i = t; i < NJ * 2048; i += kT) (&h[0][0])[i] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (j >= a.n_jobs) break;
        if (j + 1 < NJ && j + 1 < a.n_jobs) fetch8(j + 1 < NJ ? j + 1 : j, nxt);
        const SelJob jb = a.job[j];
        SelState* st = a.st + j;
        const bool find_next = last_pass && jb.has_next;
        const uint32_t prefix = a.pass == 0 ? 0u : st->prefix, mask = a.pass == 0 ? 0u : st->mask;
        const uint32_t above = prefix | ~mask;  // largest key with this prefix
        uint32_t best = 0xffffffffu;
        // Run-length combining: neighbouring samples of a frame usually share a bin (always, nearly, in the first pass, where a bin is a
        // quarter of a binade), and 64 lanes adding 1 to the same LDS word serialise -- a thread keeps (bin, count) of its current run and
        // issues one atomic per run instead of one per sample.
        uint32_t run_bin = 0xffffffffu, run_cnt = 0;
        auto visit = [&](float f) {
            const uint32_t k = f2key(f);
            if ((k & mask) == prefix) {
                const uint32_t b = (k >> a.shift) & (nb - 1);
                if (b == run_bin) { ++run_cnt; }
                else {
                    if (run_cnt) atomicAdd(&h[j][run_bin], run_cnt);
                    run_bin = b; run_cnt = 1;
                }
            } else if (find_next && k > above && k < best) best = k;  // successor candidates outside the 22-bit prefix
        };
        // 16-byte loads over the aligned body; the (<= 3 + 3) head / tail elements go to the first threads of block 0
        const float4* xv; size_t nvec, head, tail0;
        job_vec(j, xv, nvec, head, tail0);
        // first batch: already here (cur); further batches (frames beyond 8 vectors per thread): eight loads in flight each, the ragged
        // end included (clamped index, masked visit)
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (i0 + q * stride < nvec) { visit(cur[q].x); visit(cur[q].y); visit(cur[q].z); visit(cur[q].w); }
        for (size_t i = i0 + 8 * stride; i < nvec; i += 8 * stride) {
            float4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { const size_t idx = i + q * stride; v[q] = xv[idx < nvec ? idx : nvec - 1]; }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (i + q * stride < nvec) { visit(v[q].x); visit(v[q].y); visit(v[q].z); visit(v[q].w); }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) cur[q] = nxt[q];
        if (blockIdx.x == 0) {
            if ((size_t)t < head) visit(jb.x[t]);
            if (tail0 + t < jb.n && t < 4) visit(jb.x[tail0 + t]);
        }
        if (run_cnt) atomicAdd(&h[j][run_bin], run_cnt);
        if (find_next) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const uint32_t v = __shfl_xor(best, o); best = v < best ? v : best; }
            __syncthreads();
            if ((t & 63) == 0) wmin[t >> 6] = best;
            __syncthreads();
            if (t == 0) {
                for (int w = 1; w < kT / 64; ++w) best = wmin[w] < best ? wmin[w] : best;
                if (best != 0xffffffffu) atomicMin(&st->next_above, best);
            }
        }
    }
    __syncthreads();
    for (int j = 0; j < a.n_jobs; ++j)
        for (int i = t; i < nb; i += kT)
            if (h[j][i]) atomicAdd(&a.hist[j * 2048 + i], h[j][i]);
    // ---- last workgroup: pick ------------------------------------------------------------------------------------
    // Everything the workgroups exchange travels in agent-scope atomics (histogram, successor key, ticket) and is
    // read back with agent-scope atomic loads, so no cache write-back / invalidate (__threadfence: a whole-L2 flush
    // per workgroup on this multi-XCD part) is needed: only this wave's atomics must have been performed before
    // the ticket is taken.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // one word per frame in the batched form (blockIdx.y = frame); the single-array form owns the whole ticket block and arrives through the two-level ticket
    if (t == 0) is_last = gridDim.y > 1 ? atomicAdd(a.ticket, 1u) == gridDim.x - 1 : avxk::ticket_is_last(a.ticket, blockIdx.x, gridDim.x);
    __syncthreads();
    if (!is_last) return;
    for (int j = 0; j < a.n_jobs; ++j) {
        const SelJob jb = a.job[j];
        sel_pick(a.hist + j * 2048, a.st + j, a.pass, a.shift, a.bits, jb.rank0, jb.gamma, jb.has_next, jb.out, 0u, 0u);
    }
    if (t == 0 && gridDim.y > 1) *a.ticket = 0;
}

// ---- opponent prep / maps / encode ------------------------------------------------------------------
__global__ __launch_bounds__(kT) void k_opponent_prep(const float* __restrict__ UBG, size_t n, float* __restrict__ radius, float* __restrict__ L) {
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += (size_t)gridDim.x * kT) {
        const float U = UBG[i], B = UBG[n + i], G = UBG[2 * n + i];
        const float O1 = G - B, O2 = B - U;
        L[i] = ((U + B) + G) / 3.0f;
        radius[i] = __fsqrt_rn(O1 * O1 + O2 * O2);
    }
}

struct MapArgs {
    const float* UBG; size_t n; uint8_t* out; int mode;
    const double* pct;       // device: percentiles (mode-dependent)
    float eps; float M[9]; float alpha; float* mixed;  // mixed: 3 planes (mode 4, pass 1)
    const float* enc_thr; const uint8_t* coarse; uint32_t lo_key; int pass;
    float* out_f;            // non-NULL: write float32 sRGB values instead of uint8 codes
};

__device__ __forceinline__ double clip01(double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }
__device__ __forceinline__ float clip01f(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

__device__ __forceinline__ void falsecolor(float U, float B, float G, const double* pct, float eps, float (&rgb)[3]) {
    // uv_mappers.py:29-42: x / max(float(P95), eps) in float32
    const float dU = fmaxf((float)pct[0], eps), dB = fmaxf((float)pct[1], eps), dG = fmaxf((float)pct[2], eps);
    const float Un = U / dU, Bn = B / dB, Gn = G / dG;
    rgb[0] = clip01f(0.85f * Un + 0.10f * Gn);
    rgb[1] = clip01f(0.80f * Gn + 0.20f * Bn);
    rgb[2] = clip01f(0.70f * Bn + 0.40f * Un);
}

__device__ __forceinline__ float s2l(float v) { return v <= 0.04045f ? v / 12.92f : powf((v + 0.055f) / 1.055f, 2.4f); }

__device__ __forceinline__ void purple_yellow_soft(float U, double p98, float eps, float (&rgb)[3]) {
    // uv_mappers.py:90-132 with the default u_gamma 0.90, accent_gamma 0.85, accent_strength 0.05
    const float denom = fmaxf((float)p98, eps);
    const float u = powf(clip01f(U / denom), 0.90f);
    const float c0[3] = {s2l(176.f / 255.f), s2l(124.f / 255.f), s2l(232.f / 255.f)};
    const float c1[3] = {s2l(255.f / 255.f), s2l(211.f / 255.f), s2l(138.f / 255.f)};
    const float w = powf(u, 0.85f);
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = ((1.0f - u) * c0[c] + u * c1[c]) + (0.05f * w) * (c0[c] - 0.5f);
    const float Y = (0.2126f * v[0] + 0.7152f * v[1] + 0.0722f * v[2]) + eps;
    const float Yt = clip01f(0.22f + 0.55f * u);
    float gain = Yt / Y;
    gain = gain < 0.6f ? 0.6f : (gain > 1.6f ? 1.6f : gain);
#pragma unroll
    for (int c = 0; c < 3; ++c) { const float t = v[c] * gain; rgb[c] = clip01f(t / (1.0f + 0.6f * t)); }
}

__global__ __launch_bounds__(kT) void k_map_encode(MapArgs a) {
    __shared__ float thr[256];
    __shared__ uint8_t coarse[kCoarseTableBytes];
    for (int i = threadIdx.x; i < 256; i += kT) thr[i] = a.enc_thr[i];
    for (int i = threadIdx.x; i < kCoarseTableBytes; i += kT) coarse[i] = a.coarse[i];
    __syncthreads();
    const size_t n = a.n;
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += (size_t)gridDim.x * kT) {
        const float U = a.UBG[i], B = a.UBG[n + i], G = a.UBG[2 * n + i];
        float rgb[3];
        if (a.mode == 2) {  // map_opponent, uv_mappers.py:53-64 (sat/val/hsv in float64 like the reference)
            const float O1 = G - B, O2 = B - U;
            const float L = ((U + B) + G) / 3.0f;
            const float angle = atan2f(O2, O1);
            const float hue = (angle + 3.14159265358979323846f) / 6.28318530717958647692f;
            const float radius = __fsqrt_rn(O1 * O1 + O2 * O2);
            // np.percentile returns float32 here, so sat/val stay float32; in hsv_to_rgb `h*6.0 - i` mixes
            // float32 with int32, which NumPy promotes to float64: f, q, t are float64, p is float32.
            const float sat = clip01f(radius / ((float)a.pct[0] + a.eps));
            const float val = clip01f(L / ((float)a.pct[1] + a.eps));
            const float h6 = hue * 6.0f;
            const float fi = floorf(h6);
            const int ii = (int)fi;
            const double f = (double)h6 - (double)ii;
            const float p = val * (1.0f - sat);
            const float q = (float)((double)val * (1.0 - f * (double)sat));
            const float t = (float)((double)val * (1.0 - (1.0 - f) * (double)sat));
            const int im = ((ii % 6) + 6) % 6;
            rgb[0] = im == 0 ? val : im == 1 ? q : im == 2 ? p : im == 3 ? p : im == 4 ? t : val;
            rgb[1] = im == 0 ? t : im == 1 ? val : im == 2 ? val : im == 3 ? q : im == 4 ? p : p;
            rgb[2] = im == 0 ? p : im == 1 ? p : im == 2 ? t : im == 3 ? val : im == 4 ? val : q;
        } else if (a.mode == 0) {
            falsecolor(U, B, G, a.pct, a.eps, rgb);
        } else if (a.mode == 1) {  // map_linear_matrix: [U,B,G] @ M.T as an FMA chain
#pragma unroll
            for (int c = 0; c < 3; ++c) rgb[c] = fma_t(G, a.M[3 * c + 2], fma_t(B, a.M[3 * c + 1], U * a.M[3 * c]));
        } else if (a.mode == 3) {
            purple_yellow_soft(U, a.pct[0], 1e-8f, rgb);
        } else {  // mode 4: falsecolor_uv_mixed (uv_mappers.py:135-144), two passes around P99(mixed)
            float base[3], tint[3];
            falsecolor(U, B, G, a.pct, 1e-8f, base);
            purple_yellow_soft(U, a.pct[3], 1e-8f, tint);
            const float al = a.alpha;
#pragma unroll
            for (int c = 0; c < 3; ++c) rgb[c] = (1.0f - al) * base[c] + al * tint[c];
            if (a.pass == 1) {
#pragma unroll
                for (int c = 0; c < 3; ++c) a.mixed[(size_t)c * n + i] = rgb[c];
                continue;
            }
            const double p99 = a.pct[4];
            if (p99 > 1e-8) {
                const float d = (float)(p99 > 1.0 ? p99 : 1.0);
#pragma unroll
                for (int c = 0; c < 3; ++c) rgb[c] = rgb[c] / d;
            }
        }
        // honeybee.py:166-173: clip -> linear_to_srgb -> *255+0.5 -> uint8, as the threshold count
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (a.out_f) {  // float frames (honeybee.py:172-173): linear_to_srgb(clip(rgb_lin, 0, 1)).astype(dtype), uv_helpers.py:40-44
                const float l = rgb[c] < 0.f ? 0.f : (rgb[c] > 1.f ? 1.f : rgb[c]);
                a.out_f[i * 3 + c] = l <= 0.0031308f ? l * 12.92f : 1.055f * powf(l, 1.0f / 2.4f) - 0.055f;
            } else {
                a.out[i * 3 + c] = (uint8_t)quantize_coarse<float, kCoarseNFix>(rgb[c], thr, coarse, a.lo_key);
            }
        }
    }
}

// ---- analytic RGB -> HSI cube (a13), for the drop-in classic_rgb_to_hsi --------------------------------
__global__ __launch_bounds__(kT) void k_lobes_cube(const void* __restrict__ in, int in_u8, const float* __restrict__ lut_g, size_t n, int B,
                                                   const float* __restrict__ gains /*B x 3: gB,gG,gR*/, float denom, float* __restrict__ out) {
    extern __shared__ float g[];
    __shared__ float lut[256];
    for (int i = threadIdx.x; i < 3 * B; i += kT) g[i] = gains[i];
    for (int i = threadIdx.x; i < 256; i += kT) lut[i] = lut_g[i];
    __syncthreads();
    for (size_t p = (size_t)blockIdx.x * kT + threadIdx.x; p < n; p += (size_t)gridDim.x * kT) {
        float c0, c1, c2;
        if (in_u8) {
            const uint8_t* q = reinterpret_cast<const uint8_t*>(in) + p * 3;
            c0 = lut[q[0]]; c1 = lut[q[1]]; c2 = lut[q[2]];
        } else {  // float frame: srgb_to_linear applied as coded (Q6: also to already-linear inputs)
            const float* q = reinterpret_cast<const float*>(in) + p * 3;
            c0 = s2l(q[0]); c1 = s2l(q[1]); c2 = s2l(q[2]);
        }
        float* o = out + p * B;
        for (int b = 0; b < B; ++b) {
            // spec = gR*R + gG*G + gB*Bc (classic_rgb_to_hsi.py:70), then / (denom + 1e-8), clamp_min(0)
            const float s = (g[3 * b + 2] * c2 + g[3 * b + 1] * c1) + g[3 * b] * c0;
            o[b] = fmaxf(s / denom, 0.0f);
        }
    }
}


// ---- honeybee tail, recompute form ------------------------------------------------------------------------------------
// The passes above stream K float32 planes through HBM between every two stages (~100 B/px per frame for the default opponent
// mapping: catches out, blur in / out, radius + L out, three radix-select reads of both, map in).  Everything between the
// frame and its order statistics is cheap arithmetic on a 3 x 3 neighbourhood, so this form RECOMPUTES it in every pass from the
// uint8 frame (3 B/px) instead of storing it: one statistics pass (k_rgb_to_planes without an output), three radix-select
// passes whose "array" is the tile pipeline below, one map + encode pass -- 18 B/px, no plane ever written, and the frames of
// a batch share each launch (blockIdx.y = frame: five launches per STEP, not thirty per frame).
// The tile pipeline is the arithmetic of the plane kernels operation for operation (decode table, FMA-chain matrix, rescale,
// k_plane_blur_t<1>'s row / column passes with BORDER_REFLECT_101), so every pass sees bit-identical values and the result
// equals the plane route's.  SRC 1 reads raw catches from planes instead (an HSI cube's catches are not cheap to recompute).
struct BeeArgs {
    const uint8_t* in; const float* raw; uint8_t* out; float* out_f;
    int H, W; const float* lut; const float* mat; const float4* stats; int scale_mode; float t0, t1, t2;
    int mode, n_jobs, pass, shift, bits;
    SelState* st; uint32_t* hist; uint32_t* ticket; double* pct;          // per frame: kSelMax states, kSelMax x 2048 bins, 1 ticket, 8 doubles
    unsigned long long rank0; float gamma; int has_next;
    float eps; float M[9]; const float* enc_thr; const uint8_t* coarse; uint32_t lo_key;
    float* sel_planes;   // STAGE 0, pass 0: the values being selected are also written out, [frame][job][H*W] (passes 1-2 then read planes)
};

__device__ __forceinline__ void map_rgb(int mode, float U, float B, float G, const double* pct, float eps, const float* M, float (&rgb)[3]) {
    if (mode == 2) {  // map_opponent, uv_mappers.py:53-64 (see k_map_encode for the dtype notes)
        const float O1 = G - B, O2 = B - U;
        const float L = ((U + B) + G) / 3.0f;
        const float angle = atan2f(O2, O1);
        const float hue = (angle + 3.14159265358979323846f) / 6.28318530717958647692f;
        const float radius = __fsqrt_rn(O1 * O1 + O2 * O2);
        const float sat = clip01f(radius / ((float)pct[0] + eps));
        const float val = clip01f(L / ((float)pct[1] + eps));
        const float h6 = hue * 6.0f;
        const float fi = floorf(h6);
        const int ii = (int)fi;
        const double f = (double)h6 - (double)ii;
        const float p = val * (1.0f - sat);
        const float q = (float)((double)val * (1.0 - f * (double)sat));
        const float t = (float)((double)val * (1.0 - (1.0 - f) * (double)sat));
        const int im = ((ii % 6) + 6) % 6;
        rgb[0] = im == 0 ? val : im == 1 ? q : im == 2 ? p : im == 3 ? p : im == 4 ? t : val;
        rgb[1] = im == 0 ? t : im == 1 ? val : im == 2 ? val : im == 3 ? q : im == 4 ? p : p;
        rgb[2] = im == 0 ? p : im == 1 ? p : im == 2 ? t : im == 3 ? val : im == 4 ? val : q;
    } else if (mode == 0) {
        falsecolor(U, B, G, pct, eps, rgb);
    } else if (mode == 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = fma_t(G, M[3 * c + 2], fma_t(B, M[3 * c + 1], U * M[3 * c]));
    } else {
        purple_yellow_soft(U, pct[0], 1e-8f, rgb);
    }
}

// the last workgroup of a radix pass: pick the bin holding the rank, narrow the prefix, on the last pass resolve x[k], x[k+1] and
// NumPy's float32 lerp (the logic of k_sel_pass's tail, for one job)
// prefix0 / mask0: key bits every element is known to share before the first pass (0 / 0: none)
// AG: the state travels between workgroups INSIDE one launch (k_sel_all): every field is read and written with agent-scope atomics (the L2 slices of the XCDs are not
// coherent with each other for plain accesses before the kernel ends), and the histogram is left as it is (k_sel_all uses one per pass and clears them at its end).
__device__ __forceinline__ uint32_t ld_ag(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_ag(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_ag(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_ag(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <bool AG>
__device__ void sel_pick_t(uint32_t* hist, SelState* st, int pass, int shift, int bits, unsigned long long rank0, float gamma, int has_next, double* out,
                           uint32_t prefix0, uint32_t mask0) {
    __shared__ unsigned long long csum[kT];
    __shared__ int first_after[kT / 64];
    __shared__ int sel_chunk, sel_bin;
    __shared__ unsigned long long sel_rank;
    __shared__ uint32_t sel_cnt;
    const int nb = 1 << bits, t = threadIdx.x, per = nb / kT;
    const bool last_pass = pass == 2;
    const uint32_t prefix = pass == 0 ? prefix0 : (AG ? ld_ag(&st->prefix) : st->prefix), mask = pass == 0 ? mask0 : (AG ? ld_ag(&st->mask) : st->mask);
    uint32_t loc[8];
    unsigned long long sum = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) loc[i] = i < per ? __hip_atomic_load(&hist[t * per + (i < per ? i : 0)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += loc[i];
    // inclusive scan of the 256 chunk sums: within a wave by shuffles, across the four waves through LDS (two barriers; the
    // Hillis-Steele form in LDS took sixteen, and the pick is pure latency on the critical path of every pass)
    unsigned long long incl = sum;
    {
        const int lane = t & 63, wave = t >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        __syncthreads();  // csum is free (a previous job's readers are done)
        if (lane == 63) csum[wave] = incl;
        if (t == 0) sel_chunk = kT - 1;
        __syncthreads();
        for (int w = 0; w < wave; ++w) incl += csum[w];
    }
    const unsigned long long r = pass == 0 ? rank0 : (AG ? ld_ag(&st->rank) : st->rank);
    const unsigned long long excl = incl - sum;
    if (excl <= r && r < incl) sel_chunk = t;
    __syncthreads();
    if (t == sel_chunk) {
        unsigned long long cum = excl;
        int i = 0;
        for (; i < per; ++i) {
            if (cum + loc[i] > r) break;
            cum += loc[i];
        }
        if (i == per) i = per - 1;
        const int b = t * per + i;
        if constexpr (AG) {
            st_ag(&st->prefix, prefix | ((uint32_t)b << shift));
            st_ag(&st->mask, mask | ((uint32_t)(nb - 1) << shift));
            st_ag(&st->rank, r - cum);
            if (pass == 0) st_ag(&st->next_above, 0xffffffffu);
        } else {
            st->prefix = prefix | ((uint32_t)b << shift);
            st->mask = mask | ((uint32_t)(nb - 1) << shift);
            st->rank = r - cum;
            if (pass == 0) st->next_above = 0xffffffffu;
        }
        if (last_pass) { st->key_lo = prefix | (uint32_t)b; st->cnt_in_bin = loc[i]; sel_bin = b; sel_rank = r - cum; sel_cnt = loc[i]; }
    }
    if (last_pass) {
        __syncthreads();
        int f = 0x7fffffff;
        for (int i = per - 1; i >= 0; --i)
            if (loc[i] && t * per + i > sel_bin) f = t * per + i;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(f, o); f = v < f ? v : f; }
        if ((t & 63) == 0) first_after[t >> 6] = f;
        __syncthreads();
        if (t == 0) {
            int m = first_after[0];
            for (int w = 1; w < kT / 64; ++w) m = first_after[w] < m ? first_after[w] : m;
            const uint32_t next_above = __hip_atomic_load(&st->next_above, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t next_key = m != 0x7fffffff ? (prefix | (uint32_t)m) : next_above;
            const uint32_t key_lo = prefix | (uint32_t)sel_bin;
            const float lo = key2f(key_lo);
            float hi = lo;
            if (has_next && sel_rank + 1 >= sel_cnt) hi = next_key == 0xffffffffu ? lo : key2f(next_key);
            const float diff = hi - lo;
            float res = lo + diff * gamma;
            if (gamma >= 0.5f) res = hi - diff * (1.0f - gamma);
            *out = (double)res;
        }
    }
    if constexpr (!AG)
        for (int i = 0; i < per; ++i) hist[t * per + i] = 0;
    __syncthreads();
}

// One radix pass over the values of a band stack that only exists at reduced size (stack_up.h): every thread recomputes the K
// resized + normalised values of its pixels and histograms them; flush, ticket and pick as in k_sel_pass (one job).
// cand / cand_cnt (round 3): the SECOND pass also copies every value of the bin the first pass selected into cand[] (count in *cand_cnt) and tracks the smallest key
// above that bin, so the THIRD pass scans those candidates (k_sel_cand) instead of recomputing the whole stack a third time.
struct SelUpArgs { StackUp u; size_t cap_floats; uint32_t prefix0, mask0; int pass, shift, bits; SelState* st; uint32_t* hist; uint32_t* ticket; unsigned long long rank0; float gamma; int has_next; double* out;
                   float* cand; uint32_t* cand_cnt; };
constexpr uint32_t kSelStage = 3072;  // staged candidates per tile (12 KB of LDS); denser tiles take a second sweep
template <int K>
__global__ __launch_bounds__(kT) void k_sel_pass_up(const SelUpArgs a) {
    __shared__ uint32_t h[2048];
    __shared__ uint32_t wmin[kT / 64];
    __shared__ int is_last;
    __shared__ uint32_t scnt, sbase;
    __shared__ int redo;  // collect: 1 during the second sweep of a tile whose candidates outgrew the staging area (values are then written straight to cand[])
    const int nb = 1 << a.bits, t = threadIdx.x;
    const bool last_pass = a.pass == 2, collect = a.pass == 1 && a.cand != nullptr;
    const bool find_next = (last_pass || collect) && a.has_next;  // collect: the smallest key above the first pass's bin (the successor when the candidates hold none)
    for (int i = t; i < 2048; i += kT) h[i] = 0;
    if (t == 0) { scnt = 0; redo = 0; }
    const uint32_t prefix = a.pass == 0 ? a.prefix0 : a.st->prefix, mask = a.pass == 0 ? a.mask0 : a.st->mask;
    const uint32_t above = prefix | ~mask;
    uint32_t best = 0xffffffffu;
    __syncthreads();
    extern __shared__ float tile_lds[];
    float* stage = tile_lds + a.cap_floats;  // collect: this tile's candidates, up to kSelStage of them
    stack_tiles<K, true>(a.u, tile_lds, a.cap_floats, [&](int, int, float (&v)[K]) {
        const bool second = collect && redo;  // uniform
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t key = f2key(v[k]);
            if ((key & mask) == prefix) {
                if (!second) atomicAdd(&h[(key >> a.shift) & (nb - 1)], 1u);
                if (collect) {
                    const uint32_t idx = atomicAdd(&scnt, 1u);
                    if (second) a.cand[(size_t)sbase + idx] = v[k];
                    else if (idx < kSelStage) stage[idx] = v[k];
                }
            } else if (find_next && key > above && key < best) best = key;
        }
    }, [&]() -> bool {
        if (!collect) return false;  // uniform
        const uint32_t m = scnt;
        const bool was_second = redo != 0, dense = !was_second && m > kSelStage;
        __syncthreads();  // everyone has read scnt / redo
        if (was_second) {
            if (t == 0) { scnt = 0; redo = 0; }
            __syncthreads();
            return false;
        }
        if (t == 0) {
            if (m) sbase = atomicAdd(a.cand_cnt, m);  // one reservation per tile
            scnt = 0;
            redo = dense ? 1 : 0;
        }
        __syncthreads();
        if (dense) return true;  // too many for the staging area (a flat region: every value in the bin): the tile is swept again and writes straight to cand[sbase + ...]
        for (uint32_t i = t; i < m; i += kT) a.cand[(size_t)sbase + i] = stage[i];
        __syncthreads();  // the staging area is free again
        return false;
    });
    if (find_next) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t w = __shfl_xor(best, o); best = w < best ? w : best; }
        __syncthreads();
        if ((t & 63) == 0) wmin[t >> 6] = best;
        __syncthreads();
        if (t == 0) {
            for (int w = 1; w < kT / 64; ++w) best = wmin[w] < best ? wmin[w] : best;
            if (best != 0xffffffffu) atomicMin(&a.st->next_above, best);
        }
    }
    __syncthreads();
    for (int i = t; i < nb; i += kT)
        if (h[i]) atomicAdd(&a.hist[i], h[i]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) is_last = avxk::ticket_is_last(a.ticket, blockIdx.x, gridDim.x);  // two-level: up to 1,024 workgroups arrive here
    __syncthreads();
    if (!is_last) return;
    sel_pick(a.hist, a.st, a.pass, a.shift, a.bits, a.rank0, a.gamma, a.has_next, a.out, a.prefix0, a.mask0);
    if (a.pass == 0 && a.cand_cnt && t == 0) *a.cand_cnt = 0;  // the next launch collects from zero (the kernel boundary publishes it)
}

// The third radix pass over the candidates the second one collected (k_sel_pass_up): *cand_cnt values, all inside the first pass's bin.  Same histogram, successor
// search, flush, ticket and pick as the full pass; the successor found among keys above that bin is already in st->next_above.
__global__ __launch_bounds__(kT) void k_sel_cand(const SelUpArgs a) {
    __shared__ uint32_t h[2048];
    __shared__ uint32_t wmin[kT / 64];
    __shared__ int is_last;
    const int nb = 1 << a.bits, t = threadIdx.x;
    const bool find_next = a.has_next != 0;
    for (int i = t; i < 2048; i += kT) h[i] = 0;
    const uint32_t prefix = a.st->prefix, mask = a.st->mask, above = prefix | ~mask;
    const size_t n = *a.cand_cnt;
    uint32_t best = 0xffffffffu;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * kT + t; i < n; i += (size_t)gridDim.x * kT) {
        const uint32_t key = f2key(a.cand[i]);
        if ((key & mask) == prefix) atomicAdd(&h[(key >> a.shift) & (nb - 1)], 1u);
        else if (find_next && key > above && key < best) best = key;
    }
    if (find_next) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t w = __shfl_xor(best, o); best = w < best ? w : best; }
        __syncthreads();
        if ((t & 63) == 0) wmin[t >> 6] = best;
        __syncthreads();
        if (t == 0) {
            for (int w = 1; w < kT / 64; ++w) best = wmin[w] < best ? wmin[w] : best;
            if (best != 0xffffffffu) atomicMin(&a.st->next_above, best);
        }
    }
    __syncthreads();
    for (int i = t; i < nb; i += kT)
        if (h[i]) atomicAdd(&a.hist[i], h[i]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) is_last = avxk::ticket_is_last(a.ticket, blockIdx.x, gridDim.x);
    __syncthreads();
    if (!is_last) return;
    sel_pick(a.hist, a.st, 2, a.shift, a.bits, a.rank0, a.gamma, a.has_next, a.out, a.prefix0, a.mask0);
}

// SRC: 0 uint8 frames (decode table + K x 3 matrix), 1 raw catch planes.  R: blur radius 0 | 1.  STAGE: 0 one radix-select pass
// (NJ order statistics per frame), 1 map + encode.
template <int SRC, int R, int STAGE, int NJ>
__global__ __launch_bounds__(kT) void k_bee_tile(const BeeArgs a) {
    constexpr int TW = 32, TH = 32, AH = TH + 2 * R, AW = TW + 2 * R, AWP = (AW + 3) & ~3;
    __shared__ __align__(16) float A[3][AH * AWP];
    __shared__ __align__(16) float Bm[R ? 3 : 1][R ? AH * TW : 1];
    __shared__ float lut[256];
    __shared__ uint32_t h[STAGE == 0 ? NJ : 1][STAGE == 0 ? 2048 : 1];
    __shared__ float thr[STAGE == 1 ? 256 : 1];
    __shared__ uint8_t coarse[STAGE == 1 ? kCoarseTableBytes : 1];
    __shared__ uint32_t wmin[kT / 64];
    __shared__ int is_last;
    const int t = threadIdx.x, f = blockIdx.y;
    const size_t n = (size_t)a.H * a.W;
    if (SRC == 0)
        for (int i = t; i < 256; i += kT) lut[i] = a.lut[i];
    if (STAGE == 0)
        for (int i = t; i < NJ * 2048; i += kT) (&h[0][0])[i] = 0;
    if (STAGE == 1) {
        for (int i = t; i < 256; i += kT) thr[i] = a.enc_thr[i];
        for (int i = t; i < kCoarseTableBytes; i += kT) coarse[i] = a.coarse[i];
    }
    float m[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] = SRC == 0 ? a.mat[i] : 0.f;
    float4 st[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) st[k] = a.scale_mode ? a.stats[(size_t)f * 4 + k] : make_float4(0.f, 0.f, 0.f, 1.f);
    const uint8_t* in = a.in + (size_t)f * n * 3;
    const float* raw = a.raw + (size_t)f * n * 3;
    const double* pct = a.pct + (size_t)f * 8;
    SelState* sst = a.st + (size_t)f * kSelMax;
    const bool last_pass = a.pass == 2, find_next = STAGE == 0 && last_pass && a.has_next;
    uint32_t prefix[NJ], mask[NJ], above[NJ], best[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        prefix[j] = (STAGE == 0 && a.pass != 0) ? sst[j].prefix : 0u;
        mask[j] = (STAGE == 0 && a.pass != 0) ? sst[j].mask : 0u;
        above[j] = prefix[j] | ~mask[j];
        best[j] = 0xffffffffu;
    }
    const int nb = 1 << a.bits;
    auto visit = [&](int j, float v) {
        const uint32_t k = f2key(v);
        if ((k & mask[j]) == prefix[j]) atomicAdd(&h[STAGE == 0 ? j : 0][(k >> a.shift) & (nb - 1)], 1u);
        else if (find_next && k > above[j] && k < best[j]) best[j] = k;
    };
    __syncthreads();
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    // catches of a tile and its halo, rescaled (reflect-101 at the frame border).  A thread's loads for the NEXT tile are issued
    // before the current tile's passes and consumed after them: with one pixel per trip and no prefetch the loop waited out a memory
    // round trip per pixel (6.6 us per tile).
    constexpr int NIT = (AH * AW + kT - 1) / kT;
    float src[NIT][3];
    auto fetch = [&](int tile) {
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x, x0 = tx * TW, y0 = ty * TH;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = t + it * kT, ii = i < AH * AW ? i : AH * AW - 1;
            const int ly = ii / AW, lx = ii - ly * AW;
            const int gy = reflect101(y0 - R + ly, a.H), gx = reflect101(x0 - R + lx, a.W);
            const size_t px = (size_t)gy * a.W + gx;
            if (SRC == 0) {
                const uint8_t* q = in + px * 3;
                src[it][0] = __uint_as_float((uint32_t)q[0]); src[it][1] = __uint_as_float((uint32_t)q[1]); src[it][2] = __uint_as_float((uint32_t)q[2]);
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) src[it][k] = raw[(size_t)k * n + px];
            }
        }
    };
    if ((int)blockIdx.x < tiles_x * tiles_y) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < tiles_x * tiles_y; tile += gridDim.x) {
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x, x0 = tx * TW, y0 = ty * TH;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = t + it * kT;
            if (i >= AH * AW) break;
            const int ly = i / AW, lx = i - ly * AW;
            float v[3];
            if (SRC == 0) {
                const float c0 = lut[__float_as_uint(src[it][0])], c1 = lut[__float_as_uint(src[it][1])], c2 = lut[__float_as_uint(src[it][2])];
#pragma unroll
                for (int k = 0; k < 3; ++k) v[k] = fma_t(c2, m[3 * k + 2], fma_t(c1, m[3 * k + 1], c0 * m[3 * k]));
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) v[k] = src[it][k];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) A[k][ly * AWP + lx] = rescale(v[k], a.scale_mode, st[k]);
        }
        if (tile + (int)gridDim.x < tiles_x * tiles_y) fetch(tile + gridDim.x);
        __syncthreads();
        if (R) {  // row pass of k_plane_blur_t<1>: s = w[x] t0; s = fma(w[x + 1], t1, s); s = fma(w[x + 2], t2, s)
            for (int i = t; i < 3 * AH * (TW / 4); i += kT) {
                const int k = i / (AH * (TW / 4)), r2 = i - k * (AH * (TW / 4)), ly = r2 / (TW / 4), g = r2 - ly * (TW / 4);
                const float* wp = &A[k][ly * AWP + 4 * g];
                float w[8];
#pragma unroll
                for (int q = 0; q < 6; ++q) w[q] = wp[q];
                float o[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) o[x] = fma_t(w[x + 2], a.t2, fma_t(w[x + 1], a.t1, w[x] * a.t0));
                *reinterpret_cast<float4*>(&Bm[R ? k : 0][R ? ly * TW + 4 * g : 0]) = make_float4(o[0], o[1], o[2], o[3]);
            }
            __syncthreads();
        }
        {  // column pass (centre tap, then fma(x[+1] + x[-1], t2, s)) and this pass's use of (U, B, G): thread = column x, rows 4 yq .. + 3
            const int x = t & 31, yq = t >> 5;
            float w[3][6];
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int q = 0; q < (R ? 6 : 4); ++q) w[k][q] = R ? Bm[R ? k : 0][(4 * yq + q) * TW + x] : A[k][(4 * yq + q) * AWP + x];
#pragma unroll
            for (int yy = 0; yy < 4; ++yy) {
                const int y = 4 * yq + yy;
                if (y0 + y >= a.H || x0 + x >= a.W) continue;
                float ubg[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) ubg[k] = R ? fma_t(w[k][yy + 2] + w[k][yy], a.t2, w[k][yy + 1] * a.t1) : w[k][yy];
                const float U = ubg[0], B = ubg[1], G = ubg[2];
                if (STAGE == 0) {
                    float sv[3] = {U, B, G};
                    if (a.mode == 2) {
                        const float O1 = G - B, O2 = B - U;
                        sv[0] = __fsqrt_rn(O1 * O1 + O2 * O2);
                        sv[1] = ((U + B) + G) / 3.0f;
                    }
#pragma unroll
                    for (int j = 0; j < NJ; ++j) visit(j, sv[j]);
                    if (a.sel_planes) {
                        const size_t px = (size_t)(y0 + y) * a.W + x0 + x;
#pragma unroll
                        for (int j = 0; j < NJ; ++j) a.sel_planes[((size_t)f * NJ + j) * n + px] = sv[j];
                    }
                } else {
                    float rgb[3];
                    map_rgb(a.mode, U, B, G, pct, a.eps, a.M, rgb);
                    const size_t px = (size_t)(y0 + y) * a.W + x0 + x;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        if (a.out_f) {
                            const float l = rgb[c] < 0.f ? 0.f : (rgb[c] > 1.f ? 1.f : rgb[c]);
                            a.out_f[((size_t)f * n + px) * 3 + c] = l <= 0.0031308f ? l * 12.92f : 1.055f * powf(l, 1.0f / 2.4f) - 0.055f;
                        } else {
                            a.out[((size_t)f * n + px) * 3 + c] = (uint8_t)quantize_coarse<float, kCoarseNFix>(rgb[c], thr, coarse, a.lo_key);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    if (STAGE == 1) return;
    // ---- end of a radix pass: successor candidates, histogram flush, ticket, and the last workgroup of this FRAME picks ----
    uint32_t* hist = a.hist + (size_t)f * kSelMax * 2048;
    if (find_next) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            uint32_t b = best[j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const uint32_t v = __shfl_xor(b, o); b = v < b ? v : b; }
            __syncthreads();
            if ((t & 63) == 0) wmin[t >> 6] = b;
            __syncthreads();
            if (t == 0) {
                for (int w = 1; w < kT / 64; ++w) b = wmin[w] < b ? wmin[w] : b;
                if (b != 0xffffffffu) atomicMin(&sst[j].next_above, b);
            }
        }
    }
    __syncthreads();
    for (int j = 0; j < NJ; ++j)
        for (int i = t; i < nb; i += kT)
            if (h[STAGE == 0 ? j : 0][i]) atomicAdd(&hist[j * 2048 + i], h[STAGE == 0 ? j : 0][i]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) is_last = atomicAdd(a.ticket + f, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!is_last) return;
    for (int j = 0; j < NJ; ++j) sel_pick(hist + j * 2048, sst + j, a.pass, a.shift, a.bits, a.rank0, a.gamma, a.has_next, a.pct + (size_t)f * 8 + j);
    if (t == 0) a.ticket[f] = 0;
}

int grid_for(avx_ctx* ctx, size_t items) {
    const size_t want = (items + kT - 1) / kT;
    const size_t cap = (size_t)ctx->num_cus * 8;
    return (int)(want < cap ? (want ? want : 1) : cap);
}

}  // namespace

// ======================================= C ABI =======================================================
struct UvScratch {
    Stat3* partials; float4* stats; uint32_t* hist; SelState* sel; uint32_t* ticket; double* pct; float* mat;
};

// ---- the three radix passes in ONE launch (round 3) ----------------------------------------------------------------------------------------
// A percentile used to be three dependent launches of ~16-40 us each for a 1080p plane, almost all of it fixed cost (launch, re-reading the plane, the drain before
// the next pass).  Here the grid stays resident: a workgroup keeps its share of every job in REGISTERS (8 x 16 bytes per thread and job: a whole 1080p plane at one
// workgroup per CU; what does not fit is re-read), histograms pass p, flushes, takes a ticket; the LAST workgroup picks, then releases the others through a
// generation counter they poll (agent-scope atomic loads + s_sleep).  The grid is capped at one workgroup per CU (256 threads, 32 KB of LDS at NJ = 4): every
// workgroup of ONE such launch is resident at once; launches from several streams can starve each other (each gets part of its grid and waits for the rest), which is
// why this form is opt-in (run_percentiles); a wait that outlasts ~0.3 s gives up (the launch then returns without a result rather than hanging the GPU).  Measured (hummingbird 1080p, phases ablated): 50 us for one plane = visits 14.5 (the run-length LDS atomics), flush 11.6, picks 5.8,
// the bare ticket / release / poll chain 23 (~ eight dependent agent-scope round trips per pass); the three-launch form took 54.  One global histogram per pass (cleared by the last workgroup at the end: the kernel boundary publishes the zeros).
struct SelAll { int n_jobs; SelState* st; uint32_t* hist /*[3][kSelMax][2048]*/; uint32_t* ticket; uint32_t* gen; SelJob job[kSelMax]; int shift[3], bits[3]; };
template <int NJ>
__global__ __launch_bounds__(kT) void k_sel_all(const SelAll a) {
    __shared__ uint32_t h[NJ][2048];
    __shared__ uint32_t wmin[kT / 64];
    __shared__ int is_last, gave_up;
    const int t = threadIdx.x;
    const uint32_t gen0 = ld_ag(a.gen);  // read before this workgroup's first ticket: nobody can have advanced it yet
    const size_t stride = (size_t)gridDim.x * kT, i0 = (size_t)blockIdx.x * kT + t;
    auto job_vec = [&](int j, const float4*& xv, size_t& nvec, size_t& head, size_t& tail0) {
        const SelJob jb = a.job[j];
        const size_t mis = ((16 - ((uintptr_t)jb.x & 15)) & 15) / 4;
        head = mis < jb.n ? mis : jb.n;
        nvec = (jb.n - head) / 4; tail0 = head + nvec * 4;
        xv = reinterpret_cast<const float4*>(jb.x + head);
    };
    float4 keep[NJ][8];  // this thread's first eight vectors of every job: loaded once, visited in all three passes
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (j >= a.n_jobs) break;
        const float4* xv; size_t nvec, head, tail0;
        job_vec(j, xv, nvec, head, tail0);
#pragma unroll
        for (int q = 0; q < 8; ++q) { const size_t idx = i0 + q * stride; keep[j][q] = nvec ? xv[idx < nvec ? idx : nvec - 1] : make_float4(0.f, 0.f, 0.f, 0.f); }
    }
    if (t == 0) gave_up = 0;
#pragma unroll 1
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = a.shift[pass], nb = 1 << a.bits[pass];
        const bool last_pass = pass == 2;
        for (int i = t; i < NJ * 2048; i += kT) (&h[0][0])[i] = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (j >= a.n_jobs) break;
            const SelJob jb = a.job[j];
            SelState* st = a.st + j;
            const bool find_next = last_pass && jb.has_next;
            const uint32_t prefix = pass == 0 ? 0u : ld_ag(&st->prefix), mask = pass == 0 ? 0u : ld_ag(&st->mask);
            const uint32_t above = prefix | ~mask;
            uint32_t best = 0xffffffffu, run_bin = 0xffffffffu, run_cnt = 0;
            auto visit = [&](float f) {  // as k_sel_pass: run-length combined LDS atomics
                const uint32_t k = f2key(f);
                if ((k & mask) == prefix) {
                    const uint32_t b = (k >> shift) & (nb - 1);
                    if (b == run_bin) { ++run_cnt; }
                    else {
                        if (run_cnt) atomicAdd(&h[j][run_bin], run_cnt);
                        run_bin = b; run_cnt = 1;
                    }
                } else if (find_next && k > above && k < best) best = k;
            };
            const float4* xv; size_t nvec, head, tail0;
            job_vec(j, xv, nvec, head, tail0);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (i0 + q * stride < nvec) { visit(keep[j][q].x); visit(keep[j][q].y); visit(keep[j][q].z); visit(keep[j][q].w); }
            for (size_t i = i0 + 8 * stride; i < nvec; i += 8 * stride) {  // planes beyond 8 vectors per thread: re-read
                float4 v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { const size_t idx = i + q * stride; v[q] = xv[idx < nvec ? idx : nvec - 1]; }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (i + q * stride < nvec) { visit(v[q].x); visit(v[q].y); visit(v[q].z); visit(v[q].w); }
            }
            if (blockIdx.x == 0) {
                if ((size_t)t < head) visit(jb.x[t]);
                if (tail0 + t < jb.n && t < 4) visit(jb.x[tail0 + t]);
            }
            if (run_cnt) atomicAdd(&h[j][run_bin], run_cnt);
            if (find_next) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { const uint32_t v = __shfl_xor(best, o); best = v < best ? v : best; }
                __syncthreads();
                if ((t & 63) == 0) wmin[t >> 6] = best;
                __syncthreads();
                if (t == 0) {
                    for (int w = 1; w < kT / 64; ++w) best = wmin[w] < best ? wmin[w] : best;
                    if (best != 0xffffffffu) __hip_atomic_fetch_min(&st->next_above, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        __syncthreads();
        uint32_t* ghist = a.hist + (size_t)pass * kSelMax * 2048;
        for (int j = 0; j < a.n_jobs; ++j)
            for (int i = t; i < nb; i += kT)
                if (h[j][i]) __hip_atomic_fetch_add(&ghist[j * 2048 + i], h[j][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) is_last = avxk::ticket_is_last(a.ticket, blockIdx.x, gridDim.x, true);
        __syncthreads();
        if (is_last) {
            for (int j = 0; j < a.n_jobs; ++j) {
                const SelJob jb = a.job[j];
                sel_pick_t<true>(ghist + j * 2048, a.st + j, pass, shift, a.bits[pass], jb.rank0, jb.gamma, jb.has_next, jb.out, 0u, 0u);
            }
            if (last_pass)  // leave every histogram of the launch zero for the next one (plain stores: the kernel boundary publishes them)
                for (int i = t; i < 3 * kSelMax * 2048; i += kT) a.hist[i] = 0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ticket's own resets (ticket_is_last) and the state have landed before anyone is released
                __hip_atomic_fetch_add(a.gen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (!last_pass) {
            if (t == 0) {
                int spins = 0;
                while ((int)(ld_ag(a.gen) - (gen0 + (uint32_t)pass + 1u)) < 0) {
                    __builtin_amdgcn_s_sleep(16);
                    if (++spins > (1 << 20)) { gave_up = 1; break; }
                }
            }
            __syncthreads();
            if (gave_up) return;
        }
    }
}

static int uv_small_scratch(avx_ctx* ctx, hipStream_t stream, UvScratch* s, avx_ws** out_ws = nullptr) {
    avx_ws* ws = avx_workspace(ctx, stream);
    if (!ws) return AVX_ERR_NOMEM;
    if (out_ws) *out_ws = ws;
    const size_t need = (size_t)ctx->num_cus * 8 * 16 * sizeof(Stat3) + 16 * sizeof(float4) + 3 * kSelMax * 2048 * 4 + kSelMax * sizeof(SelState) + 64 + 16 * 8 + 16 * 129 * 4 + 4096;
    if (ws->uv_small == nullptr) {
        AVX_HIP(ctx, hipMalloc(&ws->uv_small, need));
        AVX_HIP(ctx, hipMemsetAsync(ws->uv_small, 0, need, stream));  // histogram and ticket start at zero; k_sel_pass leaves them so
    }
    char* p = (char*)ws->uv_small;
    s->partials = (Stat3*)p; p += (size_t)ctx->num_cus * 8 * 16 * sizeof(Stat3);
    s->stats = (float4*)p; p += 16 * sizeof(float4);
    s->hist = (uint32_t*)p; p += 3 * kSelMax * 2048 * 4;  // one per pass for k_sel_all; the per-pass kernels use the first
    s->sel = (SelState*)p; p += kSelMax * sizeof(SelState) + 8;
    s->ticket = (uint32_t*)p; p += 256;  // ticket_is_last's 1 + 32 counters; word 48: k_sel_all's generation counter
    p = (char*)(((uintptr_t)p + 15) & ~(uintptr_t)15);
    s->pct = (double*)p; p += 16 * 8;
    s->mat = (float*)p;
    return AVX_OK;
}

// exact np.percentile(x_j, q_j) (linear interpolation) of n_j device floats -> *out_j (device doubles), up to kSelMax
// order statistics resolved by the same three passes
struct PctReq { const float* x; size_t n; double q; double* out_dev; };
static int run_percentiles(avx_ctx* ctx, const UvScratch& u, const PctReq* req, int count, hipStream_t s) {
    for (int base = 0; base < count; base += kSelMax) {
        const int nj = count - base < kSelMax ? count - base : kSelMax;
        SelPass a{};
        a.n_jobs = nj; a.st = u.sel; a.hist = u.hist; a.ticket = u.ticket;
        size_t nmax = 0;
        for (int j = 0; j < nj; ++j) {
            const PctReq& r = req[base + j];
            // NumPy (2.x) evaluates the virtual index in the array's dtype: float32(n-1) * (float32(q)/float32(100)).
            const float vi = (float)(r.n - 1) * ((float)r.q / 100.0f);
            float lo = floorf(vi);
            if (lo < 0) lo = 0;
            if (lo > (float)(r.n - 1)) lo = (float)(r.n - 1);
            a.job[j] = SelJob{r.x, r.n, (unsigned long long)lo, vi - lo, (size_t)lo + 1 < r.n ? 1 : 0, r.out_dev};
            nmax = r.n > nmax ? r.n : nmax;
        }
        // few, fat workgroups: every workgroup flushes its non-empty LDS bins with global atomics
        const size_t want = (nmax + (size_t)kT * 16 - 1) / ((size_t)kT * 16);
        static const int wg_per_cu = [] { const char* e = getenv("AVX_SEL_WG"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : (v > 8 ? 8 : v); }();
        const size_t cap = (size_t)ctx->num_cus * wg_per_cu;  // measured again in round 2 (hummingbird 1080p: 3.24 / 3.02 / 2.49 / 2.07 GP/s at 1 / 2 / 4 / 8 per CU): every workgroup pays the histogram clear, flush and ticket
        const int g = (int)(want < cap ? (want ? want : 1) : cap);
        const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
        // The one-launch form (k_sel_all) is OPT-IN (AVX_SEL_ONE=1; read per call: tests flip it).  Its inner barriers need the whole grid resident, and that holds for ONE
        // instance only: at 128+ registers per lane a CU takes two of its workgroups, so when several streams run selects at once (the four frame lanes of bench.py's
        // UV legs did) the dispatcher can give each launch HALF of its workgroups and all of them wait for the other halves until the time-out fires (measured: the
        // default bench's hummingbird leg at 326 ms per step, its selects giving up).  With one stream it is 14 us per hummingbird frame faster than three launches.
        const char* one_env = getenv("AVX_SEL_ONE");
        if (one_env && one_env[0] == '1') {
            SelAll b{};
            b.n_jobs = nj; b.st = u.sel; b.hist = u.hist; b.ticket = u.ticket; b.gen = u.ticket + 48;
            for (int j = 0; j < nj; ++j) b.job[j] = a.job[j];
            for (int p = 0; p < 3; ++p) { b.shift[p] = shifts[p]; b.bits[p] = bits[p]; }
            const int g1 = (int)(want < (size_t)ctx->num_cus ? (want ? want : 1) : (size_t)ctx->num_cus);
            if (nj == 1) hipLaunchKernelGGL(k_sel_all<1>, dim3(g1), dim3(kT), 0, s, b);
            else if (nj == 2) hipLaunchKernelGGL(k_sel_all<2>, dim3(g1), dim3(kT), 0, s, b);
            else hipLaunchKernelGGL(k_sel_all<4>, dim3(g1), dim3(kT), 0, s, b);
            continue;
        }
        for (int p = 0; p < 3; ++p) {
            a.pass = p; a.shift = shifts[p]; a.bits = bits[p];
            if (nj == 1) hipLaunchKernelGGL(k_sel_pass<1>, dim3(g), dim3(kT), 0, s, a);
            else if (nj == 2) hipLaunchKernelGGL(k_sel_pass<2>, dim3(g), dim3(kT), 0, s, a);
            else hipLaunchKernelGGL(k_sel_pass<4>, dim3(g), dim3(kT), 0, s, a);
        }
    }
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}
static int run_percentile(avx_ctx* ctx, const UvScratch& u, const float* x, size_t n, double q, double* out_dev, hipStream_t s) {
    const PctReq r{x, n, q, out_dev};
    return run_percentiles(ctx, u, &r, 1, s);
}

// ---- internal entry points for other translation units of the library (mantis.hip) ---------------------
int avx_uv_percentile_device(avx_ctx* ctx, const float* x, size_t n, double q, double* out_dev, hipStream_t s) {
    UvScratch u;
    int rc = uv_small_scratch(ctx, s, &u);
    if (rc) return rc;
    return run_percentile(ctx, u, x, n, q, out_dev, s);
}

// np.percentile(q) over the H x W x K values of a reduced-size band stack read through its resize (+ safe_norm): stack_up.h
int avx_uv_percentile_up_device(avx_ctx* ctx, const StackUp& up, double q, double* out_dev, hipStream_t s, float* cand_buf /*H * W * K floats, or NULL: three full passes*/) {
    UvScratch u;
    int rc = uv_small_scratch(ctx, s, &u);
    if (rc) return rc;
    const size_t n = (size_t)up.H * up.W * up.K;
    const float vi = (float)(n - 1) * ((float)q / 100.0f);  // as run_percentiles
    float lo = floorf(vi);
    if (lo < 0) lo = 0;
    if (lo > (float)(n - 1)) lo = (float)(n - 1);
    SelUpArgs a{};
    a.u = up; a.st = u.sel; a.hist = u.hist; a.ticket = u.ticket; a.rank0 = (unsigned long long)lo; a.gamma = vi - lo; a.has_next = (size_t)lo + 1 < n ? 1 : 0; a.out = out_dev;
    static_assert(kT == 256, "AVX_STACK_TILES assumes 256 threads");
    const size_t tiles = (size_t)((up.W + kUpTW - 1) / kUpTW) * ((up.H + kUpTH - 1) / kUpTH), cap = (size_t)ctx->num_cus * 4;
    const int g = (int)(tiles < cap ? (tiles ? tiles : 1) : cap);
    a.cap_floats = stack_tile_floats(up.hs, up.ws, up.H, up.W, up.K);
    const size_t lds_tile = a.cap_floats * sizeof(float);
    AVX_REQUIRE(ctx, lds_tile <= 96 * 1024, "percentile through a resized stack: the tile's source rectangle does not fit LDS (%zu bytes)", lds_tile);
    // candidates: the second pass stages a tile's matching values in LDS (kSelStage floats behind the source rectangle) and copies them out once per tile
    const size_t lds_stage = (size_t)kSelStage * sizeof(float);
    const char* cand_env = getenv("AVX_MANTIS_CAND");  // read per call: tests flip it (0: three full passes)
    const bool use_cand = cand_buf && !(cand_env && cand_env[0] == '0') && lds_tile + lds_stage <= 120 * 1024;
    const size_t lds = lds_tile + (use_cand ? lds_stage : 0);
    AVX_REQUIRE(ctx, stack_k_tiled(up.K) && up.mm, "percentile through a resized stack: K=%d is not instantiated / no min-max table", up.K);
    AVX_STACK_K_SWITCH(up.K, AVX_HIP(ctx, hipFuncSetAttribute((const void*)k_sel_pass_up<KT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)))
    // safe_norm's values lie in [0, 1]: every key starts with the bits 10 (positive, exponent < 128), so the three digits are taken from
    // bits 29..19, 18..8, 7..0 -- the first histogram then resolves 16 mantissa steps per binade instead of 4, and its LDS atomics
    // collide a quarter as often (the plain 11/11/10 split put an eighth of a uniform [0, 1] sample into ONE bin)
    a.prefix0 = 0x80000000u; a.mask0 = 0xc0000000u;
    const int shifts[3] = {19, 8, 0}, bits[3] = {11, 11, 8};
    if (use_cand) { a.cand = cand_buf; a.cand_cnt = u.ticket + 56; }  // word 56 of the ticket block: zero at rest (uv_small_scratch), re-zeroed by the first pass
    for (int p = 0; p < 3; ++p) {
        a.pass = p; a.shift = shifts[p]; a.bits = bits[p];
        if (p == 2 && use_cand) {  // the candidates of the first pass's bin instead of the whole stack
            const int gc = (int)((size_t)ctx->num_cus < (size_t)g ? (size_t)ctx->num_cus : (size_t)g);
            hipLaunchKernelGGL(k_sel_cand, dim3(gc), dim3(kT), 0, s, a);
            break;
        }
        const size_t lds_p = (p == 1 && use_cand) ? lds : lds_tile;  // only the collecting pass carries the staging area
        AVX_STACK_K_SWITCH(up.K, hipLaunchKernelGGL(k_sel_pass_up<KT>, dim3(g), dim3(kT), lds_p, s, a))
    }
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

int avx_uv_percentiles_device(avx_ctx* ctx, int count, const float* const* x, const size_t* n, const double* q, double* const* out_dev, hipStream_t s) {
    UvScratch u;
    int rc = uv_small_scratch(ctx, s, &u);
    if (rc) return rc;
    PctReq rq[16];
    if (count > 16) return avx_fail(ctx, AVX_ERR_INVALID, "at most 16 percentiles per call");
    for (int i = 0; i < count; ++i) rq[i] = PctReq{x[i], n[i], q[i], out_dev[i]};
    return run_percentiles(ctx, u, rq, count, s);
}

int avx_uv_plane_blur_device(avx_ctx* ctx, const float* in, float* out, int K, int H, int W, int ksize, const double* taps_host, hipStream_t s) {
    BlurArgs a{};
    a.in = in; a.out = out; a.K = K; a.H = H; a.W = W; a.r = ksize / 2; a.stats = nullptr; a.scale_mode = 0;
    for (int i = 0; i < ksize && i < AVX_MAX_KSIZE; ++i) a.taps[i] = ksize == 1 ? 1.0f : (float)taps_host[i];
    return launch_plane_blur(ctx, a, s);
}


// Host side of the recompute form: batches of up to kBeeMaxFrames frames share every launch.
constexpr int kBeeMaxFrames = 32;
// float16 NHWC cube -> K float32 planes + per-workgroup statistics through k_spectral_nhwc_h (the staged, HBM-rate kernel).  The grid is
// grid_for(ctx, n) -- what k_finalize_stats is then told; same pixel-to-thread assignment and FMA order over b as k_spectral_integrate.
static int launch_spectral_nhwc_h(avx_ctx* ctx, hipStream_t s, const __half* c, size_t n, int B, const float* weights_host, int K, float* out_planes,
                                  Stat3* partials, int* g_out) {
    const int KP = K <= 4 ? 4 : (K <= 8 ? 8 : (K <= 12 ? 12 : 16));
    float wT[129 * 16];
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < KP; ++k) wT[b * KP + k] = k < K ? weights_host[(size_t)k * B + b] : 0.0f;
    // weights are constants of the caller's pipeline: device copy + host mirror in constant slot 1 of the stream's workspace,
    // compared byte for byte and re-uploaded (stream-ordered) only when they change (avx_const_upload)
    avx_ws* wsp = avx_workspace(ctx, s);
    if (!wsp) return AVX_ERR_NOMEM;
    float* dwT = nullptr;
    {
        const int rcu = avx_const_upload(ctx, wsp, 1, wT, sizeof(float) * B * KP, s, (void**)&dwT);
        if (rcu) return rcu;
    }
    const size_t lds = (size_t)kT * B * 2 + 16;
    const int g = grid_for(ctx, n);
    if (g_out) *g_out = g;
    // AVX_SPEC_PIPE=0 pins the plain (un-pipelined) tile loop
    static const bool pipe_on = [] { const char* e = getenv("AVX_SPEC_PIPE"); return !(e && e[0] == '0'); }();
    const int nv = !pipe_on ? 0 : B <= 32 ? 4 : B <= 88 ? 11 : 0;
#define AVX_SPEC2(KPV, NVV)                                                                                                                  \
    {                                                                                                                                        \
        AVX_HIP(ctx, hipFuncSetAttribute((const void*)k_spectral_nhwc_h<KPV, NVV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));   \
        hipLaunchKernelGGL((k_spectral_nhwc_h<KPV, NVV>), dim3(g), dim3(kT), lds, s, c, n, B, dwT, K, out_planes, partials);                 \
    }
#define AVX_SPEC(KPV) { if (nv == 4) AVX_SPEC2(KPV, 4) else if (nv == 11) AVX_SPEC2(KPV, 11) else AVX_SPEC2(KPV, 0) }
    if (KP == 4) AVX_SPEC(4) else if (KP == 8) AVX_SPEC(8) else if (KP == 12) AVX_SPEC(12) else AVX_SPEC(16)
#undef AVX_SPEC2
#undef AVX_SPEC
    return AVX_OK;
}

static int honeybee_recompute(avx_ctx* ctx, const uint8_t* in_hwc, uint8_t* out_hwc, int n_frames, int H, int W, const avx_honeybee_desc* d, hipStream_t s) {
    const size_t n = (size_t)H * W;
    avx_ws* ws = avx_workspace(ctx, s);
    if (!ws) return AVX_ERR_NOMEM;
    const int g = grid_for(ctx, n);  // blocks per frame of the statistics pass (<= 8 per CU)
    const size_t part_bytes = (size_t)ctx->num_cus * 8 * 3 * sizeof(Stat3);
    const size_t per_frame = part_bytes + 4 * sizeof(float4) + (size_t)kSelMax * 2048 * 4 + kSelMax * sizeof(SelState) + 64 + 8 * sizeof(double);
    if (!ws->bee_small) {
        AVX_HIP(ctx, hipMalloc(&ws->bee_small, per_frame * kBeeMaxFrames + 256));
        AVX_HIP(ctx, hipMemsetAsync(ws->bee_small, 0, per_frame * kBeeMaxFrames + 256, s));  // histograms and tickets start at zero; the kernels leave them so
    }
    char* p = (char*)ws->bee_small;
    Stat3* partials = (Stat3*)p; p += part_bytes * kBeeMaxFrames;
    float4* stats = (float4*)p; p += 4 * sizeof(float4) * kBeeMaxFrames;
    uint32_t* hist = (uint32_t*)p; p += (size_t)kSelMax * 2048 * 4 * kBeeMaxFrames;
    SelState* sel = (SelState*)p; p += kSelMax * sizeof(SelState) * kBeeMaxFrames;
    uint32_t* ticket = (uint32_t*)p; p += 64 * kBeeMaxFrames;
    p = (char*)(((uintptr_t)p + 15) & ~(uintptr_t)15);
    double* pct = (double*)p;
    float* mat = nullptr;
    int rc = d->source == 0 ? avx_const_upload(ctx, ws, 0, d->rgb_matrix, sizeof(float) * 9, s, (void**)&mat)
             : d->source == 1 ? avx_const_upload(ctx, ws, 0, d->weights_host, sizeof(float) * 3 * d->bands, s, (void**)&mat) : AVX_OK;
    if (rc) return rc;
    const int R = d->blur_ksize / 2, mode = d->mapping;
    const int nj = mode == 2 ? 2 : (mode == 0 ? 3 : (mode == 3 ? 1 : 0));
    float *raw = nullptr, *selp = nullptr;
    {
        // AVX_BEE_SELPLANES=0: all three radix passes recompute (no plane scratch)
        static const bool planes_on = [] { const char* e = getenv("AVX_BEE_SELPLANES"); return !(e && e[0] == '0'); }();
        const int fb = n_frames < kBeeMaxFrames ? n_frames : kBeeMaxFrames;
        const size_t raw_f = d->source == 1 ? n * 3 * fb : 0, sel_f = (planes_on && nj > 0) ? n * nj * fb : 0;
        if (raw_f + sel_f) {
            if ((rc = avx_ensure_scratch(ctx, ws, sizeof(float) * (raw_f + sel_f) + 256))) return rc;
            if (raw_f) raw = (float*)ws->d_scratch;
            if (sel_f) selp = (float*)ws->d_scratch + raw_f;
        }
        if (d->source == 2) raw = const_cast<float*>(d->catches);  // the catches came with their statistics (csrc/mst_fused.hip: conv_out's epilogue)
    }
    const double q = mode == 3 ? 98.0 : 95.0;
    const float vi = (float)(n - 1) * ((float)q / 100.0f);  // NumPy evaluates the virtual index in float32 (run_percentiles)
    float lo = floorf(vi);
    if (lo < 0) lo = 0;
    if (lo > (float)(n - 1)) lo = (float)(n - 1);
    const long tiles = (long)((W + 31) / 32) * ((H + 31) / 32);
    for (int f0 = 0; f0 < n_frames; f0 += kBeeMaxFrames) {
        const int F = n_frames - f0 < kBeeMaxFrames ? n_frames - f0 : kBeeMaxFrames;
        // 1) catches: statistics only (uint8 frames) or planes + statistics (HSI cubes), then the von Kries denominators
        if (d->source == 2) {
            hipLaunchKernelGGL(k_finalize_stats, dim3(1), dim3(1024), 0, s, (const Stat3*)d->catch_partials, d->n_catch_partials, 3, n, d->adaptation, d->eps, stats);
        } else if (d->source == 0) {
            hipLaunchKernelGGL(k_rgb_to_planes<3>, dim3(g, F), dim3(kT), 0, s, in_hwc + (size_t)f0 * n * 3, n, ctx->d_decode_lut, mat, 3, (float*)nullptr, partials);
        } else {
            const size_t esz = d->hsi_dtype == 0 ? 4 : 2;
            for (int f = 0; f < F; ++f) {
                const char* cube = (const char*)d->hsi + (size_t)(f0 + f) * n * d->bands * esz;
                if (d->hsi_layout == 0 && d->hsi_dtype == 1 && ((uintptr_t)cube & 15u) == 0 && d->bands <= 129) {  // what an MST++-style model hands over
                    const int rcs = launch_spectral_nhwc_h(ctx, s, (const __half*)cube, n, d->bands, d->weights_host, 3, raw + (size_t)f * 3 * n, partials + (size_t)f * g * 3, nullptr);
                    if (rcs) return rcs;
                    continue;
                }
                hipLaunchKernelGGL(k_spectral_integrate<3>, dim3(g), dim3(kT), sizeof(float) * 3 * d->bands, s, (const void*)cube, d->hsi_layout, d->hsi_dtype, n,
                                   d->bands, mat, 3, raw + (size_t)f * 3 * n, partials + (size_t)f * g * 3);
            }
        }
        if (d->source != 2) hipLaunchKernelGGL(k_finalize_stats, dim3(F), dim3(1024), 0, s, partials, g, 3, n, d->adaptation, d->eps, stats);
        BeeArgs a{};
        a.in = in_hwc ? in_hwc + (size_t)f0 * n * 3 : nullptr; a.raw = raw; a.H = H; a.W = W; a.lut = ctx->d_decode_lut; a.mat = mat; a.stats = stats;
        a.out = out_hwc + (size_t)f0 * n * 3;
        a.out_f = d->out_float ? reinterpret_cast<float*>(out_hwc) + (size_t)f0 * n * 3 : nullptr;
        a.scale_mode = d->adaptation ? 1 : 0;
        a.t0 = R ? (float)d->blur_taps_host[0] : 1.f; a.t1 = R ? (float)d->blur_taps_host[1] : 1.f; a.t2 = R ? (float)d->blur_taps_host[2] : 1.f;
        a.mode = mode; a.n_jobs = nj; a.st = sel; a.hist = hist; a.ticket = ticket; a.pct = pct;
        a.rank0 = (unsigned long long)lo; a.gamma = vi - lo; a.has_next = (size_t)lo + 1 < n ? 1 : 0;
        a.eps = d->eps;
        for (int i = 0; i < 9; ++i) a.M[i] = d->custom_matrix[i];
        a.enc_thr = ctx->d_enc_thr_f32; a.coarse = ctx->d_coarse_f32; a.lo_key = ctx->coarse_lo_key[0];
        long per = ((long)ctx->num_cus * 3 + F - 1) / F;
        const int gx = (int)(tiles < per ? tiles : (per < 1 ? 1 : per));
        const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
#define AVX_BEE(SRCV, RV, STG, NJV) hipLaunchKernelGGL((k_bee_tile<SRCV, RV, STG, NJV>), dim3(gx, F), dim3(kT), 0, s, a)
#define AVX_BEE_SR(STG, NJV)                                                                         \
        {                                                                                            \
            if (d->source == 0) { if (R) AVX_BEE(0, 1, STG, NJV); else AVX_BEE(0, 0, STG, NJV); }    \
            else { if (R) AVX_BEE(1, 1, STG, NJV); else AVX_BEE(1, 0, STG, NJV); }                   \
        }
        // 2) the order statistics the mapping needs: three radix passes.  The first recomputes the values from the frame (and writes them
        //    out as planes, 4 B per value); passes 2 and 3 only look at elements inside the selected prefix, so they read those planes with
        //    the plain selection kernel (all F frames per launch) instead of recomputing catches, rescale and blur twice more.
        a.sel_planes = selp;
        for (int ps = 0; ps < (selp ? 1 : 3) && nj > 0; ++ps) {
            a.pass = ps; a.shift = shifts[ps]; a.bits = bits[ps];
            if (nj == 1) AVX_BEE_SR(0, 1) else if (nj == 2) AVX_BEE_SR(0, 2) else AVX_BEE_SR(0, 3)
        }
        if (selp && nj > 0) {
            SelPass sp{};
            sp.n_jobs = nj; sp.st = sel; sp.hist = hist; sp.ticket = ticket; sp.x_frame_stride = (size_t)nj * n; sp.out_frame_stride = 8;
            for (int j = 0; j < nj; ++j) sp.job[j] = SelJob{selp + (size_t)j * n, n, (unsigned long long)lo, vi - lo, (size_t)lo + 1 < n ? 1 : 0, pct + j};
            const size_t want = (n + (size_t)kT * 16 - 1) / ((size_t)kT * 16), capb = (size_t)ctx->num_cus;
            const int gs = (int)(want < capb ? (want ? want : 1) : capb);
            for (int ps = 1; ps < 3; ++ps) {
                sp.pass = ps; sp.shift = shifts[ps]; sp.bits = bits[ps];
                if (nj == 1) hipLaunchKernelGGL(k_sel_pass<1>, dim3(gs, F), dim3(kT), 0, s, sp);
                else if (nj == 2) hipLaunchKernelGGL(k_sel_pass<2>, dim3(gs, F), dim3(kT), 0, s, sp);
                else hipLaunchKernelGGL(k_sel_pass<4>, dim3(gs, F), dim3(kT), 0, s, sp);
            }
        }
        // 3) map + encode
        a.pass = 0; a.shift = 0; a.bits = 11;
        AVX_BEE_SR(1, 1)
#undef AVX_BEE_SR
#undef AVX_BEE
        AVX_HIP(ctx, hipGetLastError());
    }
    return AVX_OK;
}

extern "C" {

int avx_percentile(avx_ctx* ctx, const float* data, size_t n, double q, double* out_host, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, data && out_host && n > 0 && q >= 0.0 && q <= 100.0, "avx_percentile: bad arguments");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    UvScratch u;
    int rc = uv_small_scratch(ctx, s, &u);
    if (rc) return rc;
    rc = run_percentile(ctx, u, data, n, q, u.pct, s);
    if (rc) return rc;
    AVX_HIP(ctx, hipMemcpyAsync(out_host, u.pct, sizeof(double), hipMemcpyDeviceToHost, s));
    AVX_HIP(ctx, hipStreamSynchronize(s));
    return AVX_OK;
}

int avx_spectral_integrate(avx_ctx* ctx, const void* hsi, int layout, int dtype, int H, int W, int B, const float* weights_host, int K,
                           float* out_planes, float* stats_host, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, hsi && weights_host && out_planes, "avx_spectral_integrate: NULL pointer");
    AVX_REQUIRE(ctx, H > 0 && W > 0 && B > 0 && B <= 129 && K > 0 && K <= 16, "avx_spectral_integrate: bad shape H=%d W=%d B=%d K=%d", H, W, B, K);
    AVX_REQUIRE(ctx, (layout == 0 || layout == 1) && (dtype == 0 || dtype == 1), "avx_spectral_integrate: bad layout/dtype");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    UvScratch u;
    int rc = uv_small_scratch(ctx, s, &u);
    if (rc) return rc;
    const size_t n = (size_t)H * W;
    int g = grid_for(ctx, n);
    if (layout == 0 && dtype == 1 && ((uintptr_t)hsi & 15u) == 0) {  // float16 NHWC: the staged, HBM-rate kernel
        if ((rc = launch_spectral_nhwc_h(ctx, s, (const __half*)hsi, n, B, weights_host, K, out_planes, u.partials, &g))) return rc;
    } else {
        float* dmat = nullptr;  // cached: uploaded only when the weights change
        { avx_ws* wsc = avx_workspace(ctx, s); if (!wsc) return AVX_ERR_NOMEM; const int rcu = avx_const_upload(ctx, wsc, 0, weights_host, sizeof(float) * K * B, s, (void**)&dmat); if (rcu) return rcu; }
        hipLaunchKernelGGL(k_spectral_integrate<16>, dim3(g), dim3(kT), sizeof(float) * K * B, s, hsi, layout, dtype, n, B, dmat, K, out_planes, u.partials);
    }
    hipLaunchKernelGGL(k_finalize_stats, dim3(1), dim3(1024), 0, s, u.partials, g, K, n, 0, 0.0f, u.stats);
    AVX_HIP(ctx, hipGetLastError());
    if (stats_host) {
        AVX_HIP(ctx, hipMemcpyAsync(stats_host, u.stats, sizeof(float4) * K, hipMemcpyDeviceToHost, s));
        AVX_HIP(ctx, hipStreamSynchronize(s));
    }
    return AVX_OK;
}

int avx_planes_gaussian_blur(avx_ctx* ctx, const float* in, float* out, int K, int H, int W, int ksize, const double* taps_host,
                             int scale_mode, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, in && out && in != out, "avx_planes_gaussian_blur: NULL or aliased planes");
    AVX_REQUIRE(ctx, K > 0 && K <= 16 && H > 0 && W > 0, "avx_planes_gaussian_blur: bad shape");
    AVX_REQUIRE(ctx, ksize >= 1 && ksize <= AVX_MAX_KSIZE && (ksize & 1), "avx_planes_gaussian_blur: ksize %d must be odd, 1..%d", ksize, AVX_MAX_KSIZE);
    AVX_REQUIRE(ctx, ksize == 1 || taps_host, "avx_planes_gaussian_blur: taps_host is NULL");
    AVX_REQUIRE(ctx, scale_mode == 0 || scale_mode == 1 || scale_mode == 3, "avx_planes_gaussian_blur: bad scale_mode");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    UvScratch u;
    int rc = uv_small_scratch(ctx, s, &u);
    if (rc) return rc;
    BlurArgs a{};
    a.in = in; a.out = out; a.K = K; a.H = H; a.W = W; a.r = ksize / 2; a.stats = u.stats; a.scale_mode = scale_mode;
    for (int i = 0; i < ksize; ++i) a.taps[i] = ksize == 1 ? 1.0f : (float)taps_host[i];
    return launch_plane_blur(ctx, a, s);
}

int avx_plane_stats(avx_ctx* ctx, const float* planes, int K, size_t n, int adapt, float eps, float* stats_host, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, planes && K > 0 && K <= 16 && n > 0 && adapt >= 0 && adapt <= 3, "avx_plane_stats: bad arguments");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    UvScratch u;
    int rc = uv_small_scratch(ctx, s, &u);
    if (rc) return rc;
    const int g = grid_for(ctx, n);
    hipLaunchKernelGGL(k_plane_stats<16>, dim3(g), dim3(kT), 0, s, planes, n, K, u.partials);
    hipLaunchKernelGGL(k_finalize_stats, dim3(1), dim3(1024), 0, s, u.partials, g, K, n, adapt, eps, u.stats);
    AVX_HIP(ctx, hipGetLastError());
    if (stats_host) {
        AVX_HIP(ctx, hipMemcpyAsync(stats_host, u.stats, sizeof(float4) * K, hipMemcpyDeviceToHost, s));
        AVX_HIP(ctx, hipStreamSynchronize(s));
    }
    return AVX_OK;
}

int avx_rgb_to_hsi_lobes(avx_ctx* ctx, const void* in_hwc, int in_is_u8, int H, int W, int B, const float* gains_host, float denom,
                         float* out_hwb, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, in_hwc && gains_host && out_hwb && H > 0 && W > 0 && B >= 2 && B <= 129, "avx_rgb_to_hsi_lobes: bad arguments");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    UvScratch u;
    int rc = uv_small_scratch(ctx, s, &u);
    if (rc) return rc;
    float* dmat = nullptr;  // cached: uploaded only when the gains change
    { avx_ws* wsc = avx_workspace(ctx, s); if (!wsc) return AVX_ERR_NOMEM; const int rcu = avx_const_upload(ctx, wsc, 0, gains_host, sizeof(float) * 3 * B, s, (void**)&dmat); if (rcu) return rcu; }
    const size_t n = (size_t)H * W;
    hipLaunchKernelGGL(k_lobes_cube, dim3(grid_for(ctx, n)), dim3(kT), sizeof(float) * 3 * B, s, in_hwc, in_is_u8, ctx->d_decode_lut, n, B,
                       dmat, denom, out_hwb);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

// Whole honeybee tail for n_frames frames (honeybee.py:125-175); see include/avx.h.
int avx_honeybee_u8(avx_ctx* ctx, const uint8_t* in_hwc, uint8_t* out_hwc, int n_frames, int H, int W, const avx_honeybee_desc* d,
                    float* debug_planes, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, d && d->struct_size == sizeof(avx_honeybee_desc), "avx_honeybee_u8: desc is NULL or struct_size mismatch");
    AVX_REQUIRE(ctx, out_hwc && n_frames >= 0 && H > 0 && W > 0, "avx_honeybee_u8: bad arguments");
    AVX_REQUIRE(ctx, d->source >= 0 && d->source <= 2, "avx_honeybee_u8: bad source");
    AVX_REQUIRE(ctx, d->source == 1 ? (d->hsi && d->weights_host && d->bands > 0 && d->bands <= 129) : d->source == 2 ? (d->catches && d->catch_partials && d->n_catch_partials > 0) : (in_hwc != nullptr),
                "avx_honeybee_u8: missing input");
    AVX_REQUIRE(ctx, d->source != 2 || (n_frames == 1 && d->mapping != 4 && !debug_planes && (d->blur_ksize == 0 || d->blur_ksize == 1 || d->blur_ksize == 3) &&
                                        d->n_catch_partials <= ctx->num_cus * 8 && (((uintptr_t)d->catch_partials) & 15u) == 0),
                "avx_honeybee_u8: source 2 (catches given) takes one frame, the tile schedule's mappings and blur sizes, and at most 8 statistics records per CU");
    AVX_REQUIRE(ctx, d->adaptation >= 0 && d->adaptation <= 2 && d->mapping >= 0 && d->mapping <= 4, "avx_honeybee_u8: bad adaptation/mapping");
    AVX_REQUIRE(ctx, d->blur_ksize == 0 || ((d->blur_ksize & 1) && d->blur_ksize <= AVX_MAX_KSIZE && d->blur_taps_host), "avx_honeybee_u8: bad blur");
    if (n_frames == 0) return AVX_OK;
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s0 = avx_pick_stream(ctx, stream);
    const size_t n = (size_t)H * W;
    const int g = grid_for(ctx, n);
    {   // the recompute form (k_bee_tile): every mapping but falsecolor_uv_mixed (a second order statistic over its own output), blur k <= 3
        // Measured (MI355X, opponent mapping, device-resident batches): 26.4 vs 20.2 GP/s at 1080p (8 frames per step), 35.6 vs 30.6-33.3 at 4K
        // (4 frames) once the second and third radix passes read the planes the first one writes (AVX_BEE_SELPLANES; 28.7 with all three
        // recomputing): the tile pipeline is bound by its arithmetic (three IEEE divisions, a square root and the 3 x 3 passes per pixel and
        // pass), not by bytes.  AVX_BEE_FUSED=0 pins the plane route below.
        const char* pin = getenv("AVX_BEE_FUSED");
        const bool want = (pin && *pin ? atoi(pin) != 0 : true) || d->source == 2;
        if (want && d->mapping != 4 && !debug_planes && (d->blur_ksize == 0 || d->blur_ksize == 1 || d->blur_ksize == 3))
            return honeybee_recompute(ctx, in_hwc, out_hwc, n_frames, H, W, d, s0);
    }
    // Frame lanes: the frames of a batch are independent and no kernel of this tail fills the GPU at video sizes, so frame f
    // runs on internal stream f % lanes with that stream's own workspace (AVX_UV_LANES pins the count; 1 = the caller's
    // stream only, which is also what a single frame uses).
    int want_lanes = n_frames < 4 ? n_frames : 4;
    { const char* e = getenv("AVX_UV_LANES"); if (e && *e) want_lanes = atoi(e) < n_frames ? atoi(e) : n_frames; }
    const int n_lanes = want_lanes > 1 ? avx_lanes(ctx, want_lanes) : 0;
    hipStream_t lane_s[avx_ctx::kMaxLanes];
    UvScratch lane_u[avx_ctx::kMaxLanes];
    float* lane_raw[avx_ctx::kMaxLanes];
    const int n_run = n_lanes > 1 ? n_lanes : 1;
    if (n_lanes > 1) AVX_HIP(ctx, hipEventRecord(ctx->lane_fork, s0));
    for (int l = 0; l < n_run; ++l) {
        lane_s[l] = n_lanes > 1 ? ctx->lanes[l] : s0;
        if (n_lanes > 1) AVX_HIP(ctx, hipStreamWaitEvent(lane_s[l], ctx->lane_fork, 0));
        avx_ws* ws = nullptr;
        int rc = uv_small_scratch(ctx, lane_s[l], &lane_u[l], &ws);
        if (rc) return rc;
        rc = avx_ensure_scratch(ctx, ws, sizeof(float) * n * 9 + 256);
        if (rc) return rc;
        lane_raw[l] = (float*)ws->d_scratch;
        // the colour matrix / band weights: cached per lane workspace, uploaded only when they change
        rc = d->source == 0 ? avx_const_upload(ctx, ws, 0, d->rgb_matrix, sizeof(float) * 9, lane_s[l], (void**)&lane_u[l].mat)
                            : avx_const_upload(ctx, ws, 0, d->weights_host, sizeof(float) * 3 * d->bands, lane_s[l], (void**)&lane_u[l].mat);
        if (rc) return rc;
    }
    int rc = AVX_OK;
    for (int f = 0; f < n_frames; ++f) {
        const int l = f % n_run;
        hipStream_t s = lane_s[l];
        const UvScratch& u = lane_u[l];
        float* raw = lane_raw[l];                 // 3 planes: catches
        float* ubg = raw + 3 * n;                 // 3 planes: adapted + blurred
        float* aux = ubg + 3 * n;                 // 3 planes: radius/L, or mixed
        uint8_t* out = out_hwc + (size_t)f * n * 3;
        // 1-3) catches U, B, G (+ per-plane statistics)
        if (d->source == 0) {
            hipLaunchKernelGGL(k_rgb_to_planes<3>, dim3(g), dim3(kT), 0, s, in_hwc + (size_t)f * n * 3, n, ctx->d_decode_lut, u.mat, 3, raw, u.partials);
        } else {
            const size_t esz = d->hsi_dtype == 0 ? 4 : 2;
            const char* cube = (const char*)d->hsi + (size_t)f * n * d->bands * esz;
            if (d->hsi_layout == 0 && d->hsi_dtype == 1 && ((uintptr_t)cube & 15u) == 0 && d->bands <= 129) {  // what an MST++-style model hands over
                if ((rc = launch_spectral_nhwc_h(ctx, s, (const __half*)cube, n, d->bands, d->weights_host, 3, raw, u.partials, nullptr))) return rc;
            } else {
                hipLaunchKernelGGL(k_spectral_integrate<3>, dim3(g), dim3(kT), sizeof(float) * 3 * d->bands, s, (const void*)cube, d->hsi_layout, d->hsi_dtype, n,
                                   d->bands, u.mat, 3, raw, u.partials);
            }
        }
        // 4) von Kries denominators
        hipLaunchKernelGGL(k_finalize_stats, dim3(1), dim3(1024), 0, s, u.partials, g, 3, n, d->adaptation, d->eps, u.stats);
        // 5) (x / white) then Gaussian blur
        {
            BlurArgs a{};
            a.in = raw; a.out = ubg; a.K = 3; a.H = H; a.W = W; a.r = d->blur_ksize / 2; a.stats = u.stats; a.scale_mode = d->adaptation ? 1 : 0;
            for (int i = 0; i < d->blur_ksize; ++i) a.taps[i] = (float)d->blur_taps_host[i];
            { const int rcb = launch_plane_blur(ctx, a, s); if (rcb) return rcb; }
        }
        if (debug_planes) AVX_HIP(ctx, hipMemcpyAsync(debug_planes + (size_t)f * 3 * n, ubg, sizeof(float) * 3 * n, hipMemcpyDeviceToDevice, s));
        // 6) percentiles the mapping needs, then map + encode
        MapArgs m{};
        m.UBG = ubg; m.n = n; m.out = out; m.mode = d->mapping;
        m.out_f = d->out_float ? reinterpret_cast<float*>(out_hwc) + (size_t)f * n * 3 : nullptr; m.pct = u.pct; m.eps = d->eps; m.alpha = d->mixed_alpha; m.mixed = aux;
        for (int i = 0; i < 9; ++i) m.M[i] = d->custom_matrix[i];
        m.enc_thr = ctx->d_enc_thr_f32; m.coarse = ctx->d_coarse_f32; m.lo_key = ctx->coarse_lo_key[0]; m.pass = 0;
        if (d->mapping == 2) {
            hipLaunchKernelGGL(k_opponent_prep, dim3(g), dim3(kT), 0, s, ubg, n, aux, aux + n);
            const PctReq rq[2] = {{aux, n, 95.0, u.pct + 0}, {aux + n, n, 95.0, u.pct + 1}};
            if ((rc = run_percentiles(ctx, u, rq, 2, s))) return rc;
        } else if (d->mapping == 0 || d->mapping == 4) {
            PctReq rq[4];
            int nr = 0;
            for (int k = 0; k < 3; ++k) rq[nr++] = PctReq{ubg + (size_t)k * n, n, 95.0, u.pct + k};
            if (d->mapping == 4) rq[nr++] = PctReq{ubg, n, 98.0, u.pct + 3};
            if ((rc = run_percentiles(ctx, u, rq, nr, s))) return rc;
        }
        if (d->mapping == 3) { if ((rc = run_percentile(ctx, u, ubg, n, 98.0, u.pct + 0, s))) return rc; }
        if (d->mapping == 4) {
            m.pass = 1;
            hipLaunchKernelGGL(k_map_encode, dim3(g), dim3(kT), 0, s, m);
            if ((rc = run_percentile(ctx, u, aux, 3 * n, 99.0, u.pct + 4, s))) return rc;
            m.pass = 2;
        }
        hipLaunchKernelGGL(k_map_encode, dim3(g), dim3(kT), 0, s, m);
        AVX_HIP(ctx, hipGetLastError());
    }
    if (n_lanes > 1)
        for (int l = 0; l < n_run; ++l) {  // join: the caller's stream continues when every lane has drained
            AVX_HIP(ctx, hipEventRecord(ctx->lane_done[l], lane_s[l]));
            AVX_HIP(ctx, hipStreamWaitEvent(s0, ctx->lane_done[l], 0));
        }
    return AVX_OK;
}

}  // extern "C"
