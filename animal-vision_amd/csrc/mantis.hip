// csrc/mantis.hip -- MantisShrimp.visualize (animals/mantis_shrimp.py:143-279) on the device: the band stack and
// spectral "barcode" (a17/a22) and the polarisation / unsharp / scanline / periphery finish (a23), chained from
// the shared building blocks (decode table, cubic panorama warp, area/linear resize, plane blur, Sobel,
// radix-select percentile, threshold encode).  Float contract: within 1e-4 relative of the reference; the
// categorical `argmax` of :202 can differ where two band maps tie to within rounding.
#include <cmath>
#include <type_traits>
#include <vector>

#include "dichromat_common.h"
#include "stack_up.h"

using namespace avxk;

// from uv.hip / geom.hip (same library)
int avx_uv_percentile_device(avx_ctx* ctx, const float* x, size_t n, double q, double* out_dev, hipStream_t s);
int avx_uv_percentiles_device(avx_ctx* ctx, int count, const float* const* x, const size_t* n, const double* q, double* const* out_dev, hipStream_t s);
int avx_uv_plane_blur_device(avx_ctx* ctx, const float* in, float* out, int K, int H, int W, int ksize, const double* taps_host, hipStream_t s);
extern "C" int avx_resize_hwc(avx_ctx* ctx, const void* src, int dtype, int H, int W, int C, void* dst, int Hd, int Wd, int interp, void* stream);
extern "C" int avx_sobel3_plane(avx_ctx* ctx, const float* plane, int H, int W, float* gx, float* gy, void* stream);
int avx_geom_panorama_cubic(avx_ctx* ctx, const float* src, int H, int W, int newW, int start, float* dst, hipStream_t s, bool finite_src);
int avx_geom_cache_trim(avx_ctx* ctx, avx_ws* ws, hipStream_t s, int limit);
int avx_geom_linear_tables(avx_ctx* ctx, avx_ws* ws, hipStream_t s, int H, int W, int Hd, int Wd, avx_lin_tab* ax, avx_lin_tab* ay);
int avx_geom_cubic_x_tables(avx_ctx* ctx, avx_ws* ws, hipStream_t s, int W, int newW, int start, const int** idx, const float** a);
int avx_uv_percentile_up_device(avx_ctx* ctx, const StackUp& up, double q, double* out_dev, hipStream_t s, float* cand_buf);

namespace {

constexpr int kMT = 256;
constexpr int KMAX = 16;

__device__ __forceinline__ float s2l(float v) { return v <= 0.04045f ? v / 12.92f : powf((v + 0.055f) / 1.055f, 2.4f); }
__device__ __forceinline__ float clip01f(float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }

__global__ __launch_bounds__(kMT) void k_decode_lin(const uint8_t* __restrict__ in, const float* __restrict__ lut_g, size_t n3, float* __restrict__ out) {
    __shared__ float lut[256];
    for (int i = threadIdx.x; i < 256; i += kMT) lut[i] = lut_g[i];
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * kMT + threadIdx.x; i < n3; i += (size_t)gridDim.x * kMT) out[i] = lut[in[i]];
}

// from_float01(linear_to_srgb(clip(x, 0, 1)), uint8) (mantis_shrimp.py:153,278) as the threshold count
__device__ __forceinline__ float l2s_f(float l) {  // uv_helpers.linear_to_srgb (:40-44) of a value already clipped to [0,1]
    return l <= 0.0031308f ? l * 12.92f : 1.055f * powf(l, 1.0f / 2.4f) - 0.055f;
}
// out_f != NULL (float frames): from_float01(..., float dtype) = the float32 sRGB value itself
__global__ __launch_bounds__(kMT) void k_encode_hwc(const float* __restrict__ in, size_t n3, const float* __restrict__ thr_g, const uint8_t* __restrict__ coarse_g,
                                                    uint32_t lo_key, uint8_t* __restrict__ out, float* __restrict__ out_f = nullptr) {
    __shared__ float thr[256];
    __shared__ uint8_t coarse[kCoarseTableBytes];
    for (int i = threadIdx.x; i < 256; i += kMT) thr[i] = thr_g[i];
    for (int i = threadIdx.x; i < kCoarseTableBytes; i += kMT) coarse[i] = coarse_g[i];
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * kMT + threadIdx.x; i < n3; i += (size_t)gridDim.x * kMT) {
        if (out_f) out_f[i] = l2s_f(clip01f(in[i]));
        else out[i] = (uint8_t)quantize_coarse<float, kCoarseNFix>(in[i], thr, coarse, lo_key);
    }
}

// The UV front end of a uint8 frame in one pass: decode table -> panorama warp (x-only cubic, k_panorama_cubic_x's expression; idx == NULL:
// no warp) -> the linear frame (float32 HWC) AND its baseline encode.  Replaces three launches and a 25 MB intermediate round trip
// per 1080p frame; same tables, same arithmetic, byte-identical outputs.
__global__ __launch_bounds__(kMT) void k_uv_front_u8(const uint8_t* __restrict__ in, int H, int W, const int* __restrict__ idx, const float* __restrict__ ca,
                                                     const float* __restrict__ lut_g, const float* __restrict__ thr_g, const uint8_t* __restrict__ coarse_g,
                                                     uint32_t lo_key, float* __restrict__ lin_out, uint8_t* __restrict__ base_out) {
    __shared__ float lut[256];
    __shared__ float thr[256];
    __shared__ uint8_t coarse[kCoarseTableBytes];
    for (int i = threadIdx.x; i < 256; i += kMT) { lut[i] = lut_g[i]; thr[i] = thr_g[i]; }
    for (int i = threadIdx.x; i < kCoarseTableBytes; i += kMT) coarse[i] = coarse_g[i];
    __syncthreads();
    const unsigned total = (unsigned)H * (unsigned)W;  // 32-bit pixel index (frames < 2^32 / 12 pixels: the host checks): the 64-bit form paid two 64-bit divisions per pixel
    for (unsigned i = blockIdx.x * kMT + threadIdx.x; i < total; i += gridDim.x * kMT) {
        float v[3];
        if (idx) {
            const unsigned y = i / (unsigned)W, x = i - y * (unsigned)W;
            const int* ix = idx + 4 * x;
            const float* a = ca + 4 * x;
            const uint8_t* S = in + (size_t)y * W * 3;
            const int i0 = ix[0] * 3, i1 = ix[1] * 3, i2 = ix[2] * 3, i3 = ix[3] * 3;
            const float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = lut[S[i0 + c]] * a0 + lut[S[i1 + c]] * a1 + lut[S[i2 + c]] * a2 + lut[S[i3 + c]] * a3;
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = lut[in[(size_t)i * 3 + c]];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lin_out[(size_t)i * 3 + c] = v[c];
            if (base_out) base_out[(size_t)i * 3 + c] = (uint8_t)quantize_coarse<float, kCoarseNFix>(v[c], thr, coarse, lo_key);
        }
    }
}

// classic_rgb_to_hsi on an ALREADY-LINEAR float frame (quirk Q6: linearised again) folded with the K band-pass
// windows: stack[p][k] = sum_j M[k][j] * s2l(lin[p][j])   (HxWxK, the reference's S layout before safe_norm).
// A pixel with a negative channel (cubic overshoot of the panorama warp) takes the band-by-band route, where
// the per-wavelength clamp_min(0) of classic_rgb_to_hsi.py:81 is applied before the band-pass sum.
struct StackArgs { const float* lin; size_t n; const float* M; int K; int B; const float* gains; float denom; const float* wts; float* out; };
// Round 3: the band-by-band pixels of a workgroup's 256 are COLLECTED (LDS list) and then worked off densely, one (pixel, window) pair per thread: with the route
// taken inline a wave paid the B-step serial loop for all 64 lanes as soon as one of its pixels had a negative channel -- on structured frames nearly every wave
// (hummingbird's 270 x 480 small frame: 53 us; 5 % of the pixels take the route).  The tables (B x 3 gains, K x B window weights) are staged in LDS once per
// workgroup.  A window's sum still runs over b in order with the same operations, so the values are the inline form's.
constexpr int kStackBMax = 160;  // wavelengths the LDS tables hold (the reference's grids: 31, 81, 129); longer grids read global memory
__global__ __launch_bounds__(kMT) void k_rgbf_to_stack(StackArgs a) {
    __shared__ float M[KMAX * 3];
    __shared__ float gl[kStackBMax * 3];
    __shared__ float wl[kStackBMax * KMAX];  // [b][KMAX]
    __shared__ float lc[kMT][3];             // the collected pixels: linearised channels ...
    __shared__ unsigned lp[kMT];             // ... and position in the workgroup's batch
    __shared__ unsigned cnt;
    const bool tab = a.B <= kStackBMax;
    for (int i = threadIdx.x; i < a.K * 3; i += kMT) M[i] = a.M[i];
    if (tab) {
        for (int i = threadIdx.x; i < a.B * 3; i += kMT) gl[i] = a.gains[i];
        for (int i = threadIdx.x; i < a.B * KMAX; i += kMT) { const int b = i / KMAX, k = i - b * KMAX; wl[i] = k < a.K ? a.wts[(size_t)k * a.B + b] : 0.f; }
    }
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    for (size_t base = (size_t)blockIdx.x * kMT; base < a.n; base += (size_t)gridDim.x * kMT) {
        const size_t p = base + threadIdx.x;
        if (p < a.n) {
            const float c0 = s2l(a.lin[3 * p]), c1 = s2l(a.lin[3 * p + 1]), c2 = s2l(a.lin[3 * p + 2]);
            if (c0 >= 0.f && c1 >= 0.f && c2 >= 0.f) {
                for (int k = 0; k < a.K; ++k) a.out[p * a.K + k] = fma_t(c2, M[3 * k + 2], fma_t(c1, M[3 * k + 1], c0 * M[3 * k]));
            } else {
                const unsigned i = atomicAdd(&cnt, 1u);
                lc[i][0] = c0; lc[i][1] = c1; lc[i][2] = c2; lp[i] = threadIdx.x;
            }
        }
        __syncthreads();
        const unsigned m = cnt;
        // band by band, one (collected pixel, window) pair per thread: the wavelength's value (three products and one division) then the window's FMA, b in order;
        // a zero weight leaves the sum unchanged (as the inline form's skip did)
        for (unsigned it = threadIdx.x; it < m * (unsigned)a.K; it += kMT) {
            const unsigned i = it / (unsigned)a.K, k = it - i * (unsigned)a.K;
            const float c0 = lc[i][0], c1 = lc[i][1], c2 = lc[i][2];
            float acc = 0.f;
            const float rden = 1.0f / a.denom;  // one denominator for every wavelength: div_by_r (stack_up.h)
#pragma unroll 4
            for (int b = 0; b < a.B; ++b) {
                const float g0 = tab ? gl[3 * b] : a.gains[3 * b], g1 = tab ? gl[3 * b + 1] : a.gains[3 * b + 1], g2 = tab ? gl[3 * b + 2] : a.gains[3 * b + 2];
                const float spec = div_by_r((g2 * c2 + g1 * c1) + g0 * c0, a.denom, rden);
                const float sp = spec > 0.f ? spec : 0.f;
                const float w = tab ? wl[b * KMAX + k] : a.wts[(size_t)k * a.B + b];
                if (w != 0.f) acc = fma_t(sp, w, acc);
            }
            a.out[(base + lp[i]) * a.K + k] = acc;
        }
        __syncthreads();
        if (threadIdx.x == 0) cnt = 0;
        __syncthreads();
    }
}

// per-channel min/max of an HxWxK stack -> mm[k] = {min, max} (two-level, deterministic)
__global__ __launch_bounds__(kMT) void k_stack_minmax(const float* __restrict__ S, size_t n, int K, float2* __restrict__ partial) {
    __shared__ float smn[kMT / 64][KMAX], smx[kMT / 64][KMAX];
    float mn[KMAX], mx[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { mn[k] = 3.4e38f; mx[k] = -3.4e38f; }
    for (size_t p = (size_t)blockIdx.x * kMT + threadIdx.x; p < n; p += (size_t)gridDim.x * kMT)
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) { const float v = S[p * K + k]; mn[k] = fminf(mn[k], v); mx[k] = fmaxf(mx[k], v); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < K; ++k) {
        float a = mn[k], b = mx[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a = fminf(a, __shfl_xor(a, o)); b = fmaxf(b, __shfl_xor(b, o)); }
        if (lane == 0) { smn[wave][k] = a; smx[wave][k] = b; }
    }
    __syncthreads();
    if ((int)threadIdx.x < K) {
        float a = smn[0][threadIdx.x], b = smx[0][threadIdx.x];
        for (int w = 1; w < kMT / 64; ++w) { a = fminf(a, smn[w][threadIdx.x]); b = fmaxf(b, smx[w][threadIdx.x]); }
        partial[(size_t)blockIdx.x * K + threadIdx.x] = make_float2(a, b);
    }
}
// the same over a stack that exists only at reduced size: every pixel's K resized values are recomputed (stack_up.h), nothing is written
template <int K>
__global__ __launch_bounds__(kMT) void k_up_minmax(const StackUp u, size_t cap_floats, float2* __restrict__ partial) {
    __shared__ float smn[kMT / 64][KMAX], smx[kMT / 64][KMAX];
    float mn[KMAX], mx[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { mn[k] = 3.4e38f; mx[k] = -3.4e38f; }
    extern __shared__ float tile_lds[];
    stack_tiles<K, false>(u, tile_lds, cap_floats, [&](int, int, float (&v)[K]) {
#pragma unroll
        for (int k = 0; k < K; ++k) { mn[k] = fminf(mn[k], v[k]); mx[k] = fmaxf(mx[k], v[k]); }
    });
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < u.K) {
            float a = mn[k], b = mx[k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { a = fminf(a, __shfl_xor(a, o)); b = fmaxf(b, __shfl_xor(b, o)); }
            if (lane == 0) { smn[wave][k] = a; smx[wave][k] = b; }
        }
    __syncthreads();
    if ((int)threadIdx.x < u.K) {
        float a = smn[0][threadIdx.x], b = smx[0][threadIdx.x];
        for (int w = 1; w < kMT / 64; ++w) { a = fminf(a, smn[w][threadIdx.x]); b = fmaxf(b, smx[w][threadIdx.x]); }
        partial[(size_t)blockIdx.x * u.K + threadIdx.x] = make_float2(a, b);
    }
}
__global__ void k_stack_minmax_final(const float2* partial, int nblocks, int K, float2* mm) {
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (k >= K) return;
    float a = 3.4e38f, b = -3.4e38f;
    for (int i = lane; i < nblocks; i += 64) { const float2 v = partial[(size_t)i * K + k]; a = fminf(a, v.x); b = fmaxf(b, v.y); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a = fminf(a, __shfl_xor(a, o)); b = fmaxf(b, __shfl_xor(b, o)); }
    if (lane == 0) mm[k] = make_float2(a, b);
}
// safe_norm per band, in place (uv_helpers.py:47-53)
__global__ __launch_bounds__(kMT) void k_stack_safe_norm(float* __restrict__ S, size_t n, int K, const float2* __restrict__ mm) {
    for (size_t i = (size_t)blockIdx.x * kMT + threadIdx.x; i < n * K; i += (size_t)gridDim.x * kMT) {
        const float2 m = mm[i % K];
        S[i] = stack_norm(S[i], m, 1.0f / (m.y - m.x));  // the same expression as the read-through form (stack_up.h): the two routes stay identical
    }
}

struct BarcodeArgs { const float* S; size_t n; int K; const double* p95; float lut[KMAX * 3]; float wtm, sat; float* bar /*3 planes*/; float* broad;
                     StackUp up; /* S == NULL: the normalised stack is read through its resize (stack_up.h) */ };
// mantis_shrimp.py:199-211,224: S_norm, argmax, soft/hard tint, saturation; broad = mean(S_norm).  sn: the pixel's K safe-normed
// band values on entry.  Fully unrolled over KMAX with k < K predicates (the arrays stay in registers), the argmax band's LUT row is
// carried along instead of being indexed afterwards.
template <int KT>
__device__ __forceinline__ void barcode_pixel(const BarcodeArgs& a, size_t p, float (&sn)[KT], float den) {
    float sum = 0.f, best = -1.f;
    float hard[3] = {a.lut[0], a.lut[1], a.lut[2]};
    const float rden = 1.0f / den;
#pragma unroll
    for (int k = 0; k < KT; ++k)
        if (KT < KMAX || k < a.K) {
            sn[k] = clip01f(div_by_r(sn[k], den, rden));
            sum += sn[k];
            if (sn[k] > best) { best = sn[k]; hard[0] = a.lut[3 * k]; hard[1] = a.lut[3 * k + 1]; hard[2] = a.lut[3 * k + 2]; }  // np.argmax: first maximum
        }
    const float wden = sum + 1e-8f, rwden = 1.0f / wden;
    float soft[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KT; ++k)
        if (KT < KMAX || k < a.K) {
            const float w = div_by_r(sn[k], wden, rwden);
#pragma unroll
            for (int c = 0; c < 3; ++c) soft[c] = fma_t(w, a.lut[3 * k + c], soft[c]);
        }
    float bc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) bc[c] = (1.0f - a.wtm) * soft[c] + a.wtm * hard[c];
    const float Yb = (0.2126f * bc[0] + 0.7152f * bc[1]) + 0.0722f * bc[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) a.bar[(size_t)c * a.n + p] = clip01f(Yb + (bc[c] - Yb) * (1.0f + a.sat));
    a.broad[p] = sum / (float)a.K;
}
__global__ __launch_bounds__(kMT) void k_barcode(BarcodeArgs a) {
    const float den = (float)a.p95[0] + 1e-8f;
    for (size_t p = (size_t)blockIdx.x * kMT + threadIdx.x; p < a.n; p += (size_t)gridDim.x * kMT) {
        float sn[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < a.K) sn[k] = a.S[p * a.K + k];
        barcode_pixel<KMAX>(a, p, sn, den);
    }
}
// the same with the normalised stack read through its resize, tile by tile (stack_up.h)
template <int K>
__global__ __launch_bounds__(kMT) void k_barcode_up(BarcodeArgs a, size_t cap_floats) {
    extern __shared__ float tile_lds[];
    const float den = (float)a.p95[0] + 1e-8f;
    stack_tiles<K, true>(a.up, tile_lds, cap_floats, [&](int x, int y, float (&v)[K]) { barcode_pixel<K>(a, (size_t)y * a.up.W + x, v, den); });
}

// :214-218 render = baseline_lin with red kill and haze, HWC -> 3 planes
__global__ __launch_bounds__(kMT) void k_prep_render(const float* __restrict__ lin, size_t n, float red_keep, float haze, float haze_keep, float3 tint, float* __restrict__ P) {
    for (size_t p = (size_t)blockIdx.x * kMT + threadIdx.x; p < n; p += (size_t)gridDim.x * kMT) {
        float v[3] = {clip01f(lin[3 * p] * red_keep), lin[3 * p + 1], lin[3 * p + 2]};
        const float t[3] = {tint.x, tint.y, tint.z};
#pragma unroll
        for (int c = 0; c < 3; ++c) P[(size_t)c * n + p] = haze > 0.f ? haze_keep * v[c] + haze * t[c] : v[c];
    }
}

// :226-242 the polarisation gain of one pixel from its Sobel pair.  cos(2 theta) and sin(2 theta) of theta = atan2(gy, gx) are taken algebraically,
// (gx^2 - gy^2) / (gx^2 + gy^2) and 2 gx gy / (gx^2 + gy^2) (1, 0 for a zero gradient, where atan2 gives 0): the same functions of the gradient's direction to ~2 ulp
// without atan2f + cosf + sinf (~100 of this stage's ~250 instructions per pixel; round 3).  cos2g / sin2g arrive as float32((1 - mix) * global).
__device__ __forceinline__ float pol_gain_of(float gxv, float gyv, float cos2g, float sin2g, float mix, float lin_s, float lin_gamma, float circ_s) {
    const float r2 = gxv * gxv + gyv * gyv;
    float c2 = 1.0f, s2 = 0.0f;
    if (r2 > 0.f) {
        const float inv = 1.0f / r2;
        c2 = (gxv * gxv - gyv * gyv) * inv;
        s2 = (2.0f * gxv * gyv) * inv;
    }
    const float cm = cos2g + mix * c2, sm = sin2g + mix * s2;
    const float al = powf(clip01f(0.5f * (cm + 1.0f)), lin_gamma);
    const float ac = clip01f(0.5f * (sm + 1.0f));
    return (1.0f + lin_s * al) + circ_s * ac;
}

struct PolArgs { const float* gx; const float* gy; size_t n; float cos2g, sin2g, mix, lin_s, lin_gamma, circ_s; float* gain;
                 const float* broad; int H, W; /* broad != NULL: the Sobel pair is computed here (geom.hip::k_sobel3's expressions), gx / gy unused */ };
// :226-242
__global__ __launch_bounds__(kMT) void k_polgain(PolArgs a) {
    for (size_t p = (size_t)blockIdx.x * kMT + threadIdx.x; p < a.n; p += (size_t)gridDim.x * kMT) {
        float gxv, gyv;
        if (a.broad) {  // cv2.Sobel(ksize=3, BORDER_REFLECT_101), dx and dy, as k_sobel3
            const int x = (int)(p % a.W), y = (int)(p / a.W);
            const int xm = reflect101(x - 1, a.W), xp = reflect101(x + 1, a.W);
            float dd[3], ss[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float* R = a.broad + (size_t)reflect101(y - 1 + k, a.H) * a.W;
                dd[k] = R[xp] - R[xm];
                ss[k] = R[xm] + R[x] * 2 + R[xp];
            }
            gxv = dd[0] + dd[1] * 2 + dd[2];
            gyv = ss[2] - ss[0];
        } else { gxv = a.gx[p]; gyv = a.gy[p]; }
        a.gain[p] = pol_gain_of(gxv, gyv, a.cos2g, a.sin2g, a.mix, a.lin_s, a.lin_gamma, a.circ_s);
    }
}

// :244-250: unsharp guided by pol_gain, then the barcode blend
// rows != NULL: the scanline gain of :264-265 (k_scan_gain) applied to the blended value in the same pass
__global__ __launch_bounds__(kMT) void k_unsharp_blend(const float* __restrict__ P, const float* __restrict__ Bl, const float* __restrict__ gain, const float* __restrict__ bar,
                                                       size_t n, float amount, int do_unsharp, float opacity, float* __restrict__ out,
                                                       const float* __restrict__ rows = nullptr, float scan_gain = 0.f) {
    for (size_t i = (size_t)blockIdx.x * kMT + threadIdx.x; i < 3 * n; i += (size_t)gridDim.x * kMT) {
        float r = P[i];
        if (do_unsharp) {
            float high = r - Bl[i];
            high = high < -1.f ? -1.f : (high > 1.f ? 1.f : high);
            r = clip01f(r + (amount * gain[i % n]) * high);
        }
        float v = clip01f((1.0f - opacity) * r + opacity * bar[i]);
        if (rows) v = clip01f(v * (1.0f + scan_gain * (rows[i % n] - 0.5f)));
        out[i] = v;
    }
}

__global__ __launch_bounds__(kMT) void k_rows_plane(const float* __restrict__ rows, int H, int W, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * kMT + threadIdx.x; i < (size_t)H * W; i += (size_t)gridDim.x * kMT) out[i] = rows[i / W] * 1.0f;
}
struct FinishArgs { const float* P; const float* periph; int H, W; const float* xx; const float* yy; float softness, radius; int do_periph;
                    const float* thr; const uint8_t* coarse; uint32_t lo_key; uint8_t* out; float* out_f; };
// :268-278: radial sigmoid blend with the blurred copy, then encode
__global__ __launch_bounds__(kMT) void k_finish(FinishArgs a) {
    __shared__ float thr[256];
    __shared__ uint8_t coarse[kCoarseTableBytes];
    for (int i = threadIdx.x; i < 256; i += kMT) thr[i] = a.thr[i];
    for (int i = threadIdx.x; i < kCoarseTableBytes; i += kMT) coarse[i] = a.coarse[i];
    __syncthreads();
    const size_t n = (size_t)a.H * a.W;
    for (size_t p = (size_t)blockIdx.x * kMT + threadIdx.x; p < n; p += (size_t)gridDim.x * kMT) {
        float t = 0.f;
        if (a.do_periph) {
            const float xx = a.xx[p % a.W], yy = a.yy[p / a.W];
            const float r = __fsqrt_rn(xx * xx + yy * yy);
            t = 1.0f / (1.0f + expf(-a.softness * (r - a.radius)));
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = a.P[(size_t)c * n + p];
            if (a.do_periph) v = (1.0f - t) * v + t * a.periph[(size_t)c * n + p];
            if (a.out_f) a.out_f[p * 3 + c] = l2s_f(clip01f(v));
            else a.out[p * 3 + c] = (uint8_t)quantize_coarse<float, kCoarseNFix>(v, thr, coarse, a.lo_key);
        }
    }
}

// ---- the finishing stages fused around their three Gaussian blurs (round 3) -----------------------------------------------------------
// :214-278 as launched until round 2: prep_render | blur | polgain | blur | unsharp_blend | blur | finish -- seven launches, 211 bytes per pixel of
// plane round trips.  Here each blur carries its neighbours: the stage in front of it is its PROLOGUE (the value a tile load stores in LDS), the
// stage behind it its EPILOGUE (applied to the column pass's result while the tile -- and with it the blur's own input at the same pixel -- is
// still in LDS):
//   MODE 1  pre-soften:  source = prep_render(baseline) (red kill, haze)                                    -> render planes
//   MODE 2  unsharp:     source = render (or prep_render(baseline) when pre-soften is off); epilogue = the polarisation gain of :226-242 (Sobel
//                        pair of `broad` + atan2 / cos / sin / pow, once per pixel), unsharp, barcode blend, scanline gain   -> P2 planes
//   MODE 3  periphery:   source = P2; epilogue = radial sigmoid blend with the blurred copy + encode                         -> uint8 / float frame
// A workgroup owns a 64 x 32 tile and passes its three planes through the same LDS tile one after the other; the per-pixel quantities of an
// epilogue (gain; sigmoid weight; the three codes of an output pixel) stay in registers across the planes.  The blur is k_plane_blur_t's, term for
// term (row pass = sequential FMA left to right, column pass = centre tap then fma(x[+j] + x[-j], k[r+j], s)), and every pointwise expression is
// the unfused kernel's: the frames are identical to the seven-launch route (tests/test_mantis_gpu.py; AVX_MANTIS_FUSE=0 pins that route).
// k_finish's blend weight t = 1 / (1 + exp(-softness (r - radius))), r = sqrt(xx^2 + yy^2) (:270-275), as a plane
__global__ __launch_bounds__(kMT) void k_periph_plane(const float* __restrict__ xx, const float* __restrict__ yy, int H, int W, float softness, float radius, float* __restrict__ out) {
    for (size_t p = (size_t)blockIdx.x * kMT + threadIdx.x; p < (size_t)H * W; p += (size_t)gridDim.x * kMT) {
        const float xv = xx[p % W], yv = yy[p / W];
        const float r = __fsqrt_rn(xv * xv + yv * yv);
        out[p] = 1.0f / (1.0f + expf(-softness * (r - radius)));
    }
}

struct FuseArgs {
    int H, W;
    float taps[2 * 5 + 1];
    const float* lin;     // baseline (H x W x 3 linear) when the source is prep_render, else NULL
    float red_keep, haze, haze_keep, tint[3];
    const float* in;      // 3 planes (source when lin == NULL)
    float* out;           // 3 planes (MODE 1, 2)
    // MODE 2
    const float* bar; const float* broad; const float* rows;
    float cos2g, sin2g, mix, lin_s, lin_gamma, circ_s, amount, opacity, scan_gain;
    // MODE 3
    const float* tplane;  // the blend weight of every pixel (k_periph_plane)
    const float* thr; const uint8_t* coarse; uint32_t lo_key; uint8_t* out_u8; float* out_f;
};

template <int R, int MODE, int TH>
__global__ __launch_bounds__(kMT) void k_mantis_fused_blur(const FuseArgs a) {
    constexpr int TW = 64, AH = TH + 2 * R, AW = TW + 2 * R, AWP = (AW + 3) & ~3, NT = 2 * R + 1, NIT = (TH / 4) * TW / kMT, NLD = (AH * AW + kMT - 1) / kMT;
    static_assert((TH / 4) * TW % kMT == 0, "column-pass items per thread");
    extern __shared__ __align__(16) float fsm[];
    float* A = fsm;                    // [3][AH * AWP]  source tiles of the three planes (columns x0 - R ... at 0 ...)
    float* Bm = fsm + 3 * AH * AWP;    // [3][AH * TW]   row pass
    __shared__ float thr[MODE == 3 ? 256 : 1];
    __shared__ uint8_t coarse[MODE == 3 ? kCoarseTableBytes : 1];
    if constexpr (MODE == 3) {
        for (int i = threadIdx.x; i < 256; i += kMT) thr[i] = a.thr[i];
        for (int i = threadIdx.x; i < kCoarseTableBytes; i += kMT) coarse[i] = a.coarse[i];
    }
    float tp[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) tp[j] = a.taps[j];
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const size_t n = (size_t)a.H * a.W;
    const float tint[3] = {a.tint[0], a.tint[1], a.tint[2]};
    // this thread's elements of a source tile: LDS slot (lane-constant) and position in the tile
    int l_off[NLD], l_y[NLD], l_x[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int i = threadIdx.x + q * kMT, ii = i < AH * AW ? i : 0;
        l_y[q] = ii / AW; l_x[q] = ii - l_y[q] * AW;
        l_off[q] = i < AH * AW ? l_y[q] * AWP + l_x[q] : -1;
    }
    unsigned l_g[NLD];  // ... and in the frame, relative to the tile's first halo pixel (interior tiles: no reflection, one scalar base)
#pragma unroll
    for (int q = 0; q < NLD; ++q) l_g[q] = (unsigned)(l_y[q] * a.W + l_x[q]);
    for (int tile = blockIdx.x; tile < tiles_x * tiles_y; tile += gridDim.x) {
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int x0 = tx * TW, y0 = ty * TH;
        // interior: the tile and its halo lie inside the frame (uniform) -- no reflection, no bounds tests, one scalar origin; two instantiations of the body
        const bool interior = x0 - R >= 0 && x0 + TW + R <= a.W && y0 - R >= 0 && y0 + TH + R <= a.H;
        auto body = [&](auto int_tag) {
        constexpr bool INT = decltype(int_tag)::value;
        // ---- the three planes' source tiles: every load of the tile in flight together, the epilogue's per-pixel quantities computed under them ----
        float ld[NLD][3];
        const size_t origin = INT ? (size_t)(y0 - R) * a.W + (x0 - R) : 0;
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            if (l_off[q] < 0) continue;
            size_t p;
            if constexpr (INT) p = origin + l_g[q];
            else p = (size_t)reflect101(y0 - R + l_y[q], a.H) * a.W + reflect101(x0 - R + l_x[q], a.W);
            if (a.lin) {
#pragma unroll
                for (int k = 0; k < 3; ++k) ld[q][k] = a.lin[3 * p + k];
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) ld[q][k] = a.in[(size_t)k * n + p];
            }
        }
        // per-pixel quantities of the epilogue: this thread's NIT items x 4 rows of the column pass
        float pq[NIT][4];
        uint32_t codes[NIT][4];
        (void)pq; (void)codes;
        if constexpr (MODE == 2 || MODE == 3) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = threadIdx.x + it * kMT, g = i / TW, xl = i - g * TW, x = x0 + xl;
#pragma unroll
                for (int yy = 0; yy < 4; ++yy) {
                    const int y = y0 + 4 * g + yy;
                    pq[it][yy] = 0.f; codes[it][yy] = 0;
                    if (!INT && (y >= a.H || x >= a.W)) continue;
                    if constexpr (MODE == 2) {  // k_polgain (the Sobel pair formed here)
                        const int xm = INT ? x - 1 : reflect101(x - 1, a.W), xp = INT ? x + 1 : reflect101(x + 1, a.W);
                        float dd[3], ss[3];
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const float* Rw = a.broad + (size_t)(INT ? y - 1 + k : reflect101(y - 1 + k, a.H)) * a.W;
                            dd[k] = Rw[xp] - Rw[xm];
                            ss[k] = Rw[xm] + Rw[x] * 2 + Rw[xp];
                        }
                        const float gxv = dd[0] + dd[1] * 2 + dd[2], gyv = ss[2] - ss[0];
                        pq[it][yy] = pol_gain_of(gxv, gyv, a.cos2g, a.sin2g, a.mix, a.lin_s, a.lin_gamma, a.circ_s);
                    } else {  // k_finish's radial sigmoid: a function of (x, y) only, read from the plane built once per geometry (k_periph_plane)
                        pq[it][yy] = a.tplane[(size_t)y * a.W + x];
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            if (l_off[q] < 0) continue;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float v = ld[q][k];
                if (a.lin) {  // k_prep_render
                    if (k == 0) v = clip01f(v * a.red_keep);
                    v = a.haze > 0.f ? a.haze_keep * v + a.haze * tint[k] : v;
                }
                A[k * (AH * AWP) + l_off[q]] = v;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 3 * AH * (TW / 4); i += kMT) {  // row pass (k_plane_blur_t), the three planes
            const int k = i / (AH * (TW / 4)), i2 = i - k * (AH * (TW / 4)), ly = i2 / (TW / 4), g = i2 - ly * (TW / 4);
            const float4* wp = reinterpret_cast<const float4*>(A + k * (AH * AWP) + ly * AWP + 4 * g);
            float w[(4 + 2 * R + 3) & ~3];
#pragma unroll
            for (int q = 0; q < (4 + 2 * R + 3) / 4; ++q) {
                float4 v = wp[q];
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
            *reinterpret_cast<float4*>(Bm + k * (AH * TW) + ly * TW + 4 * g) = make_float4(o[0], o[1], o[2], o[3]);
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {  // column pass + epilogue
            const int i = threadIdx.x + it * kMT, g = i / TW, xl = i - g * TW, x = x0 + xl;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float w[4 + 2 * R];
#pragma unroll
                for (int q = 0; q < 4 + 2 * R; ++q) w[q] = Bm[k * (AH * TW) + (4 * g + q) * TW + xl];
#pragma unroll
                for (int yy = 0; yy < 4; ++yy) {
                    const int yl = 4 * g + yy, y = y0 + yl;
                    float sacc = w[yy + R] * tp[R];
#pragma unroll
                    for (int j = 1; j <= R; ++j) sacc = fma_t(w[yy + R + j] + w[yy + R - j], tp[R + j], sacc);
                    if (!INT && (y >= a.H || x >= a.W)) continue;
                    const size_t p = (size_t)y * a.W + x;
                    if constexpr (MODE == 1) {
                        a.out[(size_t)k * n + p] = sacc;
                    } else {
                        const float c = A[k * (AH * AWP) + (yl + R) * AWP + xl + R];  // the blur's own input at this pixel
                        if constexpr (MODE == 2) {  // k_unsharp_blend
                            float high = c - sacc;
                            high = high < -1.f ? -1.f : (high > 1.f ? 1.f : high);
                            const float r = clip01f(c + (a.amount * pq[it][yy]) * high);
                            float v = clip01f((1.0f - a.opacity) * r + a.opacity * a.bar[(size_t)k * n + p]);
                            if (a.rows) v = clip01f(v * (1.0f + a.scan_gain * (a.rows[p] - 0.5f)));
                            a.out[(size_t)k * n + p] = v;
                        } else {  // k_finish
                            const float t = pq[it][yy];
                            const float v = (1.0f - t) * c + t * sacc;
                            if (a.out_f) a.out_f[p * 3 + k] = l2s_f(clip01f(v));
                            else codes[it][yy] |= quantize_coarse<float, kCoarseNFix>(v, thr, coarse, a.lo_key) << (8 * k);
                        }
                    }
                }
            }
            if constexpr (MODE == 3) {
                if (!a.out_f) {
#pragma unroll
                    for (int yy = 0; yy < 4; ++yy) {
                        const int y = y0 + 4 * g + yy;
                        if (!INT && (y >= a.H || x >= a.W)) continue;
                        uint8_t* o = a.out_u8 + ((size_t)y * a.W + x) * 3;
                        o[0] = (uint8_t)(codes[it][yy] & 0xff); o[1] = (uint8_t)((codes[it][yy] >> 8) & 0xff); o[2] = (uint8_t)(codes[it][yy] >> 16);
                    }
                }
            }
        }
        __syncthreads();
        };  // body
        if (interior) body(std::true_type{}); else body(std::false_type{});
    }
}

template <int MODE>
int launch_mantis_fused(avx_ctx* ctx, const FuseArgs& a, int r, hipStream_t s) {
    const char* th_env = getenv("AVX_MANTIS_FUSE_TH");  // A/B: tile height 16 | 32
    const int th = th_env && atoi(th_env) == 32 ? 32 : 16;
    const long tiles = (long)((a.W + 63) / 64) * ((a.H + th - 1) / th);
    const int g = (int)(tiles < (long)ctx->num_cus * 8 ? tiles : (long)ctx->num_cus * 8);
#define AVX_MF(RR, THV)                                                                                                                       \
    {                                                                                                                                         \
        const size_t lds = sizeof(float) * 3 * ((size_t)(THV + 2 * RR) * ((64 + 2 * RR + 3) & ~3) + (size_t)(THV + 2 * RR) * 64);              \
        AVX_HIP(ctx, hipFuncSetAttribute((const void*)k_mantis_fused_blur<RR, MODE, THV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((k_mantis_fused_blur<RR, MODE, THV>), dim3(g), dim3(kMT), lds, s, a);                                              \
    }
#define AVX_MF2(RR) case RR: if (th == 32) AVX_MF(RR, 32) else AVX_MF(RR, 16) break;
    switch (r) {
        AVX_MF2(1) AVX_MF2(2) AVX_MF2(3) AVX_MF2(4) AVX_MF2(5)
        default: return AVX_ERR_UNSUPPORTED;
    }
#undef AVX_MF2
#undef AVX_MF
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}
inline bool fuse_radius_ok(int ksize) { return ksize >= 3 && ksize <= 11 && (ksize & 1); }

int grid_for(avx_ctx* ctx, size_t items) {
    const size_t want = (items + kMT - 1) / kMT, cap = (size_t)ctx->num_cus * 16;
    return (int)(want < cap ? (want ? want : 1) : cap);
}

}  // namespace

// ---- shared front end and band stack of the UV species (also exported on their own, include/avx.h) ------------
// to_float01 + srgb_to_linear -> panorama_warp -> baseline encode.  tmp: 3*H*W floats (used when the warp is on).
static int uv_front(avx_ctx* ctx, const uint8_t* in_hwc, int H, int W, int newW, float* tmp, float* lin_out, uint8_t* base_out, hipStream_t s) {
    const size_t n = (size_t)H * W;
    int rc;
    if (!getenv("AVX_UV_FRONT_SPLIT") && n < ((size_t)1 << 31)) {  // one pass (k_uv_front_u8, 32-bit pixel indices); AVX_UV_FRONT_SPLIT=1: decode, warp and encode as separate launches
        const int* idx = nullptr;
        const float* ca = nullptr;
        if (newW > W) {
            avx_ws* ws = avx_workspace(ctx, s);
            if (!ws) return AVX_ERR_NOMEM;
            if ((rc = avx_geom_cubic_x_tables(ctx, ws, s, W, newW, (newW - W) / 2, &idx, &ca))) return rc;
        }
        hipLaunchKernelGGL(k_uv_front_u8, dim3(grid_for(ctx, n)), dim3(kMT), 0, s, in_hwc, H, W, idx, ca, ctx->d_decode_lut, ctx->d_enc_thr_f32, ctx->d_coarse_f32,
                           ctx->coarse_lo_key[0], lin_out, base_out);
        AVX_HIP(ctx, hipGetLastError());
        return AVX_OK;
    }
    if (newW > W) {
        hipLaunchKernelGGL(k_decode_lin, dim3(grid_for(ctx, 3 * n)), dim3(kMT), 0, s, in_hwc, ctx->d_decode_lut, 3 * n, tmp);
        if ((rc = avx_geom_panorama_cubic(ctx, tmp, H, W, newW, (newW - W) / 2, lin_out, s, true))) return rc;  // tmp: decoded uint8, finite
    } else {
        hipLaunchKernelGGL(k_decode_lin, dim3(grid_for(ctx, 3 * n)), dim3(kMT), 0, s, in_hwc, ctx->d_decode_lut, 3 * n, lin_out);
    }
    if (base_out)
        hipLaunchKernelGGL(k_encode_hwc, dim3(grid_for(ctx, 3 * n)), dim3(kMT), 0, s, lin_out, 3 * n, ctx->d_enc_thr_f32, ctx->d_coarse_f32, ctx->coarse_lo_key[0], base_out);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

// classic_rgb_to_hsi(_scaled) x K band windows -> HxWxK stack (raw integrate_band values).
// tab: (K*3 + B*3 + K*B) floats of device scratch; small / sstack: 3*hs*ws and hs*ws*K floats (reduced-size route).
static int band_stack(avx_ctx* ctx, const float* lin_hwc, int H, int W, int K, const float* M_host, int B, const float* gains_host, float denom,
                      const float* wts_host, int hs, int wsm, float* tab, float* small, float* sstack, float* stack, hipStream_t s) {
    // The three small tables are constants of a species: cached on the device under a hash of their contents (one
    // blocking upload the first time, no per-frame copies afterwards); `tab` is unused when the cache holds them.
    (void)tab;
    avx_ws* ws = avx_workspace(ctx, s);
    if (!ws) return AVX_ERR_NOMEM;
    { int rc0 = avx_geom_cache_trim(ctx, ws, s, 36); if (rc0) return rc0; }  // the resizes below trim at 48: they will not evict what we look up here
    uint64_t hkey = 1469598103934665603ull;
    auto mix = [&](const float* p, size_t cnt) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(p);
        for (size_t i = 0; i < cnt; ++i) { hkey ^= w[i]; hkey *= 1099511628211ull; }
    };
    mix(M_host, (size_t)K * 3); mix(gains_host, (size_t)B * 3); mix(wts_host, (size_t)B * K);
    hkey = (hkey & 0x003fffffffffffffull) | (7ull << 58);  // kind 7 (geometry tables use 1..3), component 0: a real pointer
    float* dM = nullptr;
    for (int i = 0; i < ws->n_geom_tabs; ++i)
        if (ws->geom_tabs[i].key == hkey) dM = (float*)ws->geom_tabs[i].dev;
    if (!dM) {
        if (ws->n_geom_tabs >= 60) return avx_fail(ctx, AVX_ERR_NOMEM, "band-table cache full");
        const size_t cnt = (size_t)K * 3 + (size_t)B * 3 + (size_t)B * K;
        AVX_HIP(ctx, hipMalloc((void**)&dM, cnt * sizeof(float) + 256));
        AVX_HIP(ctx, hipMemcpy(dM, M_host, sizeof(float) * K * 3, hipMemcpyHostToDevice));
        AVX_HIP(ctx, hipMemcpy(dM + K * 3, gains_host, sizeof(float) * B * 3, hipMemcpyHostToDevice));
        AVX_HIP(ctx, hipMemcpy(dM + K * 3 + 3 * B, wts_host, sizeof(float) * B * K, hipMemcpyHostToDevice));
        ws->geom_tabs[ws->n_geom_tabs++] = {hkey, dM};
    }
    float* dgains = dM + K * 3; float* dwts = dgains + 3 * B;
    const size_t n = (size_t)H * W, nsmall = (size_t)hs * wsm;
    int rc;
    if (hs != H || wsm != W) {
        if ((rc = avx_resize_hwc(ctx, lin_hwc, 0, H, W, 3, small, hs, wsm, 3, s))) return rc;
        hipLaunchKernelGGL(k_rgbf_to_stack, dim3(grid_for(ctx, nsmall)), dim3(kMT), 0, s, StackArgs{small, nsmall, dM, K, B, dgains, denom, dwts, sstack});
        if ((rc = avx_resize_hwc(ctx, sstack, 0, hs, wsm, K, stack, H, W, 1, s))) return rc;
    } else {
        hipLaunchKernelGGL(k_rgbf_to_stack, dim3(grid_for(ctx, n)), dim3(kMT), 0, s, StackArgs{lin_hwc, n, dM, K, B, dgains, denom, dwts, stack});
    }
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

extern "C" int avx_uv_front_u8(avx_ctx* ctx, const uint8_t* in_hwc, int H, int W, int pano_new_w, float* lin_hwc_out, uint8_t* baseline_hwc_out, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, in_hwc && lin_hwc_out && H > 0 && W > 0, "avx_uv_front_u8: bad arguments");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    avx_ws* ws = avx_workspace(ctx, s);
    if (!ws) return AVX_ERR_NOMEM;
    const int newW = pano_new_w > W ? pano_new_w : W;
    float* tmp = nullptr;
    if (newW > W) {
        int rc = avx_ensure_scratch(ctx, ws, sizeof(float) * 3 * (size_t)H * W);
        if (rc) return rc;
        tmp = (float*)ws->d_scratch;
    }
    return uv_front(ctx, in_hwc, H, W, newW, tmp, lin_hwc_out, baseline_hwc_out, s);
}

extern "C" int avx_panorama_warp_f32(avx_ctx* ctx, const float* src_hwc, int H, int W, int new_w, float* dst_hwc, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, src_hwc && dst_hwc && src_hwc != dst_hwc && H > 0 && W > 0 && new_w > W, "avx_panorama_warp_f32: bad arguments");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    return avx_geom_panorama_cubic(ctx, src_hwc, H, W, new_w, (new_w - W) / 2, dst_hwc, avx_pick_stream(ctx, stream), false);  // caller's float data
}

extern "C" int avx_band_stack(avx_ctx* ctx, const float* lin_hwc, int H, int W, const avx_band_stack_desc* d, float* stack_hwk_out, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, d && d->struct_size == sizeof(avx_band_stack_desc), "avx_band_stack: desc is NULL or struct_size mismatch");
    AVX_REQUIRE(ctx, lin_hwc && stack_hwk_out && H > 0 && W > 0, "avx_band_stack: bad arguments");
    AVX_REQUIRE(ctx, d->n_bands >= 1 && d->n_bands <= KMAX && d->band_matrix_host && d->n_wavelengths >= 1 && d->lobe_gains_host && d->band_weights_host && d->lobe_denom > 0.f,
                "avx_band_stack: bad band tables");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    avx_ws* ws = avx_workspace(ctx, s);
    if (!ws) return AVX_ERR_NOMEM;
    const int K = d->n_bands, B = d->n_wavelengths;
    const int hs = d->small_h > 0 ? d->small_h : H, wsm = d->small_w > 0 ? d->small_w : W;
    const size_t nsmall = (hs != H || wsm != W) ? (size_t)hs * wsm : 0;
    size_t off = 0;
    auto take = [&](size_t cnt) { size_t o = off; off += (cnt + 63) & ~(size_t)63; return o; };
    const size_t o_tab = take((size_t)K * 3 + (size_t)B * (3 + K)), o_small = take(3 * nsmall), o_sstack = take(nsmall * K);
    int rc = avx_ensure_scratch(ctx, ws, off * sizeof(float));
    if (rc) return rc;
    float* base = (float*)ws->d_scratch;
    return band_stack(ctx, lin_hwc, H, W, K, d->band_matrix_host, B, d->lobe_gains_host, d->lobe_denom, d->band_weights_host, hs, wsm, base + o_tab, base + o_small,
                      base + o_sstack, stack_hwk_out, s);
}

extern "C" int avx_percentiles_dev(avx_ctx* ctx, int count, const float* const* data_dev, const size_t* n, const double* q, double* const* out_dev, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, count >= 1 && count <= 16 && data_dev && n && q && out_dev, "avx_percentiles_dev: bad arguments");
    for (int i = 0; i < count; ++i)
        AVX_REQUIRE(ctx, data_dev[i] && out_dev[i] && n[i] > 0 && q[i] >= 0.0 && q[i] <= 100.0, "avx_percentiles_dev: bad request %d", i);
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    return avx_uv_percentiles_device(ctx, count, data_dev, n, q, out_dev, avx_pick_stream(ctx, stream));
}

extern "C" int avx_percentile_dev(avx_ctx* ctx, const float* data_dev, size_t n, double q, double* out_dev, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, data_dev && out_dev && n > 0 && q >= 0.0 && q <= 100.0, "avx_percentile_dev: bad arguments");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    return avx_uv_percentile_device(ctx, data_dev, n, q, out_dev, avx_pick_stream(ctx, stream));
}

static int mantis_check(avx_ctx* ctx, const uint8_t* in_hwc, uint8_t* out_base_hwc, uint8_t* out_hwc, int H, int W, const avx_mantis_desc* d) {
    AVX_REQUIRE(ctx, d && d->struct_size == sizeof(avx_mantis_desc), "avx_mantis_u8: desc is NULL or struct_size mismatch");
    AVX_REQUIRE(ctx, (in_hwc || d->lin_hwc_in) && out_base_hwc && out_hwc && H > 0 && W > 0, "avx_mantis_u8: bad arguments");
    AVX_REQUIRE(ctx, d->n_bands >= 1 && d->n_bands <= KMAX && d->band_matrix_host && d->band_lut_host, "avx_mantis_u8: bad band tables");
    AVX_REQUIRE(ctx, d->n_wavelengths >= 1 && d->lobe_gains_host && d->band_weights_host && d->lobe_denom > 0.f, "avx_mantis_u8: per-wavelength tables missing");
    AVX_REQUIRE(ctx, d->rows_host && d->xx_host && d->yy_host, "avx_mantis_u8: row/column tables missing");
    return AVX_OK;
}
static int mantis_frame(avx_ctx* ctx, const uint8_t* in_hwc, uint8_t* out_base_hwc, uint8_t* out_hwc, int H, int W, const avx_mantis_desc* d, hipStream_t s);

extern "C" int avx_mantis_u8(avx_ctx* ctx, const uint8_t* in_hwc, uint8_t* out_base_hwc, uint8_t* out_hwc, int H, int W, const avx_mantis_desc* d, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    const int rc = mantis_check(ctx, in_hwc, out_base_hwc, out_hwc, H, W, d);
    if (rc) return rc;
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    return mantis_frame(ctx, in_hwc, out_base_hwc, out_hwc, H, W, d, avx_pick_stream(ctx, stream));
}

// N uint8 frames (contiguous, H*W*3 each): the frames of a batch do not depend on each other, so they run on up to four "lanes"
// (streams of the context, each with its own workspace) forked from and joined back into `stream` -- a frame is ~25 dependent
// launches of 5-50 us at 1080p, most of them too small to fill the device on their own.  AVX_MANTIS_LANES=1 pins one lane.
extern "C" int avx_mantis_u8_batch(avx_ctx* ctx, const uint8_t* in_hwc, uint8_t* out_base_hwc, uint8_t* out_hwc, int n_frames, int H, int W,
                                   const avx_mantis_desc* d, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, n_frames > 0 && in_hwc, "avx_mantis_u8_batch: n_frames must be positive and the frames uint8");
    int rc = mantis_check(ctx, in_hwc, out_base_hwc, out_hwc, H, W, d);
    if (rc) return rc;
    AVX_REQUIRE(ctx, d->lin_hwc_in == nullptr && !d->out_float, "avx_mantis_u8_batch: uint8 frames only");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s0 = avx_pick_stream(ctx, stream);
    const char* le = getenv("AVX_MANTIS_LANES");
    int want = le ? atoi(le) : 4;
    want = want < 1 ? 1 : (want > n_frames ? n_frames : want);
    const int n_lanes = want > 1 ? avx_lanes(ctx, want) : 0;
    if (n_lanes > 1) {
        AVX_HIP(ctx, hipEventRecord(ctx->lane_fork, s0));
        for (int l = 0; l < n_lanes; ++l) AVX_HIP(ctx, hipStreamWaitEvent(ctx->lanes[l], ctx->lane_fork, 0));
    }
    const size_t fb = (size_t)H * W * 3;
    for (int f = 0; f < n_frames; ++f) {
        hipStream_t s = n_lanes > 1 ? ctx->lanes[f % n_lanes] : s0;
        if ((rc = mantis_frame(ctx, in_hwc + f * fb, out_base_hwc + f * fb, out_hwc + f * fb, H, W, d, s))) return rc;
    }
    if (n_lanes > 1)
        for (int l = 0; l < n_lanes; ++l) {
            AVX_HIP(ctx, hipEventRecord(ctx->lane_done[l], ctx->lanes[l]));
            AVX_HIP(ctx, hipStreamWaitEvent(s0, ctx->lane_done[l], 0));
        }
    return AVX_OK;
}

static int mantis_frame(avx_ctx* ctx, const uint8_t* in_hwc, uint8_t* out_base_hwc, uint8_t* out_hwc, int H, int W, const avx_mantis_desc* d, hipStream_t s) {
    avx_ws* ws = avx_workspace(ctx, s);
    if (!ws) return AVX_ERR_NOMEM;
    const int K = d->n_bands;
    const size_t n = (size_t)H * W;
    const int newW = d->pano_new_w > 0 ? d->pano_new_w : W;
    const int hs = d->hsi_small_h > 0 ? d->hsi_small_h : H, wsm = d->hsi_small_w > 0 ? d->hsi_small_w : W;
    const size_t nsmall = (size_t)hs * wsm;
    // scratch layout (floats)
    size_t off = 0;
    auto take = [&](size_t cnt) { size_t o = off; off += (cnt + 63) & ~(size_t)63; return o; };
    const size_t o_lin0 = take(3 * n), o_base = take(3 * n), o_small = take(3 * nsmall), o_sstack = take(nsmall * K),
                 o_stack = take(n * K), o_bar = take(3 * n), o_broad = take(n), o_gx = take(n), o_gy = take(n), o_gain = take(n), o_P0 = take(3 * n),
                 o_P1 = take(3 * n), o_P2 = take(3 * n), o_rows = take(n), o_rowsb = take(n), o_tab = take((size_t)K * 3 + H + W + H + (size_t)d->n_wavelengths * (3 + K) + 64),
                 o_part = take((size_t)ctx->num_cus * 16 * K * 2 + 64), o_mm = take(2 * KMAX + 64), o_pct = take(64);
    int rc = avx_ensure_scratch(ctx, ws, off * sizeof(float));
    if (rc) return rc;
    float* base = (float*)ws->d_scratch;
    float *lin0 = base + o_lin0, *blin = base + o_base, *small = base + o_small, *sstack = base + o_sstack, *stack = base + o_stack,
          *bar = base + o_bar, *broad = base + o_broad, *gx = base + o_gx, *gy = base + o_gy, *gain = base + o_gain, *P0 = base + o_P0, *P1 = base + o_P1,
          *P2 = base + o_P2, *rowsp = base + o_rows, *rowsb = base + o_rowsb, *tab = base + o_tab;
    float2* part = (float2*)(base + o_part);
    float2* mm = (float2*)(base + o_mm);
    double* pct = (double*)(base + o_pct);
    const int B = d->n_wavelengths;
    // constant tables of the species: cached per workspace, uploaded only when their bytes change (was six pageable copies in
    // front of every frame)
    float *dM, *drows, *dxx, *dyy, *dgains, *dwts;
    (void)tab; (void)rowsb; (void)rowsb;
    if ((rc = avx_const_upload(ctx, ws, 2, d->lobe_gains_host, sizeof(float) * B * 3, s, (void**)&dgains))) return rc;
    if ((rc = avx_const_upload(ctx, ws, 3, d->band_weights_host, sizeof(float) * B * K, s, (void**)&dwts))) return rc;
    if ((rc = avx_const_upload(ctx, ws, 4, d->band_matrix_host, sizeof(float) * K * 3, s, (void**)&dM))) return rc;
    if ((rc = avx_const_upload(ctx, ws, 5, d->rows_host, sizeof(float) * H, s, (void**)&drows))) return rc;
    if ((rc = avx_const_upload(ctx, ws, 6, d->xx_host, sizeof(float) * W, s, (void**)&dxx))) return rc;
    if ((rc = avx_const_upload(ctx, ws, 7, d->yy_host, sizeof(float) * H, s, (void**)&dyy))) return rc;
    const int g = grid_for(ctx, n);
    // 1) to_float01 + srgb_to_linear (:148-149); float frames arrive already linearised (lin_hwc_in, built by the caller's
    //    plane program: to_float01's max rule needs a frame-wide reduction)
    const float* baseline;
    float* out_f = d->out_float ? reinterpret_cast<float*>(out_hwc) : nullptr;
    if (!d->lin_hwc_in && !d->out_float) {  // uint8 frame: decode -> warp -> linear frame + baseline encode in one pass (uv_front)
        if ((rc = uv_front(ctx, in_hwc, H, W, newW, lin0, blin, out_base_hwc, s))) return rc;
        baseline = blin;
    } else {
        const float* lin_src = d->lin_hwc_in;
        if (!lin_src) {
            hipLaunchKernelGGL(k_decode_lin, dim3(grid_for(ctx, 3 * n)), dim3(kMT), 0, s, in_hwc, ctx->d_decode_lut, 3 * n, lin0);
            lin_src = lin0;
        }
        // 2) panorama warp (:152): INTER_CUBIC widen + centre crop
        baseline = lin_src;
        if (newW != W) {
            if ((rc = avx_geom_panorama_cubic(ctx, lin_src, H, W, newW, (newW - W) / 2, blin, s, d->lin_hwc_in == nullptr))) return rc;
            baseline = blin;
        }
        float* base_f = d->out_float ? reinterpret_cast<float*>(out_base_hwc) : nullptr;
        hipLaunchKernelGGL(k_encode_hwc, dim3(grid_for(ctx, 3 * n)), dim3(kMT), 0, s, baseline, 3 * n, ctx->d_enc_thr_f32, ctx->d_coarse_f32, ctx->coarse_lo_key[0], out_base_hwc, base_f);
    }
    // 3-4) RGB->HSI (optionally at reduced size) folded with the band windows -> HxWxK stack, safe_norm per band
    // AVX_MANTIS_UP=0 pins the materialised H x W x K stack on the reduced-size route too (A/B)
    const char* up_env = getenv("AVX_MANTIS_UP");  // read per call: tests flip it
    const bool virt = (hs != H || wsm != W) && !(up_env && up_env[0] == '0') && stack_k_tiled(K);
    BarcodeArgs b{};
    if (virt) {
        // reduced-size route: the resized stack is never built -- min/max, the percentile's three passes and the barcode recompute
        // each pixel's K values from the (cache-resident) small stack: five cheap passes instead of 83 MB written once and read seven times
        if ((rc = avx_resize_hwc(ctx, baseline, 0, H, W, 3, small, hs, wsm, 3, s))) return rc;
        hipLaunchKernelGGL(k_rgbf_to_stack, dim3(grid_for(ctx, nsmall)), dim3(kMT), 0, s, StackArgs{small, nsmall, dM, K, B, dgains, d->lobe_denom, dwts, sstack});
        StackUp up{};
        up.S = sstack; up.hs = hs; up.ws = wsm; up.H = H; up.W = W; up.K = K; up.mm = nullptr;
        if ((rc = avx_geom_linear_tables(ctx, ws, s, hs, wsm, H, W, &up.ax, &up.ay))) return rc;
        const size_t cap_floats = stack_tile_floats(hs, wsm, H, W, K), tile_lds = cap_floats * sizeof(float);
        AVX_REQUIRE(ctx, tile_lds <= 96 * 1024, "mantis: the source rectangle of a 64 x 16 tile does not fit LDS (%zu bytes)", tile_lds);
        AVX_STACK_K_SWITCH(K, AVX_HIP(ctx, hipFuncSetAttribute((const void*)k_up_minmax<KT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds)))
        AVX_STACK_K_SWITCH(K, AVX_HIP(ctx, hipFuncSetAttribute((const void*)k_barcode_up<KT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds)))
        const long up_tiles = (long)((W + kUpTW - 1) / kUpTW) * ((H + kUpTH - 1) / kUpTH), up_cap = (long)ctx->num_cus * 4;  // workgroups walk tiles: the per-workgroup epilogue is paid 1,024 times, not 4,096
        const int gu = (int)(up_tiles < up_cap ? up_tiles : up_cap);
        AVX_STACK_K_SWITCH(K, hipLaunchKernelGGL(k_up_minmax<KT>, dim3(gu), dim3(kMT), tile_lds, s, up, cap_floats, part))
        hipLaunchKernelGGL(k_stack_minmax_final, dim3(1), dim3(1024), 0, s, part, gu, K, mm);
        up.mm = mm;
        if ((rc = avx_uv_percentile_up_device(ctx, up, 95.0, pct, s, stack /*the materialised route's buffer: free here, H * W * K floats for the second pass's candidates*/))) return rc;
        b.S = nullptr; b.up = up;
    } else {
        if (hs != H || wsm != W) {
            if ((rc = avx_resize_hwc(ctx, baseline, 0, H, W, 3, small, hs, wsm, 3, s))) return rc;
            hipLaunchKernelGGL(k_rgbf_to_stack, dim3(grid_for(ctx, nsmall)), dim3(kMT), 0, s, StackArgs{small, nsmall, dM, K, B, dgains, d->lobe_denom, dwts, sstack});
            if ((rc = avx_resize_hwc(ctx, sstack, 0, hs, wsm, K, stack, H, W, 1, s))) return rc;
        } else {
            hipLaunchKernelGGL(k_rgbf_to_stack, dim3(g), dim3(kMT), 0, s, StackArgs{baseline, n, dM, K, B, dgains, d->lobe_denom, dwts, stack});
        }
        hipLaunchKernelGGL(k_stack_minmax, dim3(g), dim3(kMT), 0, s, stack, n, K, part);
        hipLaunchKernelGGL(k_stack_minmax_final, dim3(1), dim3(1024), 0, s, part, g, K, mm);
        hipLaunchKernelGGL(k_stack_safe_norm, dim3(grid_for(ctx, n * K)), dim3(kMT), 0, s, stack, n, K, mm);
        // 5-6) barcode
        if ((rc = avx_uv_percentile_device(ctx, stack, n * K, 95.0, pct, s))) return rc;
        b.S = stack;
    }
    b.n = n; b.K = K; b.p95 = pct; b.wtm = d->winner_take_most; b.sat = d->barcode_saturation; b.bar = bar; b.broad = broad;
    for (int i = 0; i < K * 3; ++i) b.lut[i] = d->band_lut_host[i];
    if (virt) {
        const size_t cap_floats = stack_tile_floats(hs, wsm, H, W, K);
        const long up_tiles = (long)((W + kUpTW - 1) / kUpTW) * ((H + kUpTH - 1) / kUpTH), up_cap = (long)ctx->num_cus * 4;
        AVX_STACK_K_SWITCH(K, hipLaunchKernelGGL(k_barcode_up<KT>, dim3((unsigned)(up_tiles < up_cap ? up_tiles : up_cap)), dim3(kMT), cap_floats * sizeof(float), s, b, cap_floats))
    } else {
        hipLaunchKernelGGL(k_barcode, dim3(g), dim3(kMT), 0, s, b);
    }
    // 10) scanlines: rows[y] broadcast along x and blurred -- a function of (H, W, rows, taps) only, not of the frame: built once per
    //     workspace and kept (two launches fewer per frame); the gain itself rides in the blend kernel below
    const float* rr = nullptr;
    if (d->scan_row_gain != 0.f) {
        uint64_t key = 0xcbf29ce484222325ull;
        auto mixb = [&](const void* ptr, size_t bytes) { const unsigned char* q = (const unsigned char*)ptr; for (size_t i = 0; i < bytes; ++i) key = (key ^ q[i]) * 0x100000001b3ull; };
        const int dims[3] = {H, W, d->scan_ksize};
        mixb(dims, sizeof(dims));
        mixb(d->rows_host, sizeof(float) * H);
        if (d->scan_ksize > 0) mixb(d->scan_taps_host, sizeof(double) * d->scan_ksize);
        key |= 1;  // never 0 (the "nothing cached" value)
        if (ws->scan_key != key || ws->scan_cap < n) {
            if (ws->scan_cap < n) {
                if (ws->d_scan) { AVX_HIP(ctx, hipStreamSynchronize(s)); AVX_HIP(ctx, hipFree(ws->d_scan)); }
                ws->d_scan = nullptr; ws->scan_cap = 0;
                AVX_HIP(ctx, hipMalloc((void**)&ws->d_scan, sizeof(float) * n));
                ws->scan_cap = n;
            }
            hipLaunchKernelGGL(k_rows_plane, dim3(g), dim3(kMT), 0, s, drows, H, W, d->scan_ksize > 0 ? rowsp : ws->d_scan);
            if (d->scan_ksize > 0) { if ((rc = avx_uv_plane_blur_device(ctx, rowsp, ws->d_scan, 1, H, W, d->scan_ksize, d->scan_taps_host, s))) return rc; }
            ws->scan_key = key;
        }
        rr = ws->d_scan;
    }
    const int do_unsharp = d->unsharp_ksize > 0 && d->unsharp_amount > 0.f;
    const int do_periph = d->periph_ksize > 0;
    // 7-12) the finishing stages.  Fused route (round 3): three launches, each a blur with its neighbours as prologue / epilogue (k_mantis_fused_blur); the
    // seven-launch route below serves the parameter sets the fused kernels are not instantiated for, and AVX_MANTIS_FUSE=0.
    const char* fuse_env = getenv("AVX_MANTIS_FUSE");  // read per call: tests flip it
    const bool fused = !(fuse_env && fuse_env[0] == '0') && do_unsharp && fuse_radius_ok(d->unsharp_ksize) && do_periph && fuse_radius_ok(d->periph_ksize) &&
                       (d->pre_soft_ksize <= 0 || fuse_radius_ok(d->pre_soft_ksize));
    if (fused) {
        FuseArgs f{};
        f.H = H; f.W = W;
        f.red_keep = d->red_keep; f.haze = d->haze; f.haze_keep = d->haze_keep;
        for (int c = 0; c < 3; ++c) f.tint[c] = d->haze_tint[c];
        auto set_taps = [&](int ksize, const double* taps) { for (int i = 0; i < ksize; ++i) f.taps[i] = (float)taps[i]; };
        const float* render_src = nullptr;  // NULL: MODE 2 reads prep_render(baseline) itself
        if (d->pre_soft_ksize > 0) {
            f.lin = baseline; f.in = nullptr; f.out = P1;
            set_taps(d->pre_soft_ksize, d->pre_soft_taps_host);
            if ((rc = launch_mantis_fused<1>(ctx, f, d->pre_soft_ksize / 2, s))) return rc;
            render_src = P1;
        }
        f.lin = render_src ? nullptr : baseline; f.in = render_src; f.out = P2;
        f.bar = bar; f.broad = broad; f.rows = rr;
        f.cos2g = d->cos2_global; f.sin2g = d->sin2_global; f.mix = d->orientation_mix; f.lin_s = d->pol_linear_strength; f.lin_gamma = d->pol_linear_gamma;
        f.circ_s = d->pol_circular_strength; f.amount = d->unsharp_amount; f.opacity = d->barcode_opacity; f.scan_gain = d->scan_row_gain;
        set_taps(d->unsharp_ksize, d->unsharp_taps_host);
        if ((rc = launch_mantis_fused<2>(ctx, f, d->unsharp_ksize / 2, s))) return rc;
        f.lin = nullptr; f.in = P2; f.out = nullptr;
        {   // the periphery weight plane: a function of (H, W, xx, yy, softness, radius) only -- built once per workspace and kept, like the scanline plane
            uint64_t key = 0xcbf29ce484222325ull;
            auto mixb = [&](const void* ptr, size_t bytes) { const unsigned char* q = (const unsigned char*)ptr; for (size_t i = 0; i < bytes; ++i) key = (key ^ q[i]) * 0x100000001b3ull; };
            const int dims[2] = {H, W};
            const float prm[2] = {d->periph_softness, d->periph_radius};
            mixb(dims, sizeof(dims)); mixb(prm, sizeof(prm)); mixb(d->xx_host, sizeof(float) * W); mixb(d->yy_host, sizeof(float) * H);
            key |= 1;
            if (ws->periph_key != key || ws->periph_cap < n) {
                if (ws->periph_cap < n) {
                    if (ws->d_periph) { AVX_HIP(ctx, hipStreamSynchronize(s)); AVX_HIP(ctx, hipFree(ws->d_periph)); }
                    ws->d_periph = nullptr; ws->periph_cap = 0;
                    AVX_HIP(ctx, hipMalloc((void**)&ws->d_periph, sizeof(float) * n));
                    ws->periph_cap = n;
                }
                hipLaunchKernelGGL(k_periph_plane, dim3(g), dim3(kMT), 0, s, dxx, dyy, H, W, d->periph_softness, d->periph_radius, ws->d_periph);
                ws->periph_key = key;
            }
            f.tplane = ws->d_periph;
        }
        f.thr = ctx->d_enc_thr_f32; f.coarse = ctx->d_coarse_f32; f.lo_key = ctx->coarse_lo_key[0]; f.out_u8 = out_hwc; f.out_f = out_f;
        set_taps(d->periph_ksize, d->periph_taps_host);
        if ((rc = launch_mantis_fused<3>(ctx, f, d->periph_ksize / 2, s))) return rc;
        return AVX_OK;
    }
    // 7) clear-water look
    hipLaunchKernelGGL(k_prep_render, dim3(g), dim3(kMT), 0, s, baseline, n, d->red_keep, d->haze, d->haze_keep, make_float3(d->haze_tint[0], d->haze_tint[1], d->haze_tint[2]), P0);
    float* render = P0;
    if (d->pre_soft_ksize > 0) { if ((rc = avx_uv_plane_blur_device(ctx, P0, P1, 3, H, W, d->pre_soft_ksize, d->pre_soft_taps_host, s))) return rc; render = P1; }
    // 8) polarisation gain (the Sobel pair of `broad` is formed inside k_polgain)
    PolArgs pa{gx, gy, n, d->cos2_global, d->sin2_global, d->orientation_mix, d->pol_linear_strength, d->pol_linear_gamma, d->pol_circular_strength, gain, broad, H, W};
    hipLaunchKernelGGL(k_polgain, dim3(g), dim3(kMT), 0, s, pa);
    float* other = render == P0 ? P1 : P0;
    if (do_unsharp) { if ((rc = avx_uv_plane_blur_device(ctx, render, other, 3, H, W, d->unsharp_ksize, d->unsharp_taps_host, s))) return rc; }
    // 9) barcode blend (+ the scanline gain)
    hipLaunchKernelGGL(k_unsharp_blend, dim3(grid_for(ctx, 3 * n)), dim3(kMT), 0, s, render, other, gain, bar, n, d->unsharp_amount, do_unsharp, d->barcode_opacity, P2, rr,
                       d->scan_row_gain);
    // 11-12) periphery + encode
    if (do_periph) { if ((rc = avx_uv_plane_blur_device(ctx, P2, P0, 3, H, W, d->periph_ksize, d->periph_taps_host, s))) return rc; }
    FinishArgs fa{P2, P0, H, W, dxx, dyy, d->periph_softness, d->periph_radius, do_periph, ctx->d_enc_thr_f32, ctx->d_coarse_f32, ctx->coarse_lo_key[0], out_hwc, out_f};
    hipLaunchKernelGGL(k_finish, dim3(g), dim3(kMT), 0, s, fa);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}
