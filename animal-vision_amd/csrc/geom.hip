// csrc/geom.hip -- geometric resampling for gfx950: the cv2.resize / cv2.remap /
// cv2.Sobel call sites on either side of the hot path:
//   uv_helpers.py:57-64 resize_preserve_range, :84-99 panorama_warp (INTER_CUBIC), :155-183
//   classic_rgb_to_hsi_scaled (INTER_AREA down, INTER_LINEAR up); animals/cat_widevision_utils.py:11-29
//   center_zoom (uint8 INTER_LINEAR), :46-99 animal_fov_binocular_warp (2x remap INTER_LINEAR BORDER_CONSTANT +
//   cos^2 blend); animals/mantis_shrimp.py:122-131 Sobel ksize 3.
// OpenCV semantics restated from its published algorithm (same statement as oracle/cvref.cpp: PARITY
// UNPINNED); plain float arithmetic in source order (-ffp-contract=off): bit-exact with the oracle.
// Coefficient tables are O(W + H) and built on the host inside the entry points.
#include <cfloat>
#include <cmath>
#include <vector>

#include "dichromat_common.h"
#include "stack_up.h"

using namespace avxk;

namespace {

constexpr int kGT = 256;

struct AxisLin { int* ofs; float* f; int dmax; };
struct AxisCub { int* idx; float* a; };                 // [d][4]
struct AxisArea { int* start; int* cnt; float* alpha; int maxcnt; };  // alpha [d][maxcnt]

// One thread per DESTINATION PIXEL (all C channels: the weights and the four source addresses are shared; 16-byte loads / stores when C is a multiple of 4 and the
// pointers allow).  Round 2's form was one thread per output element with three 64-bit divisions each (i % C, (i / C) % Wd, i / (C Wd): ~300 vector instructions
// around a 9-operation lerp): 28 us for the 1080p x 4-band stack of the UV species, 1.2 TB/s written.  Same statement per element (stack_lerp): identical values.
__global__ __launch_bounds__(kGT) void k_resize_linear_f32(const float* __restrict__ src, int H, int W, int C, float* __restrict__ dst, int Hd, int Wd,
                                                           AxisLin ax, AxisLin ay) {
    const unsigned npx = (unsigned)Hd * (unsigned)Wd;  // the host checks Hd * Wd < 2^31
    const bool vec = (C & 3) == 0 && ((((uintptr_t)src | (uintptr_t)dst) & 15u) == 0);
    for (unsigned p = blockIdx.x * kGT + threadIdx.x; p < npx; p += gridDim.x * kGT) {
        const unsigned y = p / (unsigned)Wd, x = p - y * (unsigned)Wd;
        const int sx = ax.ofs[x], sy0 = ay.ofs[y], sy1 = sy0 + 1 < H ? sy0 + 1 : sy0;
        const float a1 = ax.f[x], a0 = 1.f - a1, b1 = ay.f[y], b0 = 1.f - b1;
        const float* S0 = src + ((size_t)sy0 * W + sx) * C;
        const float* S1 = src + ((size_t)sy1 * W + sx) * C;
        const bool inner = (int)x < ax.dmax;
        const int o = inner ? C : 0;
        float* D = dst + (size_t)p * C;
        if (vec) {
            for (int c = 0; c < C; c += 4) {
                const float4 p00 = *reinterpret_cast<const float4*>(S0 + c), p01 = *reinterpret_cast<const float4*>(S0 + o + c);
                const float4 p10 = *reinterpret_cast<const float4*>(S1 + c), p11 = *reinterpret_cast<const float4*>(S1 + o + c);
                float4 r;
                r.x = stack_lerp(p00.x, p01.x, p10.x, p11.x, a0, a1, b0, b1, inner);  // stack_up.h: the one statement of this arithmetic
                r.y = stack_lerp(p00.y, p01.y, p10.y, p11.y, a0, a1, b0, b1, inner);
                r.z = stack_lerp(p00.z, p01.z, p10.z, p11.z, a0, a1, b0, b1, inner);
                r.w = stack_lerp(p00.w, p01.w, p10.w, p11.w, a0, a1, b0, b1, inner);
                *reinterpret_cast<float4*>(D + c) = r;
            }
        } else {
            for (int c = 0; c < C; ++c) D[c] = stack_lerp(S0[c], S0[o + c], S1[c], S1[o + c], a0, a1, b0, b1, inner);
        }
    }
}

__global__ __launch_bounds__(kGT) void k_resize_linear_u8(const uint8_t* __restrict__ src, int H, int W, int C, uint8_t* __restrict__ dst, int Hd, int Wd,
                                                          AxisLin ax, AxisLin ay) {
    const size_t total = (size_t)Hd * Wd * C;
    for (size_t i = (size_t)blockIdx.x * kGT + threadIdx.x; i < total; i += (size_t)gridDim.x * kGT) {
        const int c = (int)(i % C);
        const int x = (int)((i / C) % Wd), y = (int)(i / ((size_t)C * Wd));
        const int sx = ax.ofs[x], sy0 = ay.ofs[y], sy1 = sy0 + 1 < H ? sy0 + 1 : sy0;
        const int a0 = __float2int_rn((1.f - ax.f[x]) * 2048.f), a1 = __float2int_rn(ax.f[x] * 2048.f);  // saturate_cast<short>: |v| <= 2048
        const int b0 = __float2int_rn((1.f - ay.f[y]) * 2048.f), b1 = __float2int_rn(ay.f[y] * 2048.f);
        const uint8_t* S0 = src + ((size_t)sy0 * W + sx) * C + c;
        const uint8_t* S1 = src + ((size_t)sy1 * W + sx) * C + c;
        int r0, r1;
        if (x < ax.dmax) { r0 = S0[0] * a0 + S0[C] * a1; r1 = S1[0] * a0 + S1[C] * a1; }
        else { r0 = S0[0] * 2048; r1 = S1[0] * 2048; }
        int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
        dst[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

__global__ __launch_bounds__(kGT) void k_resize_cubic_f32(const float* __restrict__ src, int H, int W, int C, float* __restrict__ dst, int Hd, int Wd,
                                                          AxisCub ax, AxisCub ay) {
    const size_t total = (size_t)Hd * Wd * C;
    for (size_t i = (size_t)blockIdx.x * kGT + threadIdx.x; i < total; i += (size_t)gridDim.x * kGT) {
        const int c = (int)(i % C);
        const int x = (int)((i / C) % Wd), y = (int)(i / ((size_t)C * Wd));
        const int* ix = ax.idx + 4 * x;
        const float* a = ax.a + 4 * x;
        const int* iy = ay.idx + 4 * y;
        const float* b = ay.a + 4 * y;
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float* S = src + (size_t)iy[k] * W * C + c;
            r[k] = S[(size_t)ix[0] * C] * a[0] + S[(size_t)ix[1] * C] * a[1] + S[(size_t)ix[2] * C] * a[2] + S[(size_t)ix[3] * C] * a[3];
        }
        dst[i] = r[0] * b[0] + r[1] * b[1] + r[2] * b[2] + r[3] * b[3];
    }
}

// panorama_warp (uv_helpers.py:84-99) widens only x: the vertical pass of cv2.resize(INTER_CUBIC) at scale 1 has the coefficients
// (0, 1, 0, 0) exactly, so for FINITE sources r0*0 + r1*1 + r2*0 + r3*0 == r1 and three of the four source rows need not be read
// (12 loads per pixel instead of 48).  One thread per output pixel, the three channels together.
__global__ __launch_bounds__(kGT) void k_panorama_cubic_x(const float* __restrict__ src, int H, int W, float* __restrict__ dst, AxisCub ax) {
    const size_t total = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * kGT + threadIdx.x; i < total; i += (size_t)gridDim.x * kGT) {
        const int x = (int)(i % W), y = (int)(i / W);
        const int* ix = ax.idx + 4 * x;
        const float* a = ax.a + 4 * x;
        const float* S = src + (size_t)y * W * 3;
        const int i0 = ix[0] * 3, i1 = ix[1] * 3, i2 = ix[2] * 3, i3 = ix[3] * 3;
        const float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) dst[i * 3 + c] = S[i0 + c] * a0 + S[i1 + c] * a1 + S[i2 + c] * a2 + S[i3 + c] * a3;
    }
}

// cv::saturate_cast<uchar>(float): cvRound (round half to even), clamped
__device__ __forceinline__ void put_area(float* d, float v) { *d = v; }
__device__ __forceinline__ void put_area(uint8_t* d, float v) { const float r = rintf(v); *d = (uint8_t)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r)); }

// T = float or uint8_t (cv::resize INTER_AREA of a uint8 image: resizeAreaFast_<uchar, int> sums integers -- exact in float32
// below 2^24 -- and stores saturate_cast<uchar>(sum * scale); its 2x2 special case rounds (sum + 2) >> 2 instead)
template <typename T>
__global__ __launch_bounds__(kGT) void k_resize_area_fast_f32(const T* __restrict__ src, int H, int W, int C, T* __restrict__ dst, int Hd, int Wd,
                                                              int isx, int isy) {
    const size_t total = (size_t)Hd * Wd * C;
    const int area = isx * isy;
    const float scale = 1.f / area;
    for (size_t i = (size_t)blockIdx.x * kGT + threadIdx.x; i < total; i += (size_t)gridDim.x * kGT) {
        const int c = (int)(i % C);
        const int x = (int)((i / C) % Wd), y = (int)(i / ((size_t)C * Wd));
        const T* S = src + ((size_t)(y * isy) * W + (size_t)x * isx) * C + c;
        auto at = [&](int k) { const int sy = k / isx, sx = k - sy * isx; return (float)S[((size_t)sy * W + sx) * C]; };
        float sum = 0;
        int k = 0;
        for (; k <= area - 4; k += 4) sum += at(k) + at(k + 1) + at(k + 2) + at(k + 3);  // resizeAreaFast_: groups of four
        for (; k < area; ++k) sum += at(k);
        if (sizeof(T) == 1 && isx == 2 && isy == 2) put_area(dst + i, (float)(((int)sum + 2) >> 2));  // ResizeAreaFastVec, 8-bit 2x2
        else put_area(dst + i, sum * scale);
    }
}

template <typename T>
__global__ __launch_bounds__(kGT) void k_resize_area_f32(const T* __restrict__ src, int H, int W, int C, T* __restrict__ dst, int Hd, int Wd,
                                                         AxisArea ax, AxisArea ay) {
    const size_t total = (size_t)Hd * Wd * C;
    for (size_t i = (size_t)blockIdx.x * kGT + threadIdx.x; i < total; i += (size_t)gridDim.x * kGT) {
        const int c = (int)(i % C);
        const int x = (int)((i / C) % Wd), y = (int)(i / ((size_t)C * Wd));
        const int x0 = ax.start[x], nx = ax.cnt[x], y0 = ay.start[y], ny = ay.cnt[y];
        const float* al = ax.alpha + (size_t)x * ax.maxcnt;
        const float* be = ay.alpha + (size_t)y * ay.maxcnt;
        float sum = 0.f;
        for (int j = 0; j < ny; ++j) {
            const T* S = src + ((size_t)(y0 + j) * W + x0) * C + c;
            float buf = 0.f;
            for (int k = 0; k < nx; ++k) buf += (float)S[(size_t)k * C] * al[k];  // ResizeArea_Invoker: buf[dx] += S*alpha
            sum = j == 0 ? be[j] * buf : sum + be[j] * buf;               // first row of a dy starts the sum
        }
        put_area(dst + i, sum);
    }
}

// "any byte > 1" per frame (a3 quirk, for paths that need the normalised value before their first kernel)
__global__ __launch_bounds__(kGT) void k_any_gt1(const uint8_t* __restrict__ in, size_t nbytes, uint32_t* flag) {
    uint32_t seen = 0;
    for (size_t i = (size_t)blockIdx.x * kGT + threadIdx.x; i < nbytes; i += (size_t)gridDim.x * kGT) seen |= in[i] >> 1;
    if (seen) *flag = 1u;
}

// animal_fov_binocular_warp (cat_widevision_utils.py:46-99) on a uint8 frame: get_normalized_image, two remaps
// (INTER_LINEAR, BORDER_CONSTANT 0, 1/32-px quantised coordinates) and the cos^2 blend, -> float32 HWC in [0,1].
struct WarpArgs { const uint8_t* in; float* out; int H, W, Ho, Wo; const float* xL; const float* xR; const float* ymap; const float* wL; const float* wR;
                  const uint32_t* flag; };

__device__ __forceinline__ void remap_px(const uint8_t* in, int H, int W, float mx, float my, float norm, float (&o)[3]) {
    const int fx = __float2int_rn(mx * 32.f), fy = __float2int_rn(my * 32.f);
    int sx = fx >> 5, sy = fy >> 5;
    sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);
    sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
    const float tx = (fx & 31) * (1.f / 32), ty = (fy & 31) * (1.f / 32);
    const float w0 = (1.f - ty) * (1.f - tx), w1 = (1.f - ty) * tx, w2 = ty * (1.f - tx), w3 = ty * tx;
    if (sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0) { o[0] = o[1] = o[2] = 0.f; return; }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        auto at = [&](int yy, int xx) {
            if ((unsigned)xx >= (unsigned)W || (unsigned)yy >= (unsigned)H) return 0.f;  // BORDER_CONSTANT, value 0
            const float v = (float)in[((size_t)yy * W + xx) * 3 + c];
            const float n = norm == 1.f ? v : v / 255.0f;                                // get_normalized_image
            return n < 0.f ? 0.f : (n > 1.f ? 1.f : n);
        };
        o[c] = at(sy, sx) * w0 + at(sy, sx + 1) * w1 + at(sy + 1, sx) * w2 + at(sy + 1, sx + 1) * w3;
    }
}

__global__ __launch_bounds__(kGT) void k_binocular_warp(WarpArgs a) {
    const float norm = *a.flag ? 255.f : 1.f;  // all bytes <= 1: get_normalized_image does not divide
    const size_t total = (size_t)a.Ho * a.Wo;
    for (size_t i = (size_t)blockIdx.x * kGT + threadIdx.x; i < total; i += (size_t)gridDim.x * kGT) {
        const int x = (int)(i % a.Wo), y = (int)(i / a.Wo);
        float l[3], r[3];
        remap_px(a.in, a.H, a.W, a.xL[x], a.ymap[y], norm, l);
        remap_px(a.in, a.H, a.W, a.xR[x], a.ymap[y], norm, r);
        const float wl = a.wL[x], wr = a.wR[x];
        const float wsum = (wl + wr) + 1e-8f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = (l[c] * wl + r[c] * wr) / wsum;
            a.out[i * 3 + c] = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
        }
    }
}

// cv2.remap(float32 planes, mapx, mapy, INTER_LINEAR, BORDER_CONSTANT, borderValue) with per-pixel float32 maps
// (anableps.py:217-226); K planes share the maps.
__global__ __launch_bounds__(kGT) void k_remap_planes(const float* __restrict__ src, int K, int H, int W, const float* __restrict__ mapx, const float* __restrict__ mapy,
                                                      float* __restrict__ dst, float border) {
    const size_t n = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * kGT + threadIdx.x; i < n; i += (size_t)gridDim.x * kGT) {
        const int fx = __float2int_rn(mapx[i] * 32.f), fy = __float2int_rn(mapy[i] * 32.f);
        int sx = fx >> 5, sy = fy >> 5;
        sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);
        sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
        const float tx = (fx & 31) * (1.f / 32), ty = (fy & 31) * (1.f / 32);
        const float w0 = (1.f - ty) * (1.f - tx), w1 = (1.f - ty) * tx, w2 = ty * (1.f - tx), w3 = ty * tx;
        const bool outside = sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0;
        for (int k = 0; k < K; ++k) {
            const float* S = src + (size_t)k * n;
            auto at = [&](int yy, int xx) { return ((unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H) ? S[(size_t)yy * W + xx] : border; };
            dst[(size_t)k * n + i] = outside ? border : at(sy, sx) * w0 + at(sy, sx + 1) * w1 + at(sy + 1, sx) * w2 + at(sy + 1, sx + 1) * w3;
        }
    }
}

// cv2.resize(..., INTER_NEAREST): sx = min(floor(dx * (W / Wd)), W - 1) with the scale in double (resizeNN)
__global__ __launch_bounds__(kGT) void k_resize_nearest_f32(const float* __restrict__ src, int H, int W, int C, float* __restrict__ dst, int Hd, int Wd, double fx, double fy) {
    const size_t total = (size_t)Hd * Wd * C;
    for (size_t i = (size_t)blockIdx.x * kGT + threadIdx.x; i < total; i += (size_t)gridDim.x * kGT) {
        const int c = (int)(i % C);
        const int x = (int)((i / C) % Wd), y = (int)(i / ((size_t)C * Wd));
        int sx = (int)floor(x * fx), sy = (int)floor(y * fy);
        sx = sx < W - 1 ? sx : W - 1;
        sy = sy < H - 1 ? sy : H - 1;
        dst[i] = src[((size_t)sy * W + sx) * C + c];
    }
}

// make_split_frame (renderers/video.py:229-239) without the labels: left half original, right half modified,
// 1-px white seam at W//2.  16 bytes per thread where the row segment allows it.
__global__ __launch_bounds__(kGT) void k_split_compose(const uint8_t* __restrict__ orig, const uint8_t* __restrict__ mod, uint8_t* __restrict__ out, int H, int W,
                                                       int draw_seam) {
    const size_t row_bytes = (size_t)W * 3, total = (size_t)H * row_bytes;
    const size_t mid_b = (size_t)(W / 2) * 3;
    for (size_t i = ((size_t)blockIdx.x * kGT + threadIdx.x) * 4; i < total; i += (size_t)gridDim.x * kGT * 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t j = i + k;
            if (j >= total) break;
            const size_t xb = j % row_bytes;
            uint8_t v = xb < mid_b ? orig[j] : mod[j];
            if (draw_seam && xb >= mid_b && xb < mid_b + 3) v = 255;
            out[j] = v;
        }
    }
}

// cv2.Sobel(plane, CV_32F, dx, dy, ksize=3, BORDER_REFLECT_101): small-kernel forms, row filter then column filter
__global__ __launch_bounds__(kGT) void k_sobel3(const float* __restrict__ src, int H, int W, float* __restrict__ gx, float* __restrict__ gy) {
    const size_t total = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * kGT + threadIdx.x; i < total; i += (size_t)gridDim.x * kGT) {
        const int x = (int)(i % W), y = (int)(i / W);
        const int xm = reflect101(x - 1, W), xp = reflect101(x + 1, W);
        float d[3], s[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float* R = src + (size_t)reflect101(y - 1 + k, H) * W;
            d[k] = R[xp] - R[xm];
            s[k] = R[xm] + R[x] * 2 + R[xp];
        }
        gx[i] = d[0] + d[1] * 2 + d[2];
        gy[i] = s[2] - s[0];
    }
}

// ---- host-side coefficient tables (same construction as OpenCV's resizeGeneric_ / computeResizeAreaTab) ----
struct HostLin { std::vector<int> ofs; std::vector<float> f; int dmax; };
HostLin host_lin(int ssize, int dsize) {
    HostLin t; t.ofs.resize(dsize); t.f.resize(dsize); t.dmax = dsize;
    const double scale = 1.0 / ((double)dsize / ssize);
    for (int d = 0; d < dsize; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= s;
        if (s < 0) { f = 0; s = 0; }
        if (s + 1 >= ssize) { if (t.dmax > d) t.dmax = d; if (s >= ssize - 1) { f = 0; s = ssize - 1; } }
        t.ofs[d] = s; t.f[d] = f;
    }
    return t;
}
void host_cubic(int ssize, int dsize, std::vector<int>& idx, std::vector<float>& a) {
    idx.resize((size_t)dsize * 4); a.resize((size_t)dsize * 4);
    const double scale = 1.0 / ((double)dsize / ssize);
    const float A = -0.75f;
    for (int d = 0; d < dsize; ++d) {
        float x = (float)((d + 0.5) * scale - 0.5);
        const int s = (int)std::floor(x);
        x -= s;
        float* c = &a[(size_t)d * 4];
        c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
        c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
        c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
        c[3] = 1.f - c[0] - c[1] - c[2];
        for (int k = 0; k < 4; ++k) { int i = s - 1 + k; idx[(size_t)d * 4 + k] = i < 0 ? 0 : (i >= ssize ? ssize - 1 : i); }
    }
}
void host_area(int ssize, int dsize, std::vector<int>& start, std::vector<int>& cnt, std::vector<float>& alpha, int& maxcnt) {
    const double scale = (double)ssize / dsize;
    std::vector<std::vector<std::pair<int, float>>> ent(dsize);
    maxcnt = 1;
    for (int dx = 0; dx < dsize; ++dx) {
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = std::fmin(scale, ssize - fsx1);
        int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
        sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
        sx1 = sx1 < sx2 ? sx1 : sx2;
        if (sx1 - fsx1 > 1e-3) ent[dx].push_back({sx1 - 1, (float)((sx1 - fsx1) / cell)});
        for (int sx = sx1; sx < sx2; ++sx) ent[dx].push_back({sx, (float)(1.0 / cell)});
        if (fsx2 - sx2 > 1e-3) ent[dx].push_back({sx2, (float)(std::fmin(std::fmin(fsx2 - sx2, 1.), cell) / cell)});
        if ((int)ent[dx].size() > maxcnt) maxcnt = (int)ent[dx].size();
    }
    start.assign(dsize, 0); cnt.assign(dsize, 0); alpha.assign((size_t)dsize * maxcnt, 0.f);
    for (int dx = 0; dx < dsize; ++dx) {
        cnt[dx] = (int)ent[dx].size();
        start[dx] = cnt[dx] ? ent[dx][0].first : 0;
        for (int k = 0; k < cnt[dx]; ++k) alpha[(size_t)dx * maxcnt + k] = ent[dx][k].second;  // entries are consecutive source indices
    }
}

int grid_for(avx_ctx* ctx, size_t items) {
    const size_t want = (items + kGT - 1) / kGT, cap = (size_t)ctx->num_cus * 16;
    return (int)(want < cap ? (want ? want : 1) : cap);
}

// Device copies of the O(W + H) coefficient tables, cached per stream workspace under (kind, component, source size,
// target size): built and uploaded (blocking) the first time a geometry is seen, immutable afterwards, so replaying a
// resize costs no host work and no stream synchronisation.
enum { TK_LIN = 1, TK_CUB = 2, TK_AREA = 3 };
struct TableCache {
    avx_ctx* ctx; avx_ws* ws; hipStream_t s;
    static uint64_t key(int kind, int comp, int ssize, int dsize) { return ((uint64_t)kind << 58) | ((uint64_t)comp << 54) | ((uint64_t)ssize << 27) | (uint64_t)dsize; }
    void* find(uint64_t k) const {
        for (int i = 0; i < ws->n_geom_tabs; ++i)
            if (ws->geom_tabs[i].key == k) return ws->geom_tabs[i].dev;
        return nullptr;
    }
    int room() {  // once per entry point, before any lookup: nearly full -> drop everything (launches may still read the tables)
        if (ws->n_geom_tabs <= 48) return AVX_OK;
        AVX_HIP(ctx, hipStreamSynchronize(s));
        for (int i = 0; i < ws->n_geom_tabs; ++i)
            if (((ws->geom_tabs[i].key >> 54) & 0xf) < 8) (void)hipFree(ws->geom_tabs[i].dev);
        ws->n_geom_tabs = 0;
        return AVX_OK;
    }
    template <typename T> int put(uint64_t k, const std::vector<T>& v, T** out) {
        if (ws->n_geom_tabs >= 64) return avx_fail(ctx, AVX_ERR_NOMEM, "geometry table cache overflow");
        void* d = nullptr;
        AVX_HIP(ctx, hipMalloc(&d, v.size() * sizeof(T) + 256));
        if (!v.empty()) AVX_HIP(ctx, hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        ws->geom_tabs[ws->n_geom_tabs++] = {k, d};
        *out = (T*)d;
        return AVX_OK;
    }
    int lin(int ssize, int dsize, AxisLin* ax) {
        const uint64_t k0 = key(TK_LIN, 0, ssize, dsize), k1 = key(TK_LIN, 1, ssize, dsize), k2 = key(TK_LIN, 8, ssize, dsize);  // component >= 8: a scalar in the pointer slot
        if (void* d = find(k0)) {
            ax->ofs = (int*)d; ax->f = (float*)find(k1);
            ax->dmax = (int)(intptr_t)find(k2) - 1;
            return AVX_OK;
        }
        HostLin h = host_lin(ssize, dsize);
        int rc;
        if ((rc = put(k0, h.ofs, &ax->ofs)) || (rc = put(k1, h.f, &ax->f))) return rc;
        ax->dmax = h.dmax;
        if (ws->n_geom_tabs < 64) ws->geom_tabs[ws->n_geom_tabs++] = {k2, (void*)(intptr_t)(h.dmax + 1)};  // a scalar rides in the pointer slot
        else return avx_fail(ctx, AVX_ERR_NOMEM, "geometry table cache overflow");
        return AVX_OK;
    }
    int cub(int ssize, int dsize, AxisCub* ax) {
        const uint64_t k0 = key(TK_CUB, 0, ssize, dsize), k1 = key(TK_CUB, 1, ssize, dsize);
        if (void* d = find(k0)) { ax->idx = (int*)d; ax->a = (float*)find(k1); return AVX_OK; }
        std::vector<int> ix; std::vector<float> cx;
        host_cubic(ssize, dsize, ix, cx);
        int rc;
        if ((rc = put(k0, ix, &ax->idx)) || (rc = put(k1, cx, &ax->a))) return rc;
        return AVX_OK;
    }
    int area(int ssize, int dsize, AxisArea* ax) {
        const uint64_t k0 = key(TK_AREA, 0, ssize, dsize), k1 = key(TK_AREA, 1, ssize, dsize), k2 = key(TK_AREA, 2, ssize, dsize), k3 = key(TK_AREA, 8, ssize, dsize);
        if (void* d = find(k0)) {
            ax->start = (int*)d; ax->cnt = (int*)find(k1); ax->alpha = (float*)find(k2); ax->maxcnt = (int)(intptr_t)find(k3);
            return AVX_OK;
        }
        std::vector<int> sv, cv; std::vector<float> av; int m = 1;
        host_area(ssize, dsize, sv, cv, av, m);
        int rc;
        if ((rc = put(k0, sv, &ax->start)) || (rc = put(k1, cv, &ax->cnt)) || (rc = put(k2, av, &ax->alpha))) return rc;
        ax->maxcnt = m;
        if (ws->n_geom_tabs < 64) ws->geom_tabs[ws->n_geom_tabs++] = {k3, (void*)(intptr_t)m};
        else return avx_fail(ctx, AVX_ERR_NOMEM, "geometry table cache overflow");
        return AVX_OK;
    }
};

}  // namespace

// Drop every cached table when more than `limit` slots are taken (callers that look a table up and then call other
// entry points trim FIRST, so that nothing they hold is evicted under them).
int avx_geom_cache_trim(avx_ctx* ctx, avx_ws* ws, hipStream_t s, int limit) {
    if (ws->n_geom_tabs <= limit) return AVX_OK;
    AVX_HIP(ctx, hipStreamSynchronize(s));  // launches may still be reading the tables
    for (int i = 0; i < ws->n_geom_tabs; ++i)
        if (((ws->geom_tabs[i].key >> 54) & 0xf) < 8) (void)hipFree(ws->geom_tabs[i].dev);
    ws->n_geom_tabs = 0;
    return AVX_OK;
}

// uv_helpers.panorama_warp (:84-99): INTER_CUBIC widen to newW, then the centre crop [start, start+W): only the
// cropped columns are computed (the x table is sliced); the height is unchanged (cubic taps 0,1,0,0: identity).
// finite_src: the source is known to hold finite values (decoded uint8 frames): the identity vertical pass is skipped
int avx_geom_panorama_cubic(avx_ctx* ctx, const float* src, int H, int W, int newW, int start, float* dst, hipStream_t s, bool finite_src) {
    avx_ws* ws = avx_workspace(ctx, s);
    if (!ws) return AVX_ERR_NOMEM;
    TableCache tc{ctx, ws, s};
    AxisCub ax{}, ay{};
    int rc;
    if ((rc = tc.room())) return rc;
    if ((rc = tc.cub(W, newW, &ax)) || (rc = tc.cub(H, H, &ay))) return rc;
    ax.idx += 4 * (size_t)start;
    ax.a += 4 * (size_t)start;
    if (finite_src && !getenv("AVX_PANO_FULL")) hipLaunchKernelGGL(k_panorama_cubic_x, dim3(grid_for(ctx, (size_t)H * W)), dim3(kGT), 0, s, src, H, W, dst, ax);
    else hipLaunchKernelGGL(k_resize_cubic_f32, dim3(grid_for(ctx, (size_t)H * W * 3)), dim3(kGT), 0, s, src, H, W, 3, dst, H, W, ax, ay);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

// The INTER_CUBIC x tables of panorama_warp (W -> newW, cropped from column `start`), for mantis.hip's fused front end.
int avx_geom_cubic_x_tables(avx_ctx* ctx, avx_ws* ws, hipStream_t s, int W, int newW, int start, const int** idx, const float** a) {
    TableCache tc{ctx, ws, s};
    int rc = tc.room();
    if (rc) return rc;
    AxisCub ax{};
    if ((rc = tc.cub(W, newW, &ax))) return rc;
    *idx = ax.idx + 4 * (size_t)start;
    *a = ax.a + 4 * (size_t)start;
    return AVX_OK;
}

// The INTER_LINEAR tables of an H x W -> Hd x Wd resize, for kernels that recompute resized values instead of reading a
// materialised copy (stack_up.h); same cache, same lifetime rules as avx_resize_hwc's own lookups.
int avx_geom_linear_tables(avx_ctx* ctx, avx_ws* ws, hipStream_t s, int H, int W, int Hd, int Wd, avx_lin_tab* ax, avx_lin_tab* ay) {
    TableCache tc{ctx, ws, s};
    int rc = tc.room();
    if (rc) return rc;
    AxisLin x{}, y{};
    if ((rc = tc.lin(W, Wd, &x)) || (rc = tc.lin(H, Hd, &y))) return rc;
    *ax = avx_lin_tab{x.ofs, x.f, x.dmax};
    *ay = avx_lin_tab{y.ofs, y.f, y.dmax};
    return AVX_OK;
}

extern "C" {

int avx_resize_hwc(avx_ctx* ctx, const void* src, int dtype, int H, int W, int C, void* dst, int Hd, int Wd, int interp, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, src && dst && src != dst && H > 0 && W > 0 && C > 0 && Hd > 0 && Wd > 0, "avx_resize_hwc: bad arguments");
    AVX_REQUIRE(ctx, dtype == 0 || dtype == 2, "avx_resize_hwc: dtype must be 0 (float32) or 2 (uint8)");
    AVX_REQUIRE(ctx, interp >= 0 && interp <= 3, "avx_resize_hwc: interpolation must be 0 NEAREST, 1 LINEAR, 2 CUBIC or 3 AREA");
    AVX_REQUIRE(ctx, dtype == 0 || interp == 1 || interp == 3, "avx_resize_hwc: uint8 supports INTER_LINEAR and INTER_AREA only");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    if (interp == 0) {  // no tables
        hipLaunchKernelGGL(k_resize_nearest_f32, dim3(grid_for(ctx, (size_t)Hd * Wd * C)), dim3(kGT), 0, s, (const float*)src, H, W, C, (float*)dst, Hd, Wd,
                           1.0 / ((double)Wd / W), 1.0 / ((double)Hd / H));  // resizeNN: ifx = 1 / inv_scale_x
        AVX_HIP(ctx, hipGetLastError());
        return AVX_OK;
    }
    avx_ws* ws = avx_workspace(ctx, s);
    if (!ws) return AVX_ERR_NOMEM;
    TableCache tc{ctx, ws, s};
    const size_t total = (size_t)Hd * Wd * C;
    const int g = grid_for(ctx, total);
    int rc = tc.room();
    if (rc) return rc;
    if (interp == 3 && (Wd > W || Hd > H)) interp = 1;  // cv::resize: INTER_AREA when enlarging is INTER_LINEAR
    if (interp == 1) {
        AxisLin ax{}, ay{};
        if ((rc = tc.lin(W, Wd, &ax)) || (rc = tc.lin(H, Hd, &ay))) return rc;
        if (dtype == 0) {
            AVX_REQUIRE(ctx, (size_t)Hd * Wd < ((size_t)1 << 31), "avx_resize_hwc: destination larger than 2^31 pixels");
            hipLaunchKernelGGL(k_resize_linear_f32, dim3(grid_for(ctx, (size_t)Hd * Wd)), dim3(kGT), 0, s, (const float*)src, H, W, C, (float*)dst, Hd, Wd, ax, ay);  // one thread per destination pixel
        }
        else hipLaunchKernelGGL(k_resize_linear_u8, dim3(g), dim3(kGT), 0, s, (const uint8_t*)src, H, W, C, (uint8_t*)dst, Hd, Wd, ax, ay);
    } else if (interp == 2) {
        AxisCub ax{}, ay{};
        if ((rc = tc.cub(W, Wd, &ax)) || (rc = tc.cub(H, Hd, &ay))) return rc;
        hipLaunchKernelGGL(k_resize_cubic_f32, dim3(g), dim3(kGT), 0, s, (const float*)src, H, W, C, (float*)dst, Hd, Wd, ax, ay);
    } else {
        const double sx = (double)W / Wd, sy = (double)H / Hd;
        const int isx = (int)std::lrint(sx), isy = (int)std::lrint(sy);
        if (std::fabs(sx - isx) < DBL_EPSILON && std::fabs(sy - isy) < DBL_EPSILON) {
            if (dtype == 0) hipLaunchKernelGGL(k_resize_area_fast_f32<float>, dim3(g), dim3(kGT), 0, s, (const float*)src, H, W, C, (float*)dst, Hd, Wd, isx, isy);
            else hipLaunchKernelGGL(k_resize_area_fast_f32<uint8_t>, dim3(g), dim3(kGT), 0, s, (const uint8_t*)src, H, W, C, (uint8_t*)dst, Hd, Wd, isx, isy);
        } else {
            AxisArea ax{}, ay{};
            if ((rc = tc.area(W, Wd, &ax)) || (rc = tc.area(H, Hd, &ay))) return rc;
            if (dtype == 0) hipLaunchKernelGGL(k_resize_area_f32<float>, dim3(g), dim3(kGT), 0, s, (const float*)src, H, W, C, (float*)dst, Hd, Wd, ax, ay);
            else hipLaunchKernelGGL(k_resize_area_f32<uint8_t>, dim3(g), dim3(kGT), 0, s, (const uint8_t*)src, H, W, C, (uint8_t*)dst, Hd, Wd, ax, ay);
        }
    }
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

int avx_binocular_warp_u8(avx_ctx* ctx, const uint8_t* in_hwc, int H, int W, const float* xL_host, const float* xR_host, const float* ymap_host,
                          const float* wL_host, const float* wR_host, int Ho, int Wo, float* out_hwc_f32, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, in_hwc && out_hwc_f32 && xL_host && xR_host && ymap_host && wL_host && wR_host && H > 0 && W > 0 && Ho > 0 && Wo > 0,
                "avx_binocular_warp_u8: bad arguments");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    avx_ws* ws = avx_workspace(ctx, s);
    if (!ws) return AVX_ERR_NOMEM;
    const size_t tab_bytes = ((size_t)(4 * Wo + Ho) + 1024) * 4 + 4096;
    if (ws->geom_cap < tab_bytes) {
        if (ws->d_geom) { AVX_HIP(ctx, hipStreamSynchronize(s)); AVX_HIP(ctx, hipFree(ws->d_geom)); }
        ws->d_geom = nullptr; ws->geom_cap = 0;
        AVX_HIP(ctx, hipMalloc(&ws->d_geom, tab_bytes));
        ws->geom_cap = tab_bytes;
    }
    AVX_HIP(ctx, hipStreamSynchronize(s));
    float* t = (float*)ws->d_geom;
    WarpArgs a{};
    a.in = in_hwc; a.out = out_hwc_f32; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
    float* p = t;
    auto up = [&](const float* h, int n, const float** d) { *d = p; hipError_t e = hipMemcpyAsync(p, h, sizeof(float) * n, hipMemcpyHostToDevice, s); p += (n + 63) & ~63; return e; };
    AVX_HIP(ctx, up(xL_host, Wo, &a.xL)); AVX_HIP(ctx, up(xR_host, Wo, &a.xR)); AVX_HIP(ctx, up(wL_host, Wo, &a.wL)); AVX_HIP(ctx, up(wR_host, Wo, &a.wR));
    AVX_HIP(ctx, up(ymap_host, Ho, &a.ymap));
    uint32_t* flag = (uint32_t*)p;
    a.flag = flag;
    AVX_HIP(ctx, hipMemsetAsync(flag, 0, 4, s));
    AVX_HIP(ctx, hipStreamSynchronize(s));
    const size_t nbytes = (size_t)H * W * 3;
    hipLaunchKernelGGL(k_any_gt1, dim3(grid_for(ctx, nbytes)), dim3(kGT), 0, s, in_hwc, nbytes, flag);
    hipLaunchKernelGGL(k_binocular_warp, dim3(grid_for(ctx, (size_t)Ho * Wo)), dim3(kGT), 0, s, a);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

int avx_split_compose_u8(avx_ctx* ctx, const uint8_t* original_hwc, const uint8_t* modified_hwc, uint8_t* out_hwc, int H, int W, int draw_seam, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, original_hwc && modified_hwc && out_hwc && H > 0 && W > 0, "avx_split_compose_u8: bad arguments");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    hipLaunchKernelGGL(k_split_compose, dim3(grid_for(ctx, (size_t)H * W * 3 / 4 + 1)), dim3(kGT), 0, s, original_hwc, modified_hwc, out_hwc, H, W, draw_seam);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

int avx_remap_linear_planes(avx_ctx* ctx, const float* src_planes, int K, int H, int W, const float* mapx, const float* mapy, float* dst_planes,
                            float border_value, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, src_planes && mapx && mapy && dst_planes && K >= 1 && H > 0 && W > 0, "avx_remap_linear_planes: bad arguments");
    AVX_REQUIRE(ctx, src_planes != dst_planes, "avx_remap_linear_planes: in-place remap is not possible");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    hipLaunchKernelGGL(k_remap_planes, dim3(grid_for(ctx, (size_t)H * W)), dim3(kGT), 0, s, src_planes, K, H, W, mapx, mapy, dst_planes, border_value);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

int avx_sobel3_plane(avx_ctx* ctx, const float* plane, int H, int W, float* gx, float* gy, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, plane && gx && gy && H > 0 && W > 0, "avx_sobel3_plane: bad arguments");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    hipLaunchKernelGGL(k_sobel3, dim3(grid_for(ctx, (size_t)H * W)), dim3(kGT), 0, s, plane, H, W, gx, gy);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

}  // extern "C"
