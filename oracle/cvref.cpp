// oracle/cvref.cpp -- TEST INFRASTRUCTURE ONLY (the checker, never the product).
//
// OpenCV resize / remap / Sobel semantics, restated from OpenCV 4.x's published algorithm
// (imgproc/src/resize.cpp: resizeGeneric_, HResizeLinear/VResizeLinear, HResizeCubic/VResizeCubic,
// resizeAreaFast_, computeResizeAreaTab + ResizeArea_Invoker; imgwarp.cpp: remap -> remapBilinear with
// INTER_BITS = 5; deriv.cpp + filter.simd.hpp: Sobel ksize 3 small-kernel paths).
// OpenCV is neither under /root/reference nor installed anywhere in this pipeline (unpinned dependency
// `opencv-contrib-python`, requirements.txt:16): PARITY UNPINNED for every function here.
// Reference call sites: uv_helpers.py:63 (resize_preserve_range), :94 (panorama_warp, INTER_CUBIC),
// :172,:182 (INTER_AREA down / INTER_LINEAR up); animals/cat_widevision_utils.py:26 (center_zoom, uint8
// INTER_LINEAR), :89,:91 (remap INTER_LINEAR BORDER_CONSTANT); animals/mantis_shrimp.py:124-125 (Sobel).
// Plain float arithmetic in source order (compiled with -ffp-contract=off): a*b + c*d is mul, mul, add.
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <vector>

namespace {

inline int cv_floor(double v) { return (int)std::floor(v); }
inline int cv_round(double v) { return (int)std::lrint(v); }  // round half to even (default FP mode)
inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

// one axis of resizeGeneric_ for INTER_LINEAR: source offset + (1-f, f)
struct LinAxis { std::vector<int> ofs; std::vector<float> f; int dmax; };
LinAxis linear_axis(int ssize, int dsize) {
    LinAxis t; t.ofs.resize(dsize); t.f.resize(dsize); t.dmax = dsize;
    const double inv = (double)dsize / ssize, scale = 1.0 / inv;
    for (int d = 0; d < dsize; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = cv_floor(f);
        f -= s;
        if (s < 0) { f = 0; s = 0; }
        if (s + 1 >= ssize) { if (t.dmax > d) t.dmax = d; if (s >= ssize - 1) { f = 0; s = ssize - 1; } }
        t.ofs[d] = s; t.f[d] = f;
    }
    return t;
}

void cubic_coeffs(float x, float* c) {  // interpolateCubic, A = -0.75
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

}  // namespace

extern "C" {

// cv::resize(float HWC, INTER_LINEAR): D = S[s]*(1-f) + S[s+1]*f per row, then the same between rows.
void cvref_resize_linear_f32(const float* src, int H, int W, int C, float* dst, int Hd, int Wd) {
    const LinAxis tx = linear_axis(W, Wd), ty = linear_axis(H, Hd);
    std::vector<float> rows((size_t)H * Wd * C);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < Wd; ++x) {
            const int s0 = tx.ofs[x];
            const float a0 = 1.f - tx.f[x], a1 = tx.f[x];
            for (int c = 0; c < C; ++c) {
                const float* S = src + ((size_t)y * W + s0) * C + c;
                rows[((size_t)y * Wd + x) * C + c] = x < tx.dmax ? S[0] * a0 + S[C] * a1 : S[0] * 1.f;
            }
        }
    for (int y = 0; y < Hd; ++y) {
        const int s0 = ty.ofs[y], s1 = s0 + 1 < H ? s0 + 1 : s0;
        const float b0 = 1.f - ty.f[y], b1 = ty.f[y];
        const float* R0 = rows.data() + (size_t)s0 * Wd * C;
        const float* R1 = rows.data() + (size_t)s1 * Wd * C;
        for (size_t i = 0; i < (size_t)Wd * C; ++i) dst[(size_t)y * Wd * C + i] = R0[i] * b0 + R1[i] * b1;
    }
}

// cv::resize(uint8 HWC, INTER_LINEAR): 11-bit fixed-point coefficients (INTER_RESIZE_COEF_BITS),
// rows kept as int, vertical pass uchar((((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2).
void cvref_resize_linear_u8(const uint8_t* src, int H, int W, int C, uint8_t* dst, int Hd, int Wd) {
    const LinAxis tx = linear_axis(W, Wd), ty = linear_axis(H, Hd);
    std::vector<int> rows((size_t)H * Wd * C);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < Wd; ++x) {
            const int s0 = tx.ofs[x];
            const short a0 = sat_short(cv_round((1.f - tx.f[x]) * 2048.f)), a1 = sat_short(cv_round(tx.f[x] * 2048.f));
            for (int c = 0; c < C; ++c) {
                const uint8_t* S = src + ((size_t)y * W + s0) * C + c;
                rows[((size_t)y * Wd + x) * C + c] = x < tx.dmax ? S[0] * a0 + S[C] * a1 : S[0] * 2048;
            }
        }
    for (int y = 0; y < Hd; ++y) {
        const int s0 = ty.ofs[y], s1 = s0 + 1 < H ? s0 + 1 : s0;
        const short b0 = sat_short(cv_round((1.f - ty.f[y]) * 2048.f)), b1 = sat_short(cv_round(ty.f[y] * 2048.f));
        const int* R0 = rows.data() + (size_t)s0 * Wd * C;
        const int* R1 = rows.data() + (size_t)s1 * Wd * C;
        for (size_t i = 0; i < (size_t)Wd * C; ++i) {
            const int v = (((b0 * (R0[i] >> 4)) >> 16) + ((b1 * (R1[i] >> 4)) >> 16) + 2) >> 2;
            dst[(size_t)y * Wd * C + i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

// cv::resize(float HWC, INTER_CUBIC): 4 taps per axis, a = -0.75, source index clamped at the borders.
void cvref_resize_cubic_f32(const float* src, int H, int W, int C, float* dst, int Hd, int Wd) {
    const double sx = 1.0 / ((double)Wd / W), sy = 1.0 / ((double)Hd / H);
    std::vector<float> rows((size_t)H * Wd * C);
    for (int x = 0; x < Wd; ++x) {
        float f = (float)((x + 0.5) * sx - 0.5);
        const int s = cv_floor(f);
        f -= s;
        float a[4];
        cubic_coeffs(f, a);
        int idx[4];
        for (int k = 0; k < 4; ++k) { int i = s - 1 + k; idx[k] = i < 0 ? 0 : (i >= W ? W - 1 : i); }
        for (int y = 0; y < H; ++y)
            for (int c = 0; c < C; ++c) {
                const float* S = src + (size_t)y * W * C + c;
                rows[((size_t)y * Wd + x) * C + c] = S[(size_t)idx[0] * C] * a[0] + S[(size_t)idx[1] * C] * a[1] + S[(size_t)idx[2] * C] * a[2] + S[(size_t)idx[3] * C] * a[3];
            }
    }
    for (int y = 0; y < Hd; ++y) {
        float f = (float)((y + 0.5) * sy - 0.5);
        const int s = cv_floor(f);
        f -= s;
        float b[4];
        cubic_coeffs(f, b);
        const float* R[4];
        for (int k = 0; k < 4; ++k) { int i = s - 1 + k; i = i < 0 ? 0 : (i >= H ? H - 1 : i); R[k] = rows.data() + (size_t)i * Wd * C; }
        for (size_t i = 0; i < (size_t)Wd * C; ++i) dst[(size_t)y * Wd * C + i] = R[0][i] * b[0] + R[1][i] * b[1] + R[2][i] * b[2] + R[3][i] * b[3];
    }
}

// cv::resize(float HWC, INTER_AREA), shrinking.  Integer factors: resizeAreaFast_ (block sum in groups of
// four, times 1/area).  Otherwise computeResizeAreaTab + ResizeArea_Invoker (weighted row/column sums).
void cvref_resize_area_f32(const float* src, int H, int W, int C, float* dst, int Hd, int Wd) {
    const double scale_x = (double)W / Wd, scale_y = (double)H / Hd;
    const int isx = (int)std::lrint(scale_x), isy = (int)std::lrint(scale_y);
    const bool fast = std::fabs(scale_x - isx) < DBL_EPSILON && std::fabs(scale_y - isy) < DBL_EPSILON;
    if (fast) {
        const int area = isx * isy;
        const float scale = 1.f / area;
        std::vector<size_t> ofs(area);
        for (int sy = 0, k = 0; sy < isy; ++sy)
            for (int sx = 0; sx < isx; ++sx) ofs[k++] = ((size_t)sy * W + sx) * C;
        for (int y = 0; y < Hd; ++y)
            for (int x = 0; x < Wd; ++x)
                for (int c = 0; c < C; ++c) {
                    const float* S = src + ((size_t)(y * isy) * W + (size_t)x * isx) * C + c;
                    float sum = 0;
                    int k = 0;
                    for (; k <= area - 4; k += 4) sum += S[ofs[k]] + S[ofs[k + 1]] + S[ofs[k + 2]] + S[ofs[k + 3]];
                    for (; k < area; ++k) sum += S[ofs[k]];
                    dst[((size_t)y * Wd + x) * C + c] = sum * scale;
                }
        return;
    }
    struct Tab { int si, di; float alpha; };
    auto make_tab = [](int ssize, int dsize, double scale) {
        std::vector<Tab> tab;
        for (int dx = 0; dx < dsize; ++dx) {
            const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
            const double cell = std::fmin(scale, ssize - fsx1);
            int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
            sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
            sx1 = sx1 < sx2 ? sx1 : sx2;
            if (sx1 - fsx1 > 1e-3) tab.push_back({sx1 - 1, dx, (float)((sx1 - fsx1) / cell)});
            for (int sx = sx1; sx < sx2; ++sx) tab.push_back({sx, dx, (float)(1.0 / cell)});
            if (fsx2 - sx2 > 1e-3) tab.push_back({sx2, dx, (float)(std::fmin(std::fmin(fsx2 - sx2, 1.), cell) / cell)});
        }
        return tab;
    };
    const std::vector<Tab> xtab = make_tab(W, Wd, scale_x), ytab = make_tab(H, Hd, scale_y);
    std::vector<float> buf((size_t)Wd * C), sum((size_t)Wd * C);
    int prev_dy = ytab.empty() ? -1 : ytab[0].di;
    std::fill(sum.begin(), sum.end(), 0.f);
    for (size_t j = 0; j < ytab.size(); ++j) {
        const float beta = ytab[j].alpha;
        const int dy = ytab[j].di, sy = ytab[j].si;
        std::fill(buf.begin(), buf.end(), 0.f);
        const float* S = src + (size_t)sy * W * C;
        for (const Tab& t : xtab)
            for (int c = 0; c < C; ++c) buf[(size_t)t.di * C + c] += S[(size_t)t.si * C + c] * t.alpha;
        if (dy != prev_dy) {
            for (size_t i = 0; i < sum.size(); ++i) { dst[(size_t)prev_dy * Wd * C + i] = sum[i]; sum[i] = beta * buf[i]; }
            prev_dy = dy;
        } else {
            for (size_t i = 0; i < sum.size(); ++i) sum[i] += beta * buf[i];
        }
    }
    if (prev_dy >= 0)
        for (size_t i = 0; i < sum.size(); ++i) dst[(size_t)prev_dy * Wd * C + i] = sum[i];
}

// cv::resize(..., INTER_NEAREST) (resizeNN): sx = min(cvFloor(dx * ifx), W - 1), ifx = 1 / ((double)Wd / W).
void cvref_resize_nearest_f32(const float* src, int H, int W, int C, float* dst, int Hd, int Wd) {
    const double ifx = 1.0 / ((double)Wd / W), ify = 1.0 / ((double)Hd / H);
    for (int y = 0; y < Hd; ++y) {
        int sy = (int)std::floor(y * ify);
        sy = sy < H - 1 ? sy : H - 1;
        for (int x = 0; x < Wd; ++x) {
            int sx = (int)std::floor(x * ifx);
            sx = sx < W - 1 ? sx : W - 1;
            for (int c = 0; c < C; ++c) dst[((size_t)y * Wd + x) * C + c] = src[((size_t)sy * W + sx) * C + c];
        }
    }
}

// cv::remap(float HWC, mapx, mapy (CV_32FC1), INTER_LINEAR, BORDER_CONSTANT, borderValue): coordinates are
// quantised to 1/32 px (cvRound(v*32)), weights come from the float table (1-fx)(1-fy) ... of that grid.
void cvref_remap_linear_f32(const float* src, int H, int W, int C, const float* mapx, const float* mapy, float* dst, int Hd, int Wd,
                            float border) {
    for (int y = 0; y < Hd; ++y)
        for (int x = 0; x < Wd; ++x) {
            const int fx = cv_round(mapx[(size_t)y * Wd + x] * 32.f), fy = cv_round(mapy[(size_t)y * Wd + x] * 32.f);
            const int sx = sat_short(fx >> 5), sy = sat_short(fy >> 5);
            const float tx = (fx & 31) * (1.f / 32), ty = (fy & 31) * (1.f / 32);
            const float w[4] = {(1.f - ty) * (1.f - tx), (1.f - ty) * tx, ty * (1.f - tx), ty * tx};
            float* D = dst + ((size_t)y * Wd + x) * C;
            if ((unsigned)sx < (unsigned)(W - 1) && (unsigned)sy < (unsigned)(H - 1)) {
                const float* S = src + ((size_t)sy * W + sx) * C;
                for (int c = 0; c < C; ++c) D[c] = S[c] * w[0] + S[C + c] * w[1] + S[(size_t)W * C + c] * w[2] + S[(size_t)W * C + C + c] * w[3];
            } else if (sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0) {
                for (int c = 0; c < C; ++c) D[c] = border;
            } else {
                for (int c = 0; c < C; ++c) {
                    auto at = [&](int yy, int xx) { return ((unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H) ? src[((size_t)yy * W + xx) * C + c] : border; };
                    D[c] = at(sy, sx) * w[0] + at(sy, sx + 1) * w[1] + at(sy + 1, sx) * w[2] + at(sy + 1, sx + 1) * w[3];
                }
            }
        }
}

// cv::Sobel(float HW, CV_32F, dx, dy, ksize=3, BORDER_REFLECT_101) for (dx,dy) in {(1,0),(0,1)}:
// row filter then column filter with the small-kernel forms  d = S[+1] - S[-1],  s = S[-1] + S[0]*2 + S[+1].
void cvref_sobel3_f32(const float* src, int H, int W, int dx, float* dst) {
    std::vector<float> r((size_t)H * W);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float a = src[(size_t)y * W + reflect101(x - 1, W)], b = src[(size_t)y * W + x], c = src[(size_t)y * W + reflect101(x + 1, W)];
            r[(size_t)y * W + x] = dx ? c - a : a + b * 2 + c;
        }
    for (int y = 0; y < H; ++y) {
        const float* R0 = r.data() + (size_t)reflect101(y - 1, H) * W;
        const float* R1 = r.data() + (size_t)y * W;
        const float* R2 = r.data() + (size_t)reflect101(y + 1, H) * W;
        for (int x = 0; x < W; ++x) dst[(size_t)y * W + x] = dx ? R0[x] + R1[x] * 2 + R2[x] : R2[x] - R0[x];
    }
}

}  // extern "C"
