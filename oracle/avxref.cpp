// oracle/avxref.cpp -- TEST INFRASTRUCTURE ONLY (the checker, never the product).
//
// CPU restatement of the arithmetic on the animal-vision per-frame hot path for
// the stages where the reference leans on compiled third-party code whose
// operation order NumPy cannot express (there is no fma in NumPy):
//
//   * cv2.GaussianBlur  (reference: animals/animal_utils.py:144,168,171;
//                        uv_helpers.py:73) -- OpenCV is NOT under /root/reference
//     and is not installed anywhere in this pipeline, so what follows restates
//     OpenCV 4.x's published algorithm (imgproc/smooth.dispatch.cpp
//     createGaussianKernels/getGaussianKernel, filter.simd.hpp RowVec_32f /
//     SymmColumnVec_32f): PARITY UNPINNED for this stage.
//   * `pixels @ T.T`    (reference: animals/dog.py:47 and siblings,
//                        animals/animal_utils.py:64,77) -- OpenBLAS s/dgemm with
//     K=3; on the build host it is bit-identical to the FMA chain below
//     (verified by tests/test_oracle_golden.py against the imported reference).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// this library.  Build: `make -C oracle` (g++, -ffp-contract=off so that every
// fused multiply-add below is an explicit fma()/fmaf() and nothing else fuses).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

// cv::borderInterpolate(p, len, BORDER_REFLECT_101)
inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

inline float fma_t(float a, float b, float c) { return std::fmaf(a, b, c); }
inline double fma_t(double a, double b, double c) { return std::fma(a, b, c); }

// sepFilter2D restated: horizontal (row) pass first into a full intermediate,
// then vertical (column) pass.  Row pass: s = x[0]*k[0]; s = fma(x[j], k[j], s)
// left to right (RowVec_32f).  Column pass on a symmetric kernel: s = c*k[r];
// s = fma(x[+j] + x[-j], k[r+j], s) for j = 1..r (SymmColumnVec_32f, delta = 0).
template <typename T>
void sepfilter(const T* src, T* dst, int H, int W, int C, const T* kx, int nx, const T* ky, int ny) {
    const int rx = nx / 2, ry = ny / 2;
    std::vector<T> mid((size_t)H * W * C);
    std::vector<int> xi((size_t)W + 2 * rx);
    for (int x = -rx; x < W + rx; ++x) xi[x + rx] = reflect101(x, W);
    for (int y = 0; y < H; ++y) {
        const T* row = src + (size_t)y * W * C;
        T* out = mid.data() + (size_t)y * W * C;
        for (int x = 0; x < W; ++x) {
            for (int c = 0; c < C; ++c) {
                T s = row[(size_t)xi[x] * C + c] * kx[0];
                for (int j = 1; j < nx; ++j) s = fma_t(row[(size_t)xi[x + j] * C + c], kx[j], s);
                out[(size_t)x * C + c] = s;
            }
        }
    }
    const size_t stride = (size_t)W * C;
    for (int y = 0; y < H; ++y) {
        T* out = dst + (size_t)y * stride;
        const T* c0 = mid.data() + (size_t)y * stride;
        for (size_t i = 0; i < stride; ++i) out[i] = c0[i] * ky[ry];
        for (int j = 1; j <= ry; ++j) {
            const T* a = mid.data() + (size_t)reflect101(y + j, H) * stride;
            const T* b = mid.data() + (size_t)reflect101(y - j, H) * stride;
            const T k = ky[ry + j];
            for (size_t i = 0; i < stride; ++i) out[i] = fma_t(a[i] + b[i], k, out[i]);
        }
    }
}

}  // namespace

extern "C" {

// cv::getGaussianKernel(n, sigma, CV_64F) for sigma > 0 (smooth.dispatch.cpp):
// t_i = exp(-0.5/sigma^2 * (i-(n-1)/2)^2), normalised by 1/sum, in IEEE double.
void avxref_gaussian_kernel(int n, double sigma, double* out) {
    const double scale2x = -0.5 / (sigma * sigma);
    const double c = (n - 1) * 0.5;
    double sum = 0.0;
    for (int i = 0; i < n; ++i) {
        const double x = i - c;
        out[i] = std::exp(scale2x * x * x);
        sum += out[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < n; ++i) out[i] *= sum;
}

void avxref_sepfilter_f32(const float* src, float* dst, int H, int W, int C, const float* kx, int nx,
                          const float* ky, int ny) {
    sepfilter<float>(src, dst, H, W, C, kx, nx, ky, ny);
}

void avxref_sepfilter_f64(const double* src, double* dst, int H, int W, int C, const double* kx, int nx,
                          const double* ky, int ny) {
    sepfilter<double>(src, dst, H, W, C, kx, nx, ky, ny);
}

// out[n][i] = fma(in[n][2], M[i][2], fma(in[n][1], M[i][1], in[n][0]*M[i][0]))
// == NumPy `in @ M.T` for (N,3)x(3,3) on the build host (sgemm, K = 3).
void avxref_matmul3_f32(const float* in, float* out, size_t n, const float* M) {
    for (size_t p = 0; p < n; ++p) {
        const float a = in[3 * p], b = in[3 * p + 1], c = in[3 * p + 2];
        for (int i = 0; i < 3; ++i)
            out[3 * p + i] = std::fmaf(c, M[3 * i + 2], std::fmaf(b, M[3 * i + 1], a * M[3 * i]));
    }
}

// f32 pixels promoted to f64, times an f64 matrix (reference: LMS_to_RGB,
// animals/animal_utils.py:70-77, whose matrix literal has no dtype => float64).
void avxref_matmul3_f32_f64(const float* in, double* out, size_t n, const double* M) {
    for (size_t p = 0; p < n; ++p) {
        const double a = in[3 * p], b = in[3 * p + 1], c = in[3 * p + 2];
        for (int i = 0; i < 3; ++i)
            out[3 * p + i] = std::fma(c, M[3 * i + 2], std::fma(b, M[3 * i + 1], a * M[3 * i]));
    }
}

// Monotone step-function quantiser: out = #{k : thr[k] <= x}, thr ascending,
// 255 entries.  Restates clip(oetf(clip(x,0,1)),0,1)*255+0.5 -> uint8 truncation
// (reference: animals/dog.py:54-57) through its exact threshold table; the
// table itself is reference output (tests/golden/srgb_tables.npz).
void avxref_quantize_f32(const float* x, uint8_t* out, size_t n, const float* thr) {
    for (size_t i = 0; i < n; ++i) {
        int lo = 0, hi = 255;  // count of thr <= x, in [0,255]
        const float v = x[i];
        while (lo < hi) {
            const int m = (lo + hi) / 2;
            if (thr[m] <= v) lo = m + 1; else hi = m;
        }
        out[i] = (uint8_t)lo;
    }
}

void avxref_quantize_f64(const double* x, uint8_t* out, size_t n, const double* thr) {
    for (size_t i = 0; i < n; ++i) {
        int lo = 0, hi = 255;
        const double v = x[i];
        while (lo < hi) {
            const int m = (lo + hi) / 2;
            if (thr[m] <= v) lo = m + 1; else hi = m;
        }
        out[i] = (uint8_t)lo;
    }
}

}  // extern "C"
