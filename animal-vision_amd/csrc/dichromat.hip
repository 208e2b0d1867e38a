// csrc/dichromat.hip -- fused dichromat path for gfx950 (MI355X), one launch per batch of frames.
//
// Replaces, per frame (reference: animals/dog.py:14-61 template, animals/cat.py:95-112 colour core):
//   uint8 HWC load -> [a3+a4] 256-entry decode table -> [a5-a7] 3x3 colour collapse (FMA chain)
//   -> [a8] separable Gaussian with OpenCV GaussianBlur semantics (row pass, then column pass,
//      BORDER_REFLECT_101)  |  [a10] per-row channel-2 gain  -> optional chroma compression
//   -> [a11] clip / OETF / clip / *255+0.5 truncation as a 255-entry threshold count -> uint8 HWC store.
//
// Arithmetic contract (bit-exact with oracle/avxref.cpp; compiled with -ffp-contract=off so only the
// explicit fma() calls fuse):
//   colour : o_i = fma(c2, M[i][2], fma(c1, M[i][1], c0*M[i][0]))
//   row    : s = x[0]*k[0]; s = fma(x[j], k[j], s), j = 1..n-1 (left to right)
//   column : s = c*k[r];    s = fma(x[+j] + x[-j], k[r+j], s), j = 1..r
//   encode : out = #{k : thr[k] <= clip(x,0,1)}
//
// Data layout: frames are HWC uint8 in HBM (3 B/px read, 3 B/px written: 6 B/px algorithmic).  A
// workgroup owns a TW x TH output tile; the haloed tile lives in LDS as three channel planes of the
// compute type (float, or double for the cat tail), so the blur never touches HBM.
#include "avx_internal.h"

namespace {

constexpr int kThreads = 256;

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
};

__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

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

// ---- variant 0: simple reference kernel (one output sample per thread per step) -----------------
// Kept as the in-library cross-check for the tuned variants: same arithmetic contract, no tricks.
template <typename T, int COLOR, bool DARK>
__global__ __launch_bounds__(kThreads) void dichromat_simple_kernel(DichromatArgs a, Taps<T> taps) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int r = a.r, TW = a.TW, TH = a.TH;
    const int AW = TW + 2 * r, AH = TH + 2 * r;
    T* thr = reinterpret_cast<T*>(smem_raw);                 // 256
    float* lut = reinterpret_cast<float*>(thr + 256);        // 256
    T* A = reinterpret_cast<T*>(lut + 256);                  // 3 * AH * AW
    T* Bm = A + 3 * AH * AW;                                 // 3 * AH * TW (only when r > 0)
    const int tid = threadIdx.x;
    for (int i = tid; i < 256; i += kThreads) {
        thr[i] = reinterpret_cast<const T*>(a.enc_thr)[i];
        lut[i] = a.decode_lut[i];
    }
    __syncthreads();
    const int tiles_per_frame = a.tiles_x * a.tiles_y;
    const int total = tiles_per_frame * a.n_frames;
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int f = tile / tiles_per_frame;
        const int t2 = tile - f * tiles_per_frame;
        const int ty = t2 / a.tiles_x, tx = t2 - ty * a.tiles_x;
        if (DARK && a.flags[f] != 0u) continue;  // frame was not "all <= 1": main pass already right
        const int x0 = tx * TW, y0 = ty * TH;
        const uint8_t* fin = a.in + (size_t)f * a.H * a.W * 3;
        uint8_t* fout = a.out + (size_t)f * a.H * a.W * 3;
        uint32_t seen_gt1 = 0;
        // phase 1: load + decode + colour stage -> A planes
        for (int i = tid; i < AH * AW; i += kThreads) {
            const int ly = i / AW, lx = i - ly * AW;
            const int gy = reflect101(y0 - r + ly, a.H), gx = reflect101(x0 - r + lx, a.W);
            const uint8_t* p = fin + ((size_t)gy * a.W + gx) * 3;
            const uint32_t b0 = p[0], b1 = p[1], b2 = p[2];
            seen_gt1 |= (b0 | b1 | b2) >> 1;
            float c0, c1, c2;
            if (DARK) {  // get_normalized_image skips /255 when max <= 1: codes 0/1 are 0.0/1.0
                c0 = b0 ? 1.0f : 0.0f; c1 = b1 ? 1.0f : 0.0f; c2 = b2 ? 1.0f : 0.0f;
            } else {
                c0 = lut[b0]; c1 = lut[b1]; c2 = lut[b2];
            }
            T o0, o1, o2;
            if (COLOR == AVX_COLOR_MATRIX) {
                o0 = fma_t(c2, a.M[2], fma_t(c1, a.M[1], c0 * a.M[0]));
                o1 = fma_t(c2, a.M[5], fma_t(c1, a.M[4], c0 * a.M[3]));
                o2 = fma_t(c2, a.M[8], fma_t(c1, a.M[7], c0 * a.M[6]));
            } else {
                const float l = fma_t(c2, a.M[2], fma_t(c1, a.M[1], c0 * a.M[0]));
                const float m = fma_t(c2, a.M[5], fma_t(c1, a.M[4], c0 * a.M[3]));
                const float s = fma_t(c2, a.M[8], fma_t(c1, a.M[7], c0 * a.M[6]));
                const float lm = a.alpha * l + a.one_minus_alpha * m;  // separate mul, mul, add (cat.py:99)
                const double dlm = (double)lm, ds = (double)s;
                o0 = (T)__builtin_fma(ds, a.Bk[2], __builtin_fma(dlm, a.Bk[1], dlm * a.Bk[0]));
                o1 = (T)__builtin_fma(ds, a.Bk[5], __builtin_fma(dlm, a.Bk[4], dlm * a.Bk[3]));
                o2 = (T)__builtin_fma(ds, a.Bk[8], __builtin_fma(dlm, a.Bk[7], dlm * a.Bk[6]));
            }
            A[i] = o0;
            A[AH * AW + i] = o1;
            A[2 * AH * AW + i] = o2;
        }
        if (!DARK && seen_gt1) a.flags[f] = 1u;  // benign race: every writer stores the same value
        __syncthreads();
        const T* src = A;  // planes the epilogue reads, with row stride sw and origin offset
        int sw = AW, oy = 0;
        if (r > 0) {
            const int n = 2 * r + 1;
            // phase 2: row pass over all AH rows -> Bm[c][ly][x]
            for (int i = tid; i < 3 * AH * TW; i += kThreads) {
                const int c = i / (AH * TW);
                const int rem = i - c * AH * TW;
                const int ly = rem / TW, x = rem - ly * TW;
                const T* row = A + (c * AH + ly) * AW + x;
                T s = row[0] * taps.k[0];
                for (int j = 1; j < n; ++j) s = fma_t(row[j], taps.k[j], s);
                Bm[i] = s;
            }
            __syncthreads();
            src = Bm;
            sw = TW;
            oy = r;
        }
        // phase 3: column pass (if any) + post stage + encode + store, one pixel per thread step
        for (int i = tid; i < TH * TW; i += kThreads) {
            const int y = i / TW, x = i - y * TW;
            const int gy = y0 + y, gx = x0 + x;
            if (gy >= a.H || gx >= a.W) continue;
            T v[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const T* col = src + (c * AH + y + oy) * sw + x;
                if (r > 0) {
                    T s = col[0] * taps.k[r];
                    for (int j = 1; j <= r; ++j) s = fma_t(col[j * sw] + col[-j * sw], taps.k[r + j], s);
                    v[c] = s;
                } else {
                    v[c] = col[0];
                }
            }
            if (a.post_mode == AVX_POST_ROWGAIN) {  // apply_s_cone_vertical_gain, animal_utils.py:252-257
                T b = v[2] * (T)a.row_gain[gy];
                if (a.row_gain_clamp) b = b < (T)0 ? (T)0 : (b > (T)1 ? (T)1 : b);
                v[2] = b;
            }
            if (a.chroma_enable) {  // apply_chroma_compression, animal_utils.py:180-181
                const T gray = ((v[0] + v[1]) + v[2]) / (T)3;
                const T keep = (T)a.chroma_keep;
                v[0] = gray + (v[0] - gray) * keep;
                v[1] = gray + (v[1] - gray) * keep;
                v[2] = gray + (v[2] - gray) * keep;
            }
            uint8_t* q = fout + ((size_t)gy * a.W + gx) * 3;
            q[0] = (uint8_t)quantize<T>(v[0], thr);
            q[1] = (uint8_t)quantize<T>(v[1], thr);
            q[2] = (uint8_t)quantize<T>(v[2], thr);
        }
        __syncthreads();  // A/Bm are reused by the next tile
    }
}

template <typename T>
size_t simple_lds_bytes(int TW, int TH, int r) {
    const size_t AW = TW + 2 * r, AH = TH + 2 * r;
    return 256 * sizeof(T) + 256 * sizeof(float) + 3 * AH * AW * sizeof(T) + (r > 0 ? 3 * AH * (size_t)TW * sizeof(T) : 0);
}

template <typename T, int COLOR>
int launch_simple(avx_ctx* ctx, DichromatArgs& a, const avx_dichromat_desc* d, hipStream_t s) {
    // tile: widest that fits the 160 KiB LDS of a CU
    int TW = 64, TH = 32;
    while (simple_lds_bytes<T>(TW, TH, a.r) > 150 * 1024 && TH > 8) TH /= 2;
    while (simple_lds_bytes<T>(TW, TH, a.r) > 150 * 1024 && TW > 16) TW /= 2;
    AVX_REQUIRE(ctx, simple_lds_bytes<T>(TW, TH, a.r) <= 150 * 1024, "avx_dichromat_u8: ksize %d too large for LDS", d->ksize);
    a.TW = TW;
    a.TH = TH;
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    Taps<T> taps;
    for (int i = 0; i < AVX_MAX_KSIZE; ++i) taps.k[i] = (T)0;
    if (a.r > 0)
        for (int i = 0; i < d->ksize; ++i) taps.k[i] = (T)d->taps_host[i];  // getGaussianKernel: double -> ktype
    const size_t lds = simple_lds_bytes<T>(TW, TH, a.r);
    const long total = (long)a.tiles_x * a.tiles_y * a.n_frames;
    const int grid = (int)(total < 8L * ctx->num_cus ? total : 8L * ctx->num_cus);
    auto kmain = dichromat_simple_kernel<T, COLOR, false>;
    auto kdark = dichromat_simple_kernel<T, COLOR, true>;
    AVX_HIP(ctx, hipFuncSetAttribute((const void*)kmain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    AVX_HIP(ctx, hipFuncSetAttribute((const void*)kdark, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    AVX_HIP(ctx, hipMemsetAsync(a.flags, 0, sizeof(uint32_t) * a.n_frames, s));
    hipLaunchKernelGGL(kmain, dim3(grid), dim3(kThreads), lds, s, a, taps);
    AVX_HIP(ctx, hipGetLastError());
    // Fix-up for frames whose every byte is <= 1 (get_normalized_image does not divide those by 255):
    // exits at once for every other frame.
    const int gdark = (int)(total < ctx->num_cus ? total : ctx->num_cus);
    hipLaunchKernelGGL(kdark, dim3(gdark), dim3(kThreads), lds, s, a, taps);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

}  // namespace

extern "C" int avx_dichromat_u8(avx_ctx* ctx, const uint8_t* in_hwc, uint8_t* out_hwc, int n_frames, int H, int W,
                                const avx_dichromat_desc* d, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, d != nullptr && d->struct_size == sizeof(avx_dichromat_desc),
                "avx_dichromat_u8: desc is NULL or struct_size mismatch (ABI %d)", AVX_ABI_VERSION);
    AVX_REQUIRE(ctx, in_hwc && out_hwc, "avx_dichromat_u8: NULL frame pointer");
    AVX_REQUIRE(ctx, n_frames >= 0 && H > 0 && W > 0, "avx_dichromat_u8: bad shape n=%d H=%d W=%d", n_frames, H, W);
    AVX_REQUIRE(ctx, (size_t)n_frames * H * W * 3 < ((size_t)1 << 40), "avx_dichromat_u8: batch too large");
    AVX_REQUIRE(ctx, d->color_mode == AVX_COLOR_MATRIX || d->color_mode == AVX_COLOR_CAT_MERGE,
                "avx_dichromat_u8: unknown color_mode %d", d->color_mode);
    AVX_REQUIRE(ctx, d->post_mode >= AVX_POST_NONE && d->post_mode <= AVX_POST_ROWGAIN,
                "avx_dichromat_u8: unknown post_mode %d", d->post_mode);
    if (n_frames == 0) return AVX_OK;
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    DichromatArgs a{};
    a.in = in_hwc;
    a.out = out_hwc;
    a.n_frames = n_frames;
    a.H = H;
    a.W = W;
    a.r = 0;
    a.post_mode = d->post_mode;
    if (d->post_mode == AVX_POST_GAUSS) {
        AVX_REQUIRE(ctx, d->ksize >= 1 && d->ksize <= AVX_MAX_KSIZE && (d->ksize & 1), "avx_dichromat_u8: ksize %d must be odd, 1..%d", d->ksize, AVX_MAX_KSIZE);
        AVX_REQUIRE(ctx, d->taps_host != nullptr, "avx_dichromat_u8: taps_host is NULL");
        a.r = d->ksize / 2;
    }
    if (d->post_mode == AVX_POST_ROWGAIN) {
        AVX_REQUIRE(ctx, d->row_gain_host != nullptr, "avx_dichromat_u8: row_gain_host is NULL");
        if ((size_t)H > ctx->row_gain_cap) {
            if (ctx->d_row_gain) AVX_HIP(ctx, hipFree(ctx->d_row_gain));
            ctx->d_row_gain = nullptr;
            ctx->row_gain_cap = 0;
            AVX_HIP(ctx, hipMalloc((void**)&ctx->d_row_gain, sizeof(float) * H));
            ctx->row_gain_cap = H;
        }
        AVX_HIP(ctx, hipMemcpyAsync(ctx->d_row_gain, d->row_gain_host, sizeof(float) * H, hipMemcpyHostToDevice, s));
        a.row_gain = ctx->d_row_gain;
        a.row_gain_clamp = d->row_gain_clamp;
    }
    if ((size_t)n_frames > ctx->flags_cap) {
        if (ctx->d_flags) AVX_HIP(ctx, hipFree(ctx->d_flags));
        ctx->d_flags = nullptr;
        ctx->flags_cap = 0;
        size_t cap = (size_t)n_frames < 64 ? 64 : (size_t)n_frames;
        AVX_HIP(ctx, hipMalloc((void**)&ctx->d_flags, sizeof(uint32_t) * cap));
        ctx->flags_cap = cap;
    }
    a.flags = ctx->d_flags;
    a.chroma_enable = d->chroma_enable;
    a.chroma_keep = d->chroma_keep;
    a.decode_lut = ctx->d_decode_lut;
    if (d->color_mode == AVX_COLOR_MATRIX) {
        for (int i = 0; i < 9; ++i) a.M[i] = d->matrix[i];
        a.enc_thr = ctx->d_enc_thr_f32;
        return launch_simple<float, AVX_COLOR_MATRIX>(ctx, a, d, s);
    }
    // cat: RGB->LMS float32 (animal_utils.py:56-63), LMS->RGB float64 (animal_utils.py:70-76)
    static const float kRgbToLms[9] = {0.31399022f, 0.63951294f, 0.04649755f, 0.15537241f, 0.75789446f,
                                       0.08670142f, 0.01775239f, 0.10944209f, 0.87256922f};
    static const double kLmsToRgb[9] = {5.472213, -4.6419606, 0.16963711, -1.125242, 2.2931712,
                                        -0.16789523, 0.02980164, -0.19318072, 1.1636479};
    for (int i = 0; i < 9; ++i) {
        a.M[i] = kRgbToLms[i];
        a.Bk[i] = kLmsToRgb[i];
    }
    a.alpha = d->cat_alpha;
    a.one_minus_alpha = d->cat_beta;
    a.enc_thr = ctx->d_enc_thr_f64;
    return launch_simple<double, AVX_COLOR_CAT_MERGE>(ctx, a, d, s);
}
