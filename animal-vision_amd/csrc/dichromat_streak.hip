// csrc/dichromat_streak.hip -- AVX_POST_STREAK: apply_anisotropic_acuity_blur_with_streak
// (animals/animal_utils.py:147-172) fused with the rest of the dichromat template, reproduced AS CODED:
//
//   quirk Q3: the reference hands each image row `out[y]` of shape (W,3) to cv2.GaussianBlur, which sees a
//   W-row x 3-column single-channel image.  So, per image row y with sx = max(0.4, sigma_map/2), sy = sigma_map:
//     pass 1  GaussianBlur(row, (0,0), sigmaX=sx, sigmaY=0 -> sx): k1 = cvRound(8 sx + 1)|1 taps in both
//             directions: "horizontal" = ACROSS R,G,B (reflect-101 on a width of 3), then "vertical" = ALONG
//             the image row (reflect-101 on W);
//     pass 2  GaussianBlur(tmp, (0,0), sigmaX=1e-16 -> 1 tap (x*1.0, exact), sigmaY=sy): k2 taps ALONG the
//             image row again.
//   There is no blur across image rows at all, so rows are independent: one workgroup per image row, the
//   row (W x 3 floats) ping-pongs between two LDS buffers.
//   quirk Q4 (in-place mutation relied on by pig.py:35) needs nothing here: the blurred values are what
//   is encoded either way.
//
// Same arithmetic contract as the other dichromat kernels (row pass: sequential fma left to right; column
// pass: centre tap then fma(x[+j] + x[-j], k[r+j], s)); bit-exact with oracle/cpu_ref.py.
#include "dichromat_common.h"

using namespace avxk;

namespace {

constexpr int kST = 256;

struct StreakArgs {
    const float* rows;  // device: H x stride floats: [0]=k1, [1]=k2, [2..2+13) taps1, [15..15+33) taps2
    int stride;
};

template <bool DARK>
__global__ __launch_bounds__(kST) void dichromat_streak_kernel(DichromatArgs a, StreakArgs st) {
    extern __shared__ __align__(16) float smem_f[];
    float* buf0 = smem_f;                 // W*3
    float* buf1 = smem_f + (size_t)a.W * 3;
    __shared__ float lut[256];
    __shared__ float thr[256];
    __shared__ float t1[16], t2[40];
    const int tid = threadIdx.x;
    for (int i = tid; i < 256; i += kST) { lut[i] = a.decode_lut[i]; thr[i] = reinterpret_cast<const float*>(a.enc_thr)[i]; }
    const int W = a.W, n3 = a.W * 3;
    const size_t frame_bytes = (size_t)a.H * a.W * 3;
    const int total_rows = a.n_frames * a.H;
    for (int gr = blockIdx.x; gr < total_rows; gr += gridDim.x) {
        const int f = gr / a.H, y = gr - f * a.H;
        __syncthreads();  // tables of the previous row are no longer read
        if (DARK && a.flags[f] != 0u) continue;
        const float* rt = st.rows + (size_t)y * st.stride;
        const int k1 = (int)rt[0], k2 = (int)rt[1];
        const int r1 = k1 / 2, r2 = k2 / 2;
        if (tid < 13) t1[tid] = rt[2 + tid];
        if (tid < 33) t2[tid] = rt[15 + tid];
        const uint8_t* src = a.in + frame_bytes * f + (size_t)y * n3;
        uint8_t* dst = a.out + frame_bytes * f + (size_t)y * n3;
        // ---- decode + colour stage -> buf0[x*3 + c] ------------------------------------------------
        uint32_t seen = 0;
        for (int x = tid; x < W; x += kST) {
            const uint32_t b0 = src[3 * x], b1 = src[3 * x + 1], b2 = src[3 * x + 2];
            seen |= (b0 | b1 | b2) >> 1;
            float c0, c1, c2;
            if (DARK) { c0 = b0 ? 1.0f : 0.0f; c1 = b1 ? 1.0f : 0.0f; c2 = b2 ? 1.0f : 0.0f; }
            else { c0 = lut[b0]; c1 = lut[b1]; c2 = lut[b2]; }
            buf0[3 * x + 0] = fma_t(c2, a.M[2], fma_t(c1, a.M[1], c0 * a.M[0]));
            buf0[3 * x + 1] = fma_t(c2, a.M[5], fma_t(c1, a.M[4], c0 * a.M[3]));
            buf0[3 * x + 2] = fma_t(c2, a.M[8], fma_t(c1, a.M[7], c0 * a.M[6]));
        }
        if (!DARK && seen) a.flags[f] = 1u;
        __syncthreads();
        // ---- pass 1, "row" direction = across the 3 channels (width 3, reflect-101) -> buf1 ----------
        for (int i = tid; i < n3; i += kST) {
            const int x = i / 3, c = i - 3 * x;
            const float* px = buf0 + 3 * x;
            float s = px[reflect101(c - r1, 3)] * t1[0];
            for (int j = 1; j < k1; ++j) s = fma_t(px[reflect101(c - r1 + j, 3)], t1[j], s);
            buf1[i] = s;
        }
        __syncthreads();
        // ---- pass 1, "column" direction = along the image row (reflect-101 on W) -> buf0 -------------
        for (int i = tid; i < n3; i += kST) {
            const int x = i / 3, c = i - 3 * x;
            float s = buf1[i] * t1[r1];
            for (int j = 1; j <= r1; ++j)
                s = fma_t(buf1[3 * reflect101(x + j, W) + c] + buf1[3 * reflect101(x - j, W) + c], t1[r1 + j], s);
            buf0[i] = s;
        }
        __syncthreads();
        // ---- pass 2: 1-tap row direction is x*1.0 (exact); column direction along the image row, sy ----
        for (int x = tid; x < W; x += kST) {
            float v[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float s = buf0[3 * x + c] * t2[r2];
                for (int j = 1; j <= r2; ++j)
                    s = fma_t(buf0[3 * reflect101(x + j, W) + c] + buf0[3 * reflect101(x - j, W) + c], t2[r2 + j], s);
                v[c] = s;
            }
            if (a.chroma_enable) {  // apply_chroma_compression, animal_utils.py:180-181
                const float gray = ((v[0] + v[1]) + v[2]) / 3.0f;
                v[0] = gray + (v[0] - gray) * a.chroma_keep;
                v[1] = gray + (v[1] - gray) * a.chroma_keep;
                v[2] = gray + (v[2] - gray) * a.chroma_keep;
            }
            dst[3 * x + 0] = (uint8_t)quantize<float>(v[0], thr);
            dst[3 * x + 1] = (uint8_t)quantize<float>(v[1], thr);
            dst[3 * x + 2] = (uint8_t)quantize<float>(v[2], thr);
        }
    }
}

}  // namespace

int avx_launch_dichromat_streak(avx_ctx* ctx, avx_ws* ws, DichromatArgs& a, const avx_dichromat_desc* d, hipStream_t s) {
    AVX_REQUIRE(ctx, d->streak_rows_host != nullptr && d->streak_stride >= 48, "avx_dichromat_u8: streak tables missing (stride >= 48)");
    const size_t lds = sizeof(float) * (size_t)a.W * 3 * 2;
    AVX_REQUIRE(ctx, lds <= 150 * 1024, "avx_dichromat_u8: frame width %d too large for the streak kernel (row must fit LDS)", a.W);
    const size_t tbytes = sizeof(float) * (size_t)a.H * d->streak_stride;
    if (tbytes > ws->row_gain_cap * sizeof(float)) {  // the per-row table shares the row-gain buffer
        if (ws->d_row_gain) { AVX_HIP(ctx, hipStreamSynchronize(s)); AVX_HIP(ctx, hipFree(ws->d_row_gain)); }
        ws->d_row_gain = nullptr;
        ws->row_gain_cap = 0;
        AVX_HIP(ctx, hipMalloc((void**)&ws->d_row_gain, tbytes));
        ws->row_gain_cap = tbytes / sizeof(float);
    }
    AVX_HIP(ctx, hipMemcpyAsync(ws->d_row_gain, d->streak_rows_host, tbytes, hipMemcpyHostToDevice, s));
    StreakArgs st{ws->d_row_gain, d->streak_stride};
    auto kmain = dichromat_streak_kernel<false>;
    auto kdark = dichromat_streak_kernel<true>;
    AVX_HIP(ctx, hipFuncSetAttribute((const void*)kmain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    AVX_HIP(ctx, hipFuncSetAttribute((const void*)kdark, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long rows = (long)a.n_frames * a.H;
    const long cap = (long)ctx->num_cus * 4;
    const int grid = (int)(rows < cap ? rows : cap);
    AVX_HIP(ctx, hipMemsetAsync(a.flags, 0, sizeof(uint32_t) * a.n_frames, s));
    hipLaunchKernelGGL(kmain, dim3(grid), dim3(kST), lds, s, a, st);
    AVX_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(kdark, dim3(grid), dim3(kST), lds, s, a, st);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}
