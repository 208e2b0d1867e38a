// csrc/mst_mfma.hip -- fused matrix-core kernels for the two GEMM neighbourhoods of an MST++ MSAB block
// (reference: ml/MST_plus_plus/predict_code/architecture/MST_Plus_Plus.py, MS_MSA :88-139, PreNorm :57-65,
// FeedForward :141-158), float16 activations with channel groups stored 32 wide (C in {32, 64, 128}).
//
//   avx_mst_qkv_gram      to_q / to_k / to_v (:118-120) + F.normalize's column norms (:127-128) + k^T q over ALL
//                         pixels (:129) in ONE pass over x: q and k never reach memory (they are consumed from the
//                         accumulators by the Gram MFMA), only v is written.  Replaces a (N x 3C) GEMM, a second
//                         pass over its output for the Gram matrix and a strided copy of v.
//   avx_mst_ln_gemm_gelu  PreNorm's LayerNorm -> FeedForward's first 1x1 conv (C -> 4C) -> GELU: the normalised
//                         row is built in registers (K = C is the whole row) and the GELU runs on the accumulators,
//                         so the (N x 4C) hidden tensor is written once instead of written, re-read and re-written.
//
// Both are HBM-streaming kernels around v_mfma_f32_32x32x8_f16: a wave owns 32 pixels at a time; its x fragment
// (lane = pixel + 32 * half, each lane C/2 CONTIGUOUS channels of its row) is operand A for "pixels x channels"
// results and operand B for "channels x pixels" results.  The K order inside the product is free, so it is chosen
// to match that contiguous load: step s, lane half h, element j  <->  input channel h*C/2 + 4s + j; the weights are
// pre-packed on the host into exactly that fragment order ([tile][step][lane] x 4 halfs, held in LDS).
//   * "pixels x channels" (q, k): D lane (channel, half) holds pixels 8*(v/4) + 4*half + v%4 -- which IS the K-layout
//     of the next MFMA's operands, so k^T q needs only a float->half pack of the accumulators.
//   * "channels x pixels" (v, hidden): the weight columns of a tile are permuted at pack time so that D lane
//     (pixel, half) holds channels 16*half + v, v = 0..15: 32 contiguous bytes per lane, 64 per pixel and tile.
// Rounding points follow the autocast pipeline of predict_torch.py:109: GEMM results are rounded to float16 before
// anything consumes them; LayerNorm statistics, the Gram accumulation and the norms are float32.
#include <hip/hip_fp16.h>

#include "avx_internal.h"

namespace {

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));

constexpr int kT = 256;  // 4 waves, each marching over its own 32-pixel tiles

__device__ __forceinline__ float16_t mfma(half4_t a, half4_t b, float16_t c) { return __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, c, 0, 0, 0); }

__device__ __forceinline__ float gelu_erf_h(float x) {  // exact-erf GELU for a float16 result (same form as csrc/mst.hip)
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __frcp_rn(1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = 1.0f - poly * __expf(-z * z);
    return 0.5f * x * (1.0f + (x < 0.f ? -e : e));
}

// lane (p = lane & 31, h = lane >> 5) loads channels [h*C/2, (h+1)*C/2) of pixel tile*32 + p; rows past n read as zero
template <int C>
__device__ __forceinline__ void load_x(const __half* __restrict__ x, size_t tile, size_t n, int lane, half4_t (&xf)[C / 8]) {
    const int p = lane & 31, h = lane >> 5;
    const size_t row = tile * 32 + p;
    if (row < n) {
        const uint4* src = reinterpret_cast<const uint4*>(x + row * (size_t)C + h * (C / 2));
#pragma unroll
        for (int q = 0; q < C / 16; ++q) {
            const uint4 v = src[q];
            const half8_t h8 = __builtin_bit_cast(half8_t, v);
            xf[2 * q] = half4_t{h8[0], h8[1], h8[2], h8[3]};
            xf[2 * q + 1] = half4_t{h8[4], h8[5], h8[6], h8[7]};
        }
    } else {
#pragma unroll
        for (int s = 0; s < C / 8; ++s) xf[s] = half4_t{0, 0, 0, 0};
    }
}

// lane (pixel p, half h) holds channels 16h + v of a 32-channel tile: two 16-byte stores
__device__ __forceinline__ void store_tile16(__half* dst /* row base + 32*tile + 16*h */, const _Float16 (&o)[16]) {
    half8_t a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = o[i]; b[i] = o[8 + i]; }
    reinterpret_cast<uint4*>(dst)[0] = __builtin_bit_cast(uint4, a);
    reinterpret_cast<uint4*>(dst)[1] = __builtin_bit_cast(uint4, b);
}

// ---- to_q/to_k/to_v + column norms + Gram ------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(kT) void k_mst_qkv(const __half* __restrict__ x, const uint2* __restrict__ wpack /*[3*HD][KS][64]*/, size_t n,
                                                __half* __restrict__ v_out /*[n][C]*/, float* __restrict__ partial /*[blocks][HD][34][32]*/) {
    constexpr int KS = C / 8, HD = C / 32, NT = 3 * HD;
    extern __shared__ __align__(16) unsigned char smem[];
    uint2* wl = reinterpret_cast<uint2*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < NT * KS * 64; i += kT) wl[i] = wpack[i];
    __syncthreads();
    auto W = [&](int t, int s) { return __builtin_bit_cast(half4_t, wl[(t * KS + s) * 64 + lane]); };
    float16_t G[HD];
    float nq[HD], nk[HD];
#pragma unroll
    for (int hd = 0; hd < HD; ++hd) {
#pragma unroll
        for (int v = 0; v < 16; ++v) G[hd][v] = 0.f;
        nq[hd] = 0.f;
        nk[hd] = 0.f;
    }
    const size_t ntiles = (n + 31) / 32;
    const int p = lane & 31, h = lane >> 5;
    for (size_t tile = (size_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (size_t)gridDim.x * 4) {
        half4_t xf[KS];
        load_x<C>(x, tile, n, lane, xf);
        if constexpr (C > 32) asm volatile("" ::: "memory");  // keep the weight fragments in LDS: hoisted out of this loop they cost 6*C VGPRs
#pragma unroll
        for (int hd = 0; hd < HD; ++hd) {
            float16_t dq, dk;
#pragma unroll
            for (int v = 0; v < 16; ++v) { dq[v] = 0.f; dk[v] = 0.f; }
#pragma unroll
            for (int s = 0; s < KS; ++s) dq = mfma(xf[s], W(hd, s), dq);        // D[pixel][q channel]
#pragma unroll
            for (int s = 0; s < KS; ++s) dk = mfma(xf[s], W(HD + hd, s), dk);   // D[pixel][k channel]
            half4_t aq[4], ak[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    aq[s4][j] = (_Float16)dq[4 * s4 + j];
                    ak[s4][j] = (_Float16)dk[4 * s4 + j];
                    const float fq = (float)aq[s4][j], fk = (float)ak[s4][j];
                    nq[hd] = __builtin_fmaf(fq, fq, nq[hd]);
                    nk[hd] = __builtin_fmaf(fk, fk, nk[hd]);
                }
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) G[hd] = mfma(ak[s4], aq[s4], G[hd]);  // G[i][j] += sum_px k[px][i] q[px][j]
        }
        const size_t row = tile * 32 + p;
#pragma unroll
        for (int hd = 0; hd < HD; ++hd) {
            float16_t dv;
#pragma unroll
            for (int v = 0; v < 16; ++v) dv[v] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) dv = mfma(W(2 * HD + hd, s), xf[s], dv);  // D[v channel (permuted)][pixel]
            _Float16 o[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) o[v] = (_Float16)dv[v];
            if (row < n) store_tile16(v_out + row * (size_t)C + 32 * hd + 16 * h, o);
        }
    }
    // block partial: [hd][34][32] = 32 Gram rows (i = k channel, j = q channel), then sum q^2, then sum k^2
    __syncthreads();  // the weights in LDS are dead: the same bytes hold the per-wave results now
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int hd = 0; hd < HD; ++hd) {
        float* mine = red + ((size_t)wave * HD + hd) * 34 * 32;
#pragma unroll
        for (int v = 0; v < 16; ++v) mine[(8 * (v / 4) + 4 * h + (v % 4)) * 32 + p] = G[hd][v];
        const float sq = nq[hd] + __shfl_xor(nq[hd], 32), sk = nk[hd] + __shfl_xor(nk[hd], 32);
        if (h == 0) { mine[32 * 32 + p] = sq; mine[33 * 32 + p] = sk; }
    }
    __syncthreads();
    for (int i = tid; i < HD * 34 * 32; i += kT) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[(size_t)w * HD * 34 * 32 + i];
        partial[(size_t)blockIdx.x * HD * 34 * 32 + i] = s;
    }
}

// one wave per output element over the block partials (same shape as csrc/mst.hip's: deterministic tree)
__global__ void k_mst_qkv_final(const float* __restrict__ partial, int nblocks, int heads, float* gram /*[heads][32][32]*/, float* nq /*[heads*32]*/,
                                float* nk) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (i >= heads * 34 * 32) return;
    float s = 0.f;
    for (int b = lane; b < nblocks; b += 64) s += partial[(size_t)b * heads * 34 * 32 + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane != 0) return;
    const int hd = i / (34 * 32), e = i - hd * 34 * 32, r = e >> 5, c = e & 31;
    if (r < 32) gram[(hd * 32 + r) * 32 + c] = s;
    else if (r == 32) nq[hd * 32 + c] = __fsqrt_rn(s);
    else nk[hd * 32 + c] = __fsqrt_rn(s);
}

// ---- LayerNorm -> 1x1 conv (C -> 4C) -> GELU -----------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(kT) void k_mst_ffn1(const __half* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                 const uint2* __restrict__ wpack /*[4*HD][KS][64]*/, size_t n, __half* __restrict__ out /*[n][4C]*/) {
    constexpr int KS = C / 8, HD = C / 32, NT = 4 * HD;
    extern __shared__ __align__(16) unsigned char smem[];
    uint2* wl = reinterpret_cast<uint2*>(smem);
    float* gl = reinterpret_cast<float*>(smem + (size_t)NT * KS * 64 * sizeof(uint2));  // [C] gamma, [C] beta
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < NT * KS * 64; i += kT) wl[i] = wpack[i];
    for (int i = tid; i < C; i += kT) { gl[i] = gamma[i]; gl[C + i] = beta[i]; }
    __syncthreads();
    auto W = [&](int t, int s) { return __builtin_bit_cast(half4_t, wl[(t * KS + s) * 64 + lane]); };
    const size_t ntiles = (n + 31) / 32;
    const int p = lane & 31, h = lane >> 5;
    const float cnt = (float)(HD * 31);
    for (size_t tile = (size_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (size_t)gridDim.x * 4) {
        half4_t xf[KS];
        load_x<C>(x, tile, n, lane, xf);
        if constexpr (C > 32) asm volatile("" ::: "memory");  // weight fragments stay in LDS (see k_mst_qkv)
        // LayerNorm over the 31 real channels of every 32-wide group (channel c of this lane: h*C/2 + 4s + j; c % 32 == 31
        // is the zero padding): two-pass float32 statistics, the row is split over lanes p and p + 32
        auto is_pad = [&](int st, int j) { return ((h * (C / 2) + 4 * st + j) & 31) == 31; };
        float s1 = 0.f;
#pragma unroll
        for (int st = 0; st < KS; ++st)
#pragma unroll
            for (int j = 0; j < 4; ++j) s1 += is_pad(st, j) ? 0.f : (float)xf[st][j];
        const float mean = (s1 + __shfl_xor(s1, 32)) / cnt;
        float s2 = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = is_pad(s, j) ? 0.f : (float)xf[s][j] - mean;
                s2 = __builtin_fmaf(d, d, s2);
            }
        const float rstd = rsqrtf((s2 + __shfl_xor(s2, 32)) / cnt + eps);
        half4_t yf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float4 g4 = *reinterpret_cast<const float4*>(gl + h * (C / 2) + 4 * s);
            const float4 b4 = *reinterpret_cast<const float4*>(gl + C + h * (C / 2) + 4 * s);
            const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                yf[s][j] = is_pad(s, j) ? (_Float16)0.f : (_Float16)(((float)xf[s][j] - mean) * rstd * gg[j] + bb[j]);
            }
        }
        const size_t row = tile * 32 + p;
        __half* orow = out + row * (size_t)(4 * C) + 16 * h;
#pragma unroll 2
        for (int t = 0; t < NT; ++t) {
            float16_t d;
#pragma unroll
            for (int v = 0; v < 16; ++v) d[v] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) d = mfma(W(t, s), yf[s], d);  // D[hidden channel (permuted)][pixel]
            _Float16 o[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) o[v] = (_Float16)gelu_erf_h((float)(_Float16)d[v]);
            if (row < n) store_tile16(orow + 32 * t, o);
        }
    }
}

template <int C>
int launch_qkv(avx_ctx* ctx, avx_ws* ws, const void* x, const void* wpack, size_t n, void* v_out, float* gram, float* nq, float* nk, hipStream_t s) {
    constexpr int KS = C / 8, HD = C / 32;
    const size_t wbytes = (size_t)3 * HD * KS * 64 * sizeof(uint2), rbytes = sizeof(float) * 4 * HD * 34 * 32;
    const size_t lds = wbytes > rbytes ? wbytes : rbytes;
    const size_t ntiles = (n + 31) / 32;
    long blocks = (long)((ntiles + 3) / 4);
    const long cap = (long)ctx->num_cus * (lds > 64 * 1024 ? 1 : (lds > 32 * 1024 ? 2 : 4));
    if (blocks > cap) blocks = cap;
    int rc = avx_ensure_scratch(ctx, ws, sizeof(float) * (size_t)blocks * HD * 34 * 32);
    if (rc) return rc;
    float* partial = (float*)ws->d_scratch;
    auto k = k_mst_qkv<C>;
    AVX_HIP(ctx, hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kT), lds, s, (const __half*)x, (const uint2*)wpack, n, (__half*)v_out, partial);
    AVX_HIP(ctx, hipGetLastError());
    const int tot = HD * 34 * 32;
    hipLaunchKernelGGL(k_mst_qkv_final, dim3((tot * 64 + 255) / 256), dim3(256), 0, s, partial, (int)blocks, HD, gram, nq, nk);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

template <int C>
int launch_ffn1(avx_ctx* ctx, const void* x, const float* gamma, const float* beta, float eps, const void* wpack, size_t n, void* out, hipStream_t s) {
    constexpr int KS = C / 8, HD = C / 32;
    const size_t lds = (size_t)4 * HD * KS * 64 * sizeof(uint2) + sizeof(float) * 2 * C;
    const size_t ntiles = (n + 31) / 32;
    long blocks = (long)((ntiles + 3) / 4);
    const long cap = (long)ctx->num_cus * (lds > 64 * 1024 ? 1 : (lds > 32 * 1024 ? 2 : 8));
    if (blocks > cap) blocks = cap;
    auto k = k_mst_ffn1<C>;
    AVX_HIP(ctx, hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kT), lds, s, (const __half*)x, gamma, beta, eps, (const uint2*)wpack, n, (__half*)out);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

}  // namespace

extern "C" {

int avx_mst_qkv_gram(avx_ctx* ctx, const void* x, const void* wpack, size_t n_pix, int C, void* v_out, float* gram, float* nq, float* nk, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, x && wpack && v_out && gram && nq && nk && n_pix > 0, "avx_mst_qkv_gram: NULL pointer or empty tensor");
    AVX_REQUIRE(ctx, C == 32 || C == 64 || C == 128, "avx_mst_qkv_gram: C=%d (32, 64 or 128: 31-channel groups stored 32 wide)", C);
    AVX_REQUIRE(ctx, ((((uintptr_t)x | (uintptr_t)wpack | (uintptr_t)v_out)) & 15u) == 0, "avx_mst_qkv_gram: pointers must be 16-byte aligned");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    avx_ws* ws = avx_workspace(ctx, s);
    if (!ws) return AVX_ERR_NOMEM;
    if (C == 32) return launch_qkv<32>(ctx, ws, x, wpack, n_pix, v_out, gram, nq, nk, s);
    if (C == 64) return launch_qkv<64>(ctx, ws, x, wpack, n_pix, v_out, gram, nq, nk, s);
    return launch_qkv<128>(ctx, ws, x, wpack, n_pix, v_out, gram, nq, nk, s);
}

int avx_mst_ln_gemm_gelu(avx_ctx* ctx, const void* x, const float* gamma, const float* beta, float eps, const void* wpack, size_t rows, int C, void* out,
                         void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, x && gamma && beta && wpack && out && rows > 0, "avx_mst_ln_gemm_gelu: NULL pointer or empty tensor");
    AVX_REQUIRE(ctx, C == 32 || C == 64 || C == 128, "avx_mst_ln_gemm_gelu: C=%d (32, 64 or 128: 31-channel groups stored 32 wide)", C);
    AVX_REQUIRE(ctx, ((((uintptr_t)x | (uintptr_t)wpack | (uintptr_t)out)) & 15u) == 0, "avx_mst_ln_gemm_gelu: pointers must be 16-byte aligned");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    if (C == 32) return launch_ffn1<32>(ctx, x, gamma, beta, eps, wpack, rows, out, s);
    if (C == 64) return launch_ffn1<64>(ctx, x, gamma, beta, eps, wpack, rows, out, s);
    return launch_ffn1<128>(ctx, x, gamma, beta, eps, wpack, rows, out, s);
}

}  // extern "C"
