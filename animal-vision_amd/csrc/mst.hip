// csrc/mst.hip -- hand-written gfx950 kernels for the memory-bound parts of the MST++ forward pass
// (reference: ml/MST_plus_plus/predict_code/architecture/MST_Plus_Plus.py).  PyTorch-ROCm keeps the dense
// GEMMs/convs; these replace what its eager ops do badly on 31/62/124-channel NHWC tensors (rocprofv3 of
// the torch-only forward at 1080p: 33 % strided vector_norm reduce, 30 % MIOpen naive NHWC depthwise conv,
// 14 % LayerNorm moments on a float copy - profiles/r01/mstpp_1080p_kernel_stats.csv):
//
//   avx_mst_gram        MS_MSA :127-129  k^T q over ALL pixels per head + the L2 norms of every q/k column,
//                       in ONE pass over the (N x 3C) qkv tensor (F.normalize is folded into the d x d result)
//   avx_dwconv3x3_nhwc  depthwise 3x3 (pos_emb :104-106, FeedForward :147) on channels-last tensors,
//                       fp32 accumulate, optional exact-erf GELU on the output
//   avx_layernorm_rows  PreNorm's nn.LayerNorm(dim) (:57-65) over the last dim of an (N x C) tensor,
//                       fp32 statistics straight from the fp16 rows
// Tensors are torch tensors' data_ptr()s; dtype 0 = float32, 1 = float16.
#include <hip/hip_fp16.h>

#include "avx_internal.h"

namespace {

constexpr int kMT = 256;

template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<__half>(const __half* p) { return __half2float(*p); }
template <typename T> __device__ __forceinline__ void st(T* p, float v);
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<__half>(__half* p, float v) { *p = __float2half(v); }

// ---- Gram + column norms ----------------------------------------------------------------------------
// Per head (d = C/heads = 31, padded to 32): G[i][j] = sum_n k[n][i] q[n][j];  sq[j] = sum_n q[n][j]^2 (same for k).
// A workgroup walks its pixel range in tiles of PB pixels staged in LDS as float [p][head*32 + i]; each wave
// owns one head (x a slice of the tile when heads < 4) and each lane a 4x4 register tile of G.
constexpr int PB = 64;

template <typename T>
__global__ __launch_bounds__(kMT) void k_mst_gram(const T* __restrict__ qkv, size_t n, int C, int heads, size_t px_per_block,
                                                  float* __restrict__ partial /*[blocks][heads][34][32]*/) {
    extern __shared__ float sm[];
    const int CP = heads * 32;
    float* sq = sm;             // [PB][CP]
    float* sk = sm + PB * CP;   // [PB][CP]
    const int d = C / heads;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int head = wave % heads, slice = wave / heads, nslices = 4 / heads;  // heads in {1,2,4}
    const int it = lane >> 3, jt = lane & 7;  // 4x4 tile origin (4*it, 4*jt)
    float acc[4][4] = {};
    float aq[4] = {}, ak[4] = {};
    const size_t p_begin = (size_t)blockIdx.x * px_per_block;
    const size_t p_end = p_begin + px_per_block < n ? p_begin + px_per_block : n;
    for (size_t p0 = p_begin; p0 < p_end; p0 += PB) {
        const int np = (int)(p_end - p0 < PB ? p_end - p0 : PB);
        __syncthreads();
        for (int i = tid; i < PB * CP; i += kMT) {
            const int p = i / CP, cp = i - p * CP;
            const int h = cp >> 5, ii = cp & 31;
            float vq = 0.f, vk = 0.f;
            if (p < np && ii < d) {
                const T* row = qkv + (p0 + p) * (size_t)(3 * C);
                vq = ld<T>(row + h * d + ii);
                vk = ld<T>(row + C + h * d + ii);
            }
            sq[i] = vq;
            sk[i] = vk;
        }
        __syncthreads();
        const int per = PB / nslices;
        for (int p = slice * per; p < (slice + 1) * per; ++p) {
            const float4 kv = *reinterpret_cast<const float4*>(sk + p * CP + head * 32 + 4 * it);
            const float4 qv = *reinterpret_cast<const float4*>(sq + p * CP + head * 32 + 4 * jt);
            const float k4[4] = {kv.x, kv.y, kv.z, kv.w}, q4[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_fmaf(k4[a], q4[b], acc[a][b]);
            if (it == 0)
#pragma unroll
                for (int b = 0; b < 4; ++b) aq[b] = __builtin_fmaf(q4[b], q4[b], aq[b]);
            if (jt == 0)
#pragma unroll
                for (int a = 0; a < 4; ++a) ak[a] = __builtin_fmaf(k4[a], k4[a], ak[a]);
        }
    }
    // combine the slices of a head through LDS (fixed order: deterministic), then write the block partial
    __syncthreads();
    float* red = sm;  // [4 waves][34][32]
    {
        float* mine = red + wave * 34 * 32;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) mine[(4 * it + a) * 32 + 4 * jt + b] = acc[a][b];
        if (it == 0)
#pragma unroll
            for (int b = 0; b < 4; ++b) mine[32 * 32 + 4 * jt + b] = aq[b];
        if (jt == 0)
#pragma unroll
            for (int a = 0; a < 4; ++a) mine[33 * 32 + 4 * it + a] = ak[a];
    }
    __syncthreads();
    for (int i = tid; i < heads * 34 * 32; i += kMT) {
        const int h = i / (34 * 32), e = i - h * 34 * 32;
        float s = 0.f;
        for (int sl = 0; sl < nslices; ++sl) s += red[(sl * heads + h) * 34 * 32 + e];
        partial[((size_t)blockIdx.x * heads + h) * 34 * 32 + e] = s;
    }
}

// one wave per output element: lanes stride over the block partials, fixed-shape shuffle tree (deterministic)
__global__ void k_mst_gram_final(const float* __restrict__ partial, int nblocks, int heads, int d, float* gram /*[heads][32][32]*/,
                                 float* nq /*[heads*d]*/, float* nk) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (i >= heads * 34 * 32) return;
    const int h = i / (34 * 32), e = i - h * 34 * 32;
    float s = 0.f;
    for (int b = lane; b < nblocks; b += 64) s += partial[((size_t)b * heads + h) * 34 * 32 + e];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane != 0) return;
    const int r = e >> 5, c = e & 31;
    if (r < 32) gram[(h * 32 + r) * 32 + c] = s;
    else if (r == 32) { if (c < d) nq[h * d + c] = __fsqrt_rn(s); }
    else { if (c < d) nk[h * d + c] = __fsqrt_rn(s); }
}

// ---- depthwise 3x3, NHWC, zero padding 1, stride 1, optional GELU -----------------------------------
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// exact-erf GELU for a float16 RESULT: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, three orders below half
// an fp16 ulp of the output), one v_exp + one v_rcp instead of libm's erff polynomial ladder
__device__ __forceinline__ float gelu_erf_h(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = 1.0f - poly * __expf(-z * z);
    return 0.5f * x * (1.0f + (x < 0.f ? -e : e));
}

template <typename T, int VEC>
__global__ __launch_bounds__(kMT) void k_dwconv3x3(const T* __restrict__ x, const float* __restrict__ w_g /*[C][9]*/, T* __restrict__ y, int B,
                                                   int H, int W, int C, int gelu) {
    extern __shared__ float w[];  // [9][C] (tap-major so a channel group reads contiguous weights)
    for (int i = threadIdx.x; i < 9 * C; i += kMT) {
        const int c = i / 9, t = i - 9 * c;
        w[t * C + c] = w_g[i];
    }
    __syncthreads();
    const int CV = C / VEC;
    const size_t total = (size_t)B * H * W * CV;
    for (size_t i = (size_t)blockIdx.x * kMT + threadIdx.x; i < total; i += (size_t)gridDim.x * kMT) {
        const int cv = (int)(i % CV);
        const size_t pix = i / CV;
        const int xw = (int)(pix % W);
        const int yh = (int)((pix / W) % H);
        const size_t b = pix / ((size_t)W * H);
        const int c0 = cv * VEC;
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = yh + ky - 1;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = xw + kx - 1;
                if (xx < 0 || xx >= W) continue;
                const T* src = x + ((b * H + yy) * (size_t)W + xx) * C + c0;
                const float* wt = w + (ky * 3 + kx) * C + c0;
                if constexpr (VEC == 4 && sizeof(T) == 2) {
                    const uint2 raw = *reinterpret_cast<const uint2*>(src);
                    const __half2 h01 = *reinterpret_cast<const __half2*>(&raw.x), h23 = *reinterpret_cast<const __half2*>(&raw.y);
                    const float2 f01 = __half22float2(h01), f23 = __half22float2(h23);
                    acc[0] = __builtin_fmaf(f01.x, wt[0], acc[0]); acc[1] = __builtin_fmaf(f01.y, wt[1], acc[1]);
                    acc[2] = __builtin_fmaf(f23.x, wt[2], acc[2]); acc[3] = __builtin_fmaf(f23.y, wt[3], acc[3]);
                } else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] = __builtin_fmaf(ld<T>(src + v), wt[v], acc[v]);
                }
            }
        }
        T* dst = y + pix * C + c0;
        if constexpr (VEC == 4 && sizeof(T) == 2) {
            float o[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) o[v] = gelu ? gelu_erf(acc[v]) : acc[v];
            const __half2 h01 = __floats2half2_rn(o[0], o[1]), h23 = __floats2half2_rn(o[2], o[3]);
            uint2 raw;
            raw.x = *reinterpret_cast<const uint32_t*>(&h01);
            raw.y = *reinterpret_cast<const uint32_t*>(&h23);
            *reinterpret_cast<uint2*>(dst) = raw;
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) st<T>(dst + v, gelu ? gelu_erf(acc[v]) : acc[v]);
        }
    }
}

// fp16, C % 8 == 0: a thread owns ONE group of 8 channels (its 72 weights live in registers for the whole launch) and
// walks (x, 4-row strip) items: each of the 6 input rows of a strip is loaded once (3 pixels x 16 B) and feeds up to
// three output rows, so 4.5 16-byte loads serve 8 channels x 4 outputs (the one-output-per-thread form needs 9 8-byte
// loads + 36 LDS weight reads for 4 channels).  float32 accumulation, exact-erf GELU.
constexpr int kDwR = 4;
__global__ __launch_bounds__(kMT) void k_dwconv3x3_h8(const __half* __restrict__ x, const float* __restrict__ w_g /*[C][9]*/, __half* __restrict__ y, int B, int H,
                                                     int W, int C, int gelu, const __half* __restrict__ res /*same shape as y, or NULL*/,
                                                     const float* __restrict__ bias /*[C] or NULL*/) {
    const int CV = C / 8;
    const int cg = (int)(((size_t)blockIdx.x * kMT + threadIdx.x) % CV);  // fixed per thread: grid stride is a multiple of CV
    float wt[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int v = 0; v < 8; ++v) wt[t][v] = w_g[(size_t)(cg * 8 + v) * 9 + t];
    float bs[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) bs[v] = bias ? bias[cg * 8 + v] : 0.f;
    const int strips = (H + kDwR - 1) / kDwR;
    const size_t total = (size_t)B * strips * W * CV;
    for (size_t i = (size_t)blockIdx.x * kMT + threadIdx.x; i < total; i += (size_t)gridDim.x * kMT) {
        const size_t p = i / CV;
        const int xw = (int)(p % W);
        const int st = (int)((p / W) % strips);
        const size_t b = p / ((size_t)W * strips);
        const int y0 = st * kDwR;
        float acc[kDwR][8];
#pragma unroll
        for (int r = 0; r < kDwR; ++r)
#pragma unroll
            for (int v = 0; v < 8; ++v) acc[r][v] = 0.f;
        // Branch-free loads: every tap reads a CLAMPED (always valid) address and out-of-image taps are zeroed with a select
        // afterwards, so the 18 loads of an item are in flight together (a guarded load per tap made each wait for the last).
        uint4 raw[kDwR + 2][3];
#pragma unroll
        for (int rr = 0; rr < kDwR + 2; ++rr) {  // input row y0 - 1 + rr
            const int yy = y0 - 1 + rr;
            const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
            const __half* rowp = x + ((b * H + yc) * (size_t)W) * C + (size_t)cg * 8;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = xw + kx - 1;
                const int xxc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
                raw[rr][kx] = *reinterpret_cast<const uint4*>(rowp + (size_t)xxc * C);
            }
        }
#pragma unroll
        for (int rr = 0; rr < kDwR + 2; ++rr) {
            const int yy = y0 - 1 + rr;
            const bool yok = yy >= 0 && yy < H;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = xw + kx - 1;
                const bool ok = yok && xx >= 0 && xx < W;
                uint4 rw = raw[rr][kx];
                rw.x = ok ? rw.x : 0u; rw.y = ok ? rw.y : 0u; rw.z = ok ? rw.z : 0u; rw.w = ok ? rw.w : 0u;
                const __half2* hp = reinterpret_cast<const __half2*>(&rw);
                float f[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float2 t2 = __half22float2(hp[q]); f[2 * q] = t2.x; f[2 * q + 1] = t2.y; }
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {  // this input row is tap row ky of output row r = rr - ky
                    const int r = rr - ky;
                    if (r < 0 || r >= kDwR) continue;
#pragma unroll
                    for (int v = 0; v < 8; ++v) acc[r][v] = __builtin_fmaf(f[v], wt[ky * 3 + kx][v], acc[r][v]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < kDwR; ++r) {
            const int yo = y0 + r;
            if (yo >= H) break;
            __half2 o[4];
            const size_t off = ((b * H + yo) * (size_t)W + xw) * C + (size_t)cg * 8;
            uint4 rraw = make_uint4(0, 0, 0, 0);
            if (res) rraw = *reinterpret_cast<const uint4*>(res + off);
            const __half2* rp = reinterpret_cast<const __half2*>(&rraw);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float a0 = acc[r][2 * q], a1 = acc[r][2 * q + 1];
                if (gelu) { a0 = gelu_erf_h(a0); a1 = gelu_erf_h(a1); }
                const float2 rr = __half22float2(rp[q]);  // epilogue: + residual + per-channel bias (both optional, zero when absent)
                o[q] = __floats2half2_rn(a0 + rr.x + bs[2 * q], a1 + rr.y + bs[2 * q + 1]);
            }
            *reinterpret_cast<uint4*>(y + off) = *reinterpret_cast<const uint4*>(o);
        }
    }
}

// ---- LayerNorm over the last dim of (rows x C), biased variance, fp32 statistics ---------------------
template <typename T>
__global__ __launch_bounds__(kMT) void k_layernorm(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   T* __restrict__ y, size_t rows, int C, float eps, int RB, int group, int real) {
    extern __shared__ float tile[];  // [RB][CP], CP odd: a lane per row walks its row without bank conflicts
    const int CP = C | 1;
    for (size_t r0 = (size_t)blockIdx.x * RB; r0 < rows; r0 += (size_t)gridDim.x * RB) {
        const int nr = (int)(rows - r0 < (size_t)RB ? rows - r0 : RB);
        __syncthreads();
        for (int i = threadIdx.x; i < nr * C; i += kMT) {  // coalesced: the RB rows are one contiguous span
            const int r = i / C, c = i - r * C;
            tile[r * CP + c] = ld<T>(x + r0 * C + i);
        }
        __syncthreads();
        if ((int)threadIdx.x < nr) {
            float* row = tile + threadIdx.x * CP;
            // channels c with (c % group) >= real are zero padding (31-channel groups stored 32 wide): not part of the
            // statistics, and they stay zero
            const float cnt = (float)(C / group * real);
            float s = 0.f;
            for (int c = 0; c < C; ++c) s += (c % group) < real ? row[c] : 0.f;
            const float mean = s / cnt;
            float v = 0.f;
            for (int c = 0; c < C; ++c) { const float dlt = (c % group) < real ? row[c] - mean : 0.f; v = __builtin_fmaf(dlt, dlt, v); }
            const float rstd = rsqrtf(v / cnt + eps);
            for (int c = 0; c < C; ++c) row[c] = (c % group) < real ? (row[c] - mean) * rstd * gamma[c] + beta[c] : 0.f;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nr * C; i += kMT) {
            const int r = i / C, c = i - r * C;
            st<T>(y + r0 * C + i, tile[r * CP + c]);
        }
    }
}

}  // namespace

extern "C" {

int avx_mst_gram(avx_ctx* ctx, const void* qkv, int dtype, size_t n_pix, int C, int heads, float* gram, float* nq, float* nk, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, qkv && gram && nq && nk && n_pix > 0, "avx_mst_gram: NULL pointer or empty tensor");
    AVX_REQUIRE(ctx, (heads == 1 || heads == 2 || heads == 4) && C % heads == 0 && C / heads <= 32, "avx_mst_gram: C=%d heads=%d unsupported", C, heads);
    AVX_REQUIRE(ctx, dtype == 0 || dtype == 1, "avx_mst_gram: dtype");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    avx_ws* ws = avx_workspace(ctx, s);
    if (!ws) return AVX_ERR_NOMEM;
    long blocks = (long)((n_pix + 8 * PB - 1) / (8 * PB));
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    size_t per = (n_pix + blocks - 1) / blocks;
    per = (per + PB - 1) / PB * PB;
    blocks = (long)((n_pix + per - 1) / per);
    const size_t pbytes = sizeof(float) * (size_t)blocks * heads * 34 * 32;
    int rc = avx_ensure_scratch(ctx, ws, pbytes);
    if (rc) return rc;
    float* partial = (float*)ws->d_scratch;
    size_t lds = sizeof(float) * 2 * PB * heads * 32;
    const size_t red = sizeof(float) * 4 * 34 * 32;
    if (lds < red) lds = red;
    const int d = C / heads;
    if (dtype == 0) {
        AVX_HIP(ctx, hipFuncSetAttribute((const void*)k_mst_gram<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_mst_gram<float>, dim3((unsigned)blocks), dim3(kMT), lds, s, (const float*)qkv, n_pix, C, heads, per, partial);
    } else {
        AVX_HIP(ctx, hipFuncSetAttribute((const void*)k_mst_gram<__half>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_mst_gram<__half>, dim3((unsigned)blocks), dim3(kMT), lds, s, (const __half*)qkv, n_pix, C, heads, per, partial);
    }
    const int tot = heads * 34 * 32;
    hipLaunchKernelGGL(k_mst_gram_final, dim3((tot * 64 + 255) / 256), dim3(256), 0, s, partial, (int)blocks, heads, d, gram, nq, nk);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

static int dwconv_impl(avx_ctx* ctx, const void* x, const float* w_c9, void* y, int dtype, int B, int H, int W, int C, int gelu_out, const void* res,
                       const float* bias, void* stream);

int avx_dwconv3x3_nhwc(avx_ctx* ctx, const void* x, const float* w_c9, void* y, int dtype, int B, int H, int W, int C, int gelu_out, void* stream) {
    return dwconv_impl(ctx, x, w_c9, y, dtype, B, H, W, C, gelu_out, nullptr, nullptr, stream);
}

int avx_dwconv3x3_nhwc_add(avx_ctx* ctx, const void* x, const float* w_c9, void* y, int B, int H, int W, int C, int gelu_out, const void* residual,
                           const float* bias, void* stream) {
    return dwconv_impl(ctx, x, w_c9, y, 1, B, H, W, C, gelu_out, residual, bias, stream);
}

static int dwconv_impl(avx_ctx* ctx, const void* x, const float* w_c9, void* y, int dtype, int B, int H, int W, int C, int gelu_out, const void* res,
                       const float* bias, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, x && w_c9 && y && x != y && B > 0 && H > 0 && W > 0 && C > 0 && C <= 1024, "avx_dwconv3x3_nhwc: bad arguments");
    AVX_REQUIRE(ctx, dtype == 0 || dtype == 1, "avx_dwconv3x3_nhwc: dtype");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    const size_t lds = sizeof(float) * 9 * C;
    const bool vec4 = dtype == 1 && (C % 4 == 0) && (((uintptr_t)x | (uintptr_t)y) & 7u) == 0;
    const size_t items = (size_t)B * H * W * (vec4 ? C / 4 : C);
    size_t g = (items + kMT - 1) / kMT;
    const size_t cap = (size_t)ctx->num_cus * 16;
    if (g > cap) g = cap;
    const bool vec8 = dtype == 1 && (C % 8 == 0) && (kMT % (C / 8) == 0) && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)res) & 15u) == 0;
    if (vec8) {
        const size_t it8 = (size_t)B * ((H + kDwR - 1) / kDwR) * W * (C / 8);
        size_t g8 = (it8 + kMT - 1) / kMT;
        if (g8 > cap) g8 = cap;
        hipLaunchKernelGGL(k_dwconv3x3_h8, dim3((unsigned)g8), dim3(kMT), 0, s, (const __half*)x, w_c9, (__half*)y, B, H, W, C, gelu_out, (const __half*)res, bias);
        AVX_HIP(ctx, hipGetLastError());
        return AVX_OK;
    }
    AVX_REQUIRE(ctx, !res && !bias, "avx_dwconv3x3_nhwc_add: the fused epilogue needs float16, C %% 8 == 0 and 16-byte aligned tensors");
    if (dtype == 0) hipLaunchKernelGGL((k_dwconv3x3<float, 1>), dim3((unsigned)g), dim3(kMT), lds, s, (const float*)x, w_c9, (float*)y, B, H, W, C, gelu_out);
    else if (vec4) hipLaunchKernelGGL((k_dwconv3x3<__half, 4>), dim3((unsigned)g), dim3(kMT), lds, s, (const __half*)x, w_c9, (__half*)y, B, H, W, C, gelu_out);
    else hipLaunchKernelGGL((k_dwconv3x3<__half, 1>), dim3((unsigned)g), dim3(kMT), lds, s, (const __half*)x, w_c9, (__half*)y, B, H, W, C, gelu_out);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

static int layernorm_impl(avx_ctx* ctx, const void* x, const float* gamma, const float* beta, void* y, int dtype, size_t rows, int C, float eps, int group,
                          int real, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, x && gamma && beta && y && rows > 0 && C > 0 && C <= 1024, "avx_layernorm_rows: bad arguments");
    AVX_REQUIRE(ctx, dtype == 0 || dtype == 1, "avx_layernorm_rows: dtype");
    AVX_REQUIRE(ctx, group >= 1 && real >= 1 && real <= group && C % group == 0, "avx_layernorm_rows: bad channel grouping");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    const int CP = C | 1;
    int RB = kMT;
    while ((size_t)RB * CP * sizeof(float) > 48 * 1024 && RB > 16) RB /= 2;
    const size_t lds = sizeof(float) * (size_t)RB * CP;
    size_t g = (rows + RB - 1) / RB;
    const size_t cap = (size_t)ctx->num_cus * 8;
    if (g > cap) g = cap;
    if (dtype == 0) {
        AVX_HIP(ctx, hipFuncSetAttribute((const void*)k_layernorm<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_layernorm<float>, dim3((unsigned)g), dim3(kMT), lds, s, (const float*)x, gamma, beta, (float*)y, rows, C, eps, RB, group, real);
    } else {
        AVX_HIP(ctx, hipFuncSetAttribute((const void*)k_layernorm<__half>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_layernorm<__half>, dim3((unsigned)g), dim3(kMT), lds, s, (const __half*)x, gamma, beta, (__half*)y, rows, C, eps, RB, group, real);
    }
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

int avx_layernorm_rows(avx_ctx* ctx, const void* x, const float* gamma, const float* beta, void* y, int dtype, size_t rows, int C, float eps,
                       void* stream) {
    return layernorm_impl(ctx, x, gamma, beta, y, dtype, rows, C, eps, C, C, stream);
}

int avx_layernorm_rows_grouped(avx_ctx* ctx, const void* x, const float* gamma, const float* beta, void* y, int dtype, size_t rows, int C, float eps,
                               int group, int real, void* stream) {
    return layernorm_impl(ctx, x, gamma, beta, y, dtype, rows, C, eps, group, real, stream);
}

}  // extern "C"
