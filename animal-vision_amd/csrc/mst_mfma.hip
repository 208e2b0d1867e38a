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
#include "mst_common.h"

namespace {

constexpr int kT = 256;  // 4 waves, each marching over its own 32-pixel tiles

__device__ __forceinline__ float16_t mfma(half4_t a, half4_t b, float16_t c) { return __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float16_t mfma16(half8_t a, half8_t b, float16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

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
        if (v_out)  // NULL: the attention tail forms v from x itself (avx_mst_attn_tail with wv): this pass only reads
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
    // a wave sums FOUR neighbouring entries over the workgroups' partials: lane l takes workgroups l, l + 64, ... with 16-byte loads (a quarter
    // of each 64-byte sector it touches instead of a sixteenth: the one-entry form moved 32x the bytes it summed), then the lanes combine
    const int i0 = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 4, lane = threadIdx.x & 63, tot = heads * 34 * 32;
    if (i0 >= tot) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = lane; b < nblocks; b += 64) {
        const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)b * tot + i0);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s.x += __shfl_xor(s.x, o); s.y += __shfl_xor(s.y, o); s.z += __shfl_xor(s.z, o); s.w += __shfl_xor(s.w, o); }
    if (lane != 0) return;
    const float r4[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = i0 + q, hd = i / (34 * 32), e = i - hd * 34 * 32, r = e >> 5, c = e & 31;
        if (r < 32) gram[(hd * 32 + r) * 32 + c] = r4[q];
        else if (r == 32) nq[hd * 32 + c] = __fsqrt_rn(r4[q]);
        else nk[hd * 32 + c] = __fsqrt_rn(r4[q]);
    }
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
            for (int v = 0; v < 16; v += 2) {
                const float2_t g = gelu_fast2(float2_t{(float)(_Float16)d[v], (float)(_Float16)d[v + 1]});
                o[v] = (_Float16)g.x;
                o[v + 1] = (_Float16)g.y;
            }
            if (row < n) store_tile16(orow + 32 * t, o);
        }
    }
}

// ---- depthwise 3x3 -> GELU -> 1x1 conv (4C -> C) -> + residual -------------------------------------------------------
// FeedForward :147-149 and the block's `+ x` (:184).  A workgroup owns a tile of ROWS rows x 32 pixels:
//   phase 1 (all 256 threads, thread = one group of 8 hidden channels x a few pixels, its 72 taps in registers for the
//            whole launch): depthwise 3x3 on ROWS-row strips straight from global memory (each input row of a strip is
//            loaded once and feeds up to three output rows), packed float32 FMAs, exact-erf GELU, float16 result into an
//            LDS tile [ROWS][32 px][4C] (row pitch padded by 16 B: the fragment reads below are conflict-free);
//   phase 2 (one (row, 32-channel output tile) per wave): the tile is operand B of "channels x pixels" MFMAs against
//            the pre-packed W2 fragments, then + residual, float16, 32 contiguous bytes per lane.
// The (N x 4C) GELU output never reaches memory.

template <int C, int ROWS, int PXT, bool PF, int MINW>
__global__ __launch_bounds__(kT, MINW) void k_mst_ffn2(const __half* __restrict__ hid /*[B][H][W][4C]*/, const float* __restrict__ w9 /*[4C][9]*/,
                                                    const uint2* __restrict__ w2pack /*[C/32][4C/8][64]*/, const __half* res /*[B][H][W][C]*/,
                                                    __half* out /*may alias res*/, int B, int H, int W) {
    constexpr int C4 = 4 * C, CV = C4 / 8, PXS = kT / CV, TW = 32 * PXT, IPT = TW / PXS, KS2 = C4 / 8, NT = C / 32;
    constexpr int PITCH = C4 * 2 + 16;  // bytes per pixel of the LDS tile
    static_assert(ROWS * PXT * NT == 4, "one (row, 32-pixel group, output tile) per wave");
    static_assert(!PF || IPT % 2 == 0, "the prefetching schedule alternates two register buffers");
    extern __shared__ __align__(16) unsigned char smem[];
    uint2* wl = reinterpret_cast<uint2*>(smem);
    float* w9l = reinterpret_cast<float*>(smem + (size_t)NT * KS2 * 64 * sizeof(uint2));  // [9][4C] depthwise taps, tap-major
    unsigned char* tileb = reinterpret_cast<unsigned char*>(w9l + 9 * C4);                 // [ROWS][TW][PITCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < NT * KS2 * 64; i += kT) wl[i] = w2pack[i];
    for (int i = tid; i < 9 * C4; i += kT) { const int c = i / 9, t = i - 9 * c; w9l[t * C4 + c] = w9[i]; }
    const int cg = tid % CV, pslot = tid / CV;
    const int xt = (W + TW - 1) / TW, strips = (H + ROWS - 1) / ROWS;
    const long total = (long)B * strips * xt;
    const int p = lane & 31, h = lane >> 5;
    const int mr = wave / (PXT * NT), mp = (wave / NT) % PXT, mt = wave % NT;  // phase 2: this wave's row, 32-pixel group and output-channel tile
    struct Pos { int x0, y0; long b; };
    auto pos_of = [&](long tile) { return Pos{(int)(tile % xt) * TW, (int)((tile / xt) % strips) * ROWS, tile / ((long)xt * strips)}; };
    // Branch-free loads: every tap reads a CLAMPED (always valid) address and out-of-image taps are zeroed with a select when
    // they are consumed, so all 3 * (ROWS + 2) loads of an item are in flight together (a guarded load per tap made each one
    // wait for the previous).  Latency is hidden by occupancy: the taps live in LDS, not in 72 registers per thread.
    auto issue = [&](const Pos& t, int i, uint4 (&raw)[ROWS + 2][3]) {
        const int xw = t.x0 + pslot + PXS * i, xc = xw < W ? xw : W - 1;
#pragma unroll
        for (int rr = 0; rr < ROWS + 2; ++rr) {  // input row y0 - 1 + rr
            const int yy = t.y0 - 1 + rr;
            const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
            const __half* rowp = hid + ((t.b * H + yc) * (size_t)W) * C4 + (size_t)cg * 8;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = xc + kx - 1;
                const int xxc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
                raw[rr][kx] = *reinterpret_cast<const uint4*>(rowp + (size_t)xxc * C4);
            }
        }
    };
    auto consume = [&](const Pos& t, int i, const uint4 (&raw)[ROWS + 2][3]) {
        const int px = pslot + PXS * i, xw = t.x0 + px, xc = xw < W ? xw : W - 1;
        const bool interior = t.x0 > 0 && t.x0 + TW < W && t.y0 > 0 && t.y0 + ROWS < H;  // uniform: no tap leaves the image
        float2_t acc[ROWS][4];
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[r][q] = float2_t{0.f, 0.f};
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            float2_t wt[3][4];  // this tap column's weights (LDS; lanes of one channel group read the same 32 bytes)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float4 a = *reinterpret_cast<const float4*>(w9l + (ky * 3 + kx) * C4 + cg * 8);
                const float4 c = *reinterpret_cast<const float4*>(w9l + (ky * 3 + kx) * C4 + cg * 8 + 4);
                wt[ky][0] = float2_t{a.x, a.y}; wt[ky][1] = float2_t{a.z, a.w}; wt[ky][2] = float2_t{c.x, c.y}; wt[ky][3] = float2_t{c.z, c.w};
            }
            const int xx = xc + kx - 1;
            const bool xok = xx >= 0 && xx < W;
#pragma unroll
            for (int rr = 0; rr < ROWS + 2; ++rr) {
                const int yy = t.y0 - 1 + rr;
                uint4 rw = raw[rr][kx];
                if (!interior) {
                    const bool ok = xok && yy >= 0 && yy < H;
                    rw.x = ok ? rw.x : 0u; rw.y = ok ? rw.y : 0u; rw.z = ok ? rw.z : 0u; rw.w = ok ? rw.w : 0u;
                }
                const half8_t h8 = __builtin_bit_cast(half8_t, rw);
                float2_t f[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) f[q] = float2_t{(float)h8[2 * q], (float)h8[2 * q + 1]};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {  // this input row is tap row ky of output row r = rr - ky
                    const int r = rr - ky;
                    if (r < 0 || r >= ROWS) continue;
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[r][q] = __builtin_elementwise_fma(f[q], wt[ky][q], acc[r][q]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            half8_t o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float2_t g = gelu_fast2(acc[r][q]);
                o[2 * q] = (_Float16)g.x;
                o[2 * q + 1] = (_Float16)g.y;
            }
            *reinterpret_cast<uint4*>(tileb + ((size_t)r * TW + px) * PITCH + cg * 16) = __builtin_bit_cast(uint4, o);
        }
    };
    for (long tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const Pos t = pos_of(tile);
        __syncthreads();  // phase 2 of the previous tile is done with the LDS tile (and the tables are loaded, first time round)
        // Two schedules, picked per channel count by measurement (1080p: C = 32 385 vs 405 us, C = 64 261 vs 223 us):
        //  PF = false: one item at a time, <= 168 VGPRs, three workgroups per CU hide the load latency;
        //  PF = true : the next item's taps are loaded into a second register buffer while this one computes (252 VGPRs, two
        //              workgroups per CU).  The items are unrolled so the loads in flight can be counted: across a loop
        //              back-edge the compiler falls back to s_waitcnt vmcnt(0) and the prefetch is serialised.
        if constexpr (!PF) {
#pragma unroll 1
            for (int i = 0; i < IPT; ++i) {
                uint4 raw[ROWS + 2][3];
                issue(t, i, raw);
                consume(t, i, raw);
            }
        } else {
            uint4 rawA[ROWS + 2][3], rawB[ROWS + 2][3];
            issue(t, 0, rawA);
#pragma unroll
            for (int i = 0; i < IPT; i += 2) {  // straight-line code: the compiler counts the loads in flight (vmcnt) exactly
                issue(t, i + 1, rawB);
                consume(t, i, rawA);
                if (i + 2 < IPT) issue(t, i + 2, rawA);
                consume(t, i + 1, rawB);
            }
        }
        __syncthreads();
        {
            float16_t d;
#pragma unroll
            for (int v = 0; v < 16; ++v) d[v] = 0.f;
            const unsigned char* src = tileb + ((size_t)mr * TW + mp * 32 + p) * PITCH + h * (C4 / 2) * 2;  // this lane's 2C contiguous hidden channels
#pragma unroll
            for (int s2 = 0; s2 < KS2 / 2; ++s2) {
                const half8_t h8 = __builtin_bit_cast(half8_t, *reinterpret_cast<const uint4*>(src + 16 * s2));
                const half4_t b0 = half4_t{h8[0], h8[1], h8[2], h8[3]}, b1 = half4_t{h8[4], h8[5], h8[6], h8[7]};
                d = mfma(__builtin_bit_cast(half4_t, wl[(mt * KS2 + 2 * s2) * 64 + lane]), b0, d);
                d = mfma(__builtin_bit_cast(half4_t, wl[(mt * KS2 + 2 * s2 + 1) * 64 + lane]), b1, d);
            }
            const int yo = t.y0 + mr, xw = t.x0 + mp * 32 + p;
            if (yo < H && xw < W) {
                const size_t off = ((t.b * H + yo) * (size_t)W + xw) * C + 32 * mt + 16 * h;
                const half8_t r0 = __builtin_bit_cast(half8_t, reinterpret_cast<const uint4*>(res + off)[0]);
                const half8_t r1 = __builtin_bit_cast(half8_t, reinterpret_cast<const uint4*>(res + off)[1]);
                _Float16 o[16];
#pragma unroll
                for (int v = 0; v < 8; ++v) { o[v] = (_Float16)(d[v] + (float)r0[v]); o[8 + v] = (_Float16)(d[8 + v] + (float)r1[v]); }
                store_tile16(out + off, o);
            }
        }
    }
}

// ---- out = [add +] a @ W [+ a2 @ W2] (C -> C) --------------------------------------------------------------------
// MS_MSA's fused `attn @ v -> proj` (one C x C matrix per frame) accumulated onto pos_emb + x, and the decoder's 1x1
// fusion conv over [up | skip] (:257) as two half-K products (the concatenated tensor is never built).
template <int C>
__global__ __launch_bounds__(kT) void k_mst_rowgemm_add(const __half* __restrict__ a, const uint2* __restrict__ wpack /*[C/32][KS][64]*/,
                                                        const __half* __restrict__ a2 /*or NULL*/, const uint2* __restrict__ wpack2, const __half* add /*or NULL*/,
                                                        __half* out /*may alias add*/, size_t n) {
    constexpr int KS = C / 8, NT = C / 32;
    extern __shared__ __align__(16) unsigned char smem[];
    uint2* wl = reinterpret_cast<uint2*>(smem);
    uint2* wl2 = wl + NT * KS * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < NT * KS * 64; i += kT) { wl[i] = wpack[i]; if (a2) wl2[i] = wpack2[i]; }
    __syncthreads();
    const size_t ntiles = (n + 31) / 32;
    const int p = lane & 31, h = lane >> 5;
    for (size_t tile = (size_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (size_t)gridDim.x * 4) {
        half4_t xf[KS], xg[KS];
        load_x<C>(a, tile, n, lane, xf);
        if (a2) load_x<C>(a2, tile, n, lane, xg);
        const size_t row = tile * 32 + p;
        if constexpr (C > 32) asm volatile("" ::: "memory");  // weight fragments stay in LDS (see k_mst_qkv)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const size_t off = row * (size_t)C + 32 * t + 16 * h;
            half8_t r0 = {0, 0, 0, 0, 0, 0, 0, 0}, r1 = r0;
            if (add && row < n) {
                r0 = __builtin_bit_cast(half8_t, reinterpret_cast<const uint4*>(add + off)[0]);
                r1 = __builtin_bit_cast(half8_t, reinterpret_cast<const uint4*>(add + off)[1]);
            }
            float16_t d;
#pragma unroll
            for (int v = 0; v < 16; ++v) d[v] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) d = mfma(__builtin_bit_cast(half4_t, wl[(t * KS + s) * 64 + lane]), xf[s], d);
            if (a2) {
#pragma unroll
                for (int s = 0; s < KS; ++s) d = mfma(__builtin_bit_cast(half4_t, wl2[(t * KS + s) * 64 + lane]), xg[s], d);
            }
            _Float16 o[16];
#pragma unroll
            for (int v = 0; v < 8; ++v) { o[v] = (_Float16)(d[v] + (float)r0[v]); o[8 + v] = (_Float16)(d[8 + v] + (float)r1[v]); }
            if (row < n) store_tile16(out + off, o);
        }
    }
}

template <int C>
int launch_rowgemm_add(avx_ctx* ctx, const void* a, const void* wpack, const void* a2, const void* wpack2, const void* add, void* out, size_t n, hipStream_t s) {
    constexpr int KS = C / 8, NT = C / 32;
    const size_t lds = (size_t)2 * NT * KS * 64 * sizeof(uint2);
    const size_t ntiles = (n + 31) / 32;
    long blocks = (long)((ntiles + 3) / 4);
    const long cap = (long)ctx->num_cus * (C == 128 ? 2 : 8);
    if (blocks > cap) blocks = cap;
    auto k = k_mst_rowgemm_add<C>;
    AVX_HIP(ctx, hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kT), lds, s, (const __half*)a, (const uint2*)wpack, (const __half*)a2, (const uint2*)wpack2,
                       (const __half*)add, (__half*)out, n);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

// ---- attention matrix -> fused projection weight, in fragment order -----------------------------------------------------
// MS_MSA :127-135 after the Gram pass: attn = softmax_j((k_i . q_j) / (|k_i| |q_j|) * rescale) per head (F.normalize's
// eps = 1e-12 on the norms; the padding column 31 takes no part), then M = blockdiag(attn_h^T) @ W_proj^T so that
// proj(attn @ v) == v @ M.  One small launch instead of a dozen elementwise ones; M leaves already packed for
// k_mst_rowgemm_add ("channels x pixels" fragments, see the header).
__global__ __launch_bounds__(256) void k_mst_attn_pack(const float* __restrict__ gram /*[HD][32][32]*/, const float* __restrict__ nq, const float* __restrict__ nk,
                                                       const float* __restrict__ rescale /*[HD]*/, const float* __restrict__ wpt /*[C][C] = W_proj^T*/, int C,
                                                       __half* __restrict__ mpack /*[C/32][C/8][64][4] | layout 1: [C/32][C/16][64][8]*/, int layout) {
    __shared__ float A[4][32][33];
    const int HD = C / 32, KS = C / 8, tid = threadIdx.x;
    for (int idx = tid; idx < HD * 1024; idx += 256) {
        const int hd = idx >> 10, i = (idx >> 5) & 31, j = idx & 31;
        A[hd][i][j] = gram[idx] / (fmaxf(nk[hd * 32 + i], 1e-12f) * fmaxf(nq[hd * 32 + j], 1e-12f)) * rescale[hd];
    }
    __syncthreads();
    if (tid < HD * 32) {
        float* row = A[tid >> 5][tid & 31];
        float m = row[0];
        for (int j = 1; j < 31; ++j) m = fmaxf(m, row[j]);
        float sum = 0.f;
        for (int j = 0; j < 31; ++j) { const float e = __expf(row[j] - m); row[j] = e; sum += e; }
        const float inv = 1.0f / sum;
        for (int j = 0; j < 31; ++j) row[j] *= inv;
        row[31] = 0.f;
    }
    __syncthreads();
    {  // every block repeats the (tiny) softmax above and then packs its own 256 entries of M
        const int idx = blockIdx.x * 256 + tid;
        int k = 0, n = 0;
        if (layout == 0) {  // v_mfma_f32_32x32x8_f16 fragments of k_mst_rowgemm_add
            const int e = idx & 3, lane = (idx >> 2) & 63, s = (idx >> 8) % KS, t = (idx >> 8) / KS;
            const int hh = lane >> 5, m = lane & 31;
            k = hh * (C / 2) + 4 * s + e;                                    // input channel (row of M)
            n = 32 * t + 16 * ((m & 7) >> 2) + 4 * (m >> 3) + (m & 3);       // output channel (column of M)
        } else if (layout == 1) {  // v_mfma_f32_32x32x16_f16 fragments of k_mst_attn_tail (csrc/mst_fused.hip): lane half hh carries input channels
                  // [hh C/2, (hh + 1) C/2) in steps of 8, and a result lane (pixel, hd) ends up with output channels hd C/2 + 16 t + v
            const int e = idx & 7, lane = (idx >> 3) & 63, s = (idx >> 9) % (C / 16), t = (idx >> 9) / (C / 16);
            const int hh = lane >> 5, m = lane & 31;
            k = hh * (C / 2) + 8 * s + e;
            n = ((m >> 2) & 1) * (C / 2) + 16 * t + (m & 3) + 4 * (m >> 3);
        }
        if (layout == 2) {  // block-diagonal v_mfma_f32_16x16x32_f16 fragments of k_mst_attn_tail_mx (csrc/mst_fused.hip): [C/8 octets][C/16 steps][64][8]; result row
                            // m = 8 s + c is output channel 8 o + c at tile row s, K slot q of step t2 carries row q & 1 and input channels 16 t2 + 8 (q >> 1) + e
            const int e = idx & 7, lane = (idx >> 3) & 63, t2 = (idx >> 9) % (C / 16), o = (idx >> 9) / (C / 16);
            const int m = lane & 15, q2 = lane >> 4;
            if ((q2 & 1) != (m >> 3)) { mpack[idx] = __float2half(0.f); return; }
            k = 16 * t2 + 8 * (q2 >> 1) + e;
            n = 8 * o + (m & 7);
        }
        const int hd = k >> 5, j = k & 31;
        float acc = 0.f;
        for (int i = 0; i < 32; ++i) acc = __builtin_fmaf(A[hd][i][j], wpt[(size_t)(hd * 32 + i) * C + n], acc);
        mpack[idx] = __float2half(acc);
    }
}

// ---- dense 3x3 conv (C -> C, zero padding 1) [+ add]: MST.embedding / mapping (:199, :228), conv_out (:277) ----------------
// Implicit GEMM with K = 9 taps x C: a wave owns 32 consecutive pixels of one image row; for every tap its x fragment is
// the (clamped, then zero-masked) neighbour row chunk, multiplied against that tap's pre-packed C x C weight.
template <int C>
__global__ __launch_bounds__(kT) void k_mst_conv3x3(const __half* __restrict__ x /*[B][H][W][C]*/, const uint2* __restrict__ wpack /*[9][C/32][KS][64]*/,
                                                    const __half* __restrict__ add /*or NULL*/, __half* __restrict__ out, int B, int H, int W) {
    constexpr int KS = C / 8, NT = C / 32;
    extern __shared__ __align__(16) unsigned char smem[];
    uint2* wl = reinterpret_cast<uint2*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 9 * NT * KS * 64; i += kT) wl[i] = wpack[i];
    __syncthreads();
    const int xt = (W + 31) / 32;
    const long total = (long)B * H * xt;
    const int p = lane & 31, h = lane >> 5;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < total; tile += (long)gridDim.x * 4) {
        const int x0 = (int)(tile % xt) * 32, y = (int)((tile / xt) % H);
        const long b = tile / ((long)xt * H);
        const int xw = x0 + p, xc = xw < W ? xw : W - 1;
        half4_t xf[9][KS];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = xc + tap % 3 - 1;
            const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy), xxc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            const uint4* src = reinterpret_cast<const uint4*>(x + ((b * H + yc) * (size_t)W + xxc) * C + h * (C / 2));
            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
#pragma unroll
            for (int q = 0; q < C / 16; ++q) {
                uint4 v = src[q];
                v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
                const half8_t h8 = __builtin_bit_cast(half8_t, v);
                xf[tap][2 * q] = half4_t{h8[0], h8[1], h8[2], h8[3]};
                xf[tap][2 * q + 1] = half4_t{h8[4], h8[5], h8[6], h8[7]};
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float16_t d;
#pragma unroll
            for (int v = 0; v < 16; ++v) d[v] = 0.f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int s = 0; s < KS; ++s) d = mfma(__builtin_bit_cast(half4_t, wl[((tap * NT + t) * KS + s) * 64 + lane]), xf[tap][s], d);
            if (xw < W) {
                const size_t off = ((b * H + y) * (size_t)W + xw) * C + 32 * t + 16 * h;
                half8_t r0 = {0, 0, 0, 0, 0, 0, 0, 0}, r1 = r0;
                if (add) {
                    r0 = __builtin_bit_cast(half8_t, reinterpret_cast<const uint4*>(add + off)[0]);
                    r1 = __builtin_bit_cast(half8_t, reinterpret_cast<const uint4*>(add + off)[1]);
                }
                _Float16 o[16];
#pragma unroll
                for (int v = 0; v < 8; ++v) { o[v] = (_Float16)(d[v] + (float)r0[v]); o[8 + v] = (_Float16)(d[8 + v] + (float)r1[v]); }
                store_tile16(out + off, o);
            }
        }
    }
}

// ---- pos_emb: depthwise 3x3 -> GELU -> depthwise 3x3 (+ x + bias) in one pass (MS_MSA :104-106, :136-137) ------------------
// A workgroup owns a TH x TW pixel tile: phase 1 evaluates the first conv + GELU on the (TH+2) x (TW+2) halo region into LDS
// (float16, zero outside the image: the second conv's zero padding applies to THIS map), phase 2 runs the second conv from
// LDS and adds the residual and the projection bias.  The intermediate map never reaches memory (2 of the 5 passes saved).
template <int C, int TH, int TW>
__global__ __launch_bounds__(kT) void k_mst_posemb(const __half* __restrict__ v, const float* __restrict__ w1 /*[C][9]*/, const float* __restrict__ w2 /*[C][9]*/,
                                                   const __half* __restrict__ res, const float* __restrict__ bias /*[C]*/, __half* __restrict__ out, int B, int H,
                                                   int W) {
    constexpr int CV = C / 8, PH = TH + 2, PW = TW + 2;
    static_assert(PH % 2 == 0 && kT % CV == 0, "row pairs in phase 1; fixed channel group per thread");
    extern __shared__ __align__(16) unsigned char smem[];
    __half* wl = reinterpret_cast<__half*>(smem);                                   // [2][9][C] taps, tap-major, float16 (the model's own values: the float32 table holds them exactly)
    float* bl = reinterpret_cast<float*>(wl + 2 * 9 * C);                           // [C] bias
    unsigned char* mid = reinterpret_cast<unsigned char*>(bl + C);                  // [PH][PW][C] float16
    const int tid = threadIdx.x;
    for (int i = tid; i < 9 * C; i += kT) { const int c = i / 9, t = i - 9 * c; wl[t * C + c] = __float2half(w1[i]); wl[9 * C + t * C + c] = __float2half(w2[i]); }
    for (int i = tid; i < C; i += kT) bl[i] = bias ? bias[i] : 0.f;
    const int tx = (W + TW - 1) / TW, ty = (H + TH - 1) / TH;
    const long total = (long)B * ty * tx;
    const int cg = tid % CV;  // kT % CV == 0: a thread keeps its channel group, so each phase holds its 72 taps (nine 16-byte registers of float16) for the launch
    auto taps = [&](const __half* wbase, uint4 (&wt)[9]) {
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const uint4*>(wbase + t * C + cg * 8);
    };
    // float16 data x float16 taps, float32 accumulation: one v_fma_mix_f32 per MAC, no conversion instructions (mst_common.h)
    auto mac8 = [&](float (&acc)[8], const uint4& hv, const uint4& wv) {
        fma_mix_lo(acc[0], hv.x, wv.x); fma_mix_hi(acc[1], hv.x, wv.x);
        fma_mix_lo(acc[2], hv.y, wv.y); fma_mix_hi(acc[3], hv.y, wv.y);
        fma_mix_lo(acc[4], hv.z, wv.z); fma_mix_hi(acc[5], hv.z, wv.z);
        fma_mix_lo(acc[6], hv.w, wv.w); fma_mix_hi(acc[7], hv.w, wv.w);
    };
    __syncthreads();
    for (long tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int x0 = (int)(tile % tx) * TW, y0 = (int)((tile / tx) % ty) * TH;
        const long b = tile / ((long)tx * ty);
        // every pixel phase 1 touches (rows y0 - 2 .. y0 + TH + 1, columns x0 - 2 .. x0 + TW + 1) inside the frame: no clamping, no masks
        const bool interior = y0 >= 2 && y0 + TH + 2 <= H && x0 >= 2 && x0 + TW + 2 <= W;
        __syncthreads();  // the previous tile's phase 2 is done with `mid`
        uint4 wt1[9];  // reloaded per tile and phase (nine LDS reads): resident for the launch, the two sets cost 72 registers and a wave of occupancy
        taps(wl, wt1);
#pragma unroll 1
        for (int it = tid; it < (PH / 2) * PW * CV; it += kT) {  // one item = two vertically adjacent pixels of the halo region
            const int pp = it / CV, c = pp % PW, r = 2 * (pp / PW);
            const int yy = y0 - 1 + r, xx = x0 - 1 + c;
            uint4 raw[4][3];
            if (interior) {
                const __half* base = v + ((b * H + yy - 1) * (size_t)W + xx - 1) * C + cg * 8;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) raw[rr][kx] = *reinterpret_cast<const uint4*>(base + ((size_t)rr * W + kx) * C);
            } else {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)  // branch-free: clamped addresses, out-of-image taps zeroed
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int y2 = yy - 1 + rr, x2 = xx + kx - 1;
                        const bool ok = y2 >= 0 && y2 < H && x2 >= 0 && x2 < W;
                        const int yc = y2 < 0 ? 0 : (y2 >= H ? H - 1 : y2), xc = x2 < 0 ? 0 : (x2 >= W ? W - 1 : x2);
                        uint4 rw = *reinterpret_cast<const uint4*>(v + ((b * H + yc) * (size_t)W + xc) * C + cg * 8);
                        rw.x = ok ? rw.x : 0u; rw.y = ok ? rw.y : 0u; rw.z = ok ? rw.z : 0u; rw.w = ok ? rw.w : 0u;
                        raw[rr][kx] = rw;
                    }
            }
            float acc[2][8];
#pragma unroll
            for (int o2 = 0; o2 < 2; ++o2)
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[o2][q] = 0.f;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int o2 = 0; o2 < 2; ++o2) {  // input row rr is tap row ky = rr - o2 of output o2
                        const int ky = rr - o2;
                        if (ky < 0 || ky > 2) continue;
                        mac8(acc[o2], raw[rr][kx], wt1[ky * 3 + kx]);
                    }
#pragma unroll
            for (int o2 = 0; o2 < 2; ++o2) {
                const int ym = yy + o2;
                const bool inside = interior || (ym >= 0 && ym < H && xx >= 0 && xx < W);
                unsigned o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float2_t g = gelu_fast2(float2_t{acc[o2][2 * q], acc[o2][2 * q + 1]});
                    o[q] = inside ? pack_f16(g.x, g.y) : 0u;
                }
                *reinterpret_cast<uint4*>(mid + (((size_t)(r + o2) * PW + c) * C + cg * 8) * 2) = uint4{o[0], o[1], o[2], o[3]};
            }
        }
        __syncthreads();
        uint4 wt2[9];
        taps(wl + 9 * C, wt2);
#pragma unroll 1
        for (int it = tid; it < TH * TW * CV; it += kT) {
            const int pp = it / CV, c = pp % TW, r = pp / TW;
            const int yo = y0 + r, xo = x0 + c;
            if (yo >= H || xo >= W) continue;
            float acc[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) mac8(acc, *reinterpret_cast<const uint4*>(mid + (((size_t)(r + t / 3) * PW + c + t % 3) * C + cg * 8) * 2), wt2[t]);
            const size_t off = ((b * H + yo) * (size_t)W + xo) * C + cg * 8;
            uint4 rr = uint4{0u, 0u, 0u, 0u};
            if (res) rr = *reinterpret_cast<const uint4*>(res + off);
            const unsigned rw[4] = {rr.x, rr.y, rr.z, rr.w};
            unsigned o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // (acc + residual) + bias, the float16 residual taken as it is (x * 1.0 + acc)
                float s0 = acc[2 * q], s1 = acc[2 * q + 1];
                fma_mix_lo(s0, rw[q], 0x3c003c00u); fma_mix_hi(s1, rw[q], 0x3c003c00u);
                o[q] = pack_f16(s0 + bl[cg * 8 + 2 * q], s1 + bl[cg * 8 + 2 * q + 1]);
            }
            *reinterpret_cast<uint4*>(out + off) = uint4{o[0], o[1], o[2], o[3]};
        }
    }
}

template <int C, int TH, int TW>
int launch_posemb(avx_ctx* ctx, const void* v, const float* w1, const float* w2, const void* res, const float* bias, void* out, int B, int H, int W, hipStream_t s) {
    const size_t lds = sizeof(__half) * 2 * 9 * C + sizeof(float) * C + (size_t)(TH + 2) * (TW + 2) * C * 2;
    const long total = (long)B * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    const long cap = (long)ctx->num_cus * (lds > 40 * 1024 ? 3 : 5);
    auto k = k_mst_posemb<C, TH, TW>;
    AVX_HIP(ctx, hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)(total < cap ? total : cap)), dim3(kT), lds, s, (const __half*)v, w1, w2, (const __half*)res, bias, (__half*)out, B, H, W);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

// ---- ConvTranspose2d(C -> C/2, kernel 2, stride 2) + bias: MST decoder upsampling (:214, :256) -------------------------
// Kernel size == stride: output pixel (2y + dy, 2x + dx) depends on input pixel (y, x) alone, through tap (dy, dx) -- four
// independent C x C/2 GEMMs whose results interleave in the output.  A wave takes 32 consecutive input pixels of a row and
// writes, per tap and output-channel tile, 32 contiguous bytes per lane at the scattered output pixel.
// With `skip` / `wskip` the decoder's 1x1 fusion conv over [up | skip] (:257) rides along: its `up` half is folded into the taps on
// the host (W_tap @ W_up^T, a C x C/2 matrix per tap; bias likewise) and its `skip` half is a second product on the output pixel's
// own skip row -- `up` and the concatenation never exist, and the level's input is read once instead of being written and re-read.
// GRAM (C = 64 -> 32 output channels): the Gram pass of the MSAB block that follows runs on the rounded output in registers, as in
// csrc/mst_fused.hip::k_mst_conv3x3_lds<., true> (a lane holds channels 16 h + v of its output pixel: the K = 16 operand order of k_mst_qkv16).
template <int C, bool GRAM = false>
__global__ __launch_bounds__(kT) void k_mst_convt2x2(const __half* __restrict__ x /*[B][H][W][C]*/, const uint2* __restrict__ wpack /*[4][C/64][C/8][64]*/,
                                                     const float* __restrict__ bias /*[C/2]*/, const __half* __restrict__ skip /*[B][2H][2W][C/2] or NULL*/,
                                                     const uint2* __restrict__ wskip /*[C/64][C/16][64]*/, __half* __restrict__ out /*[B][2H][2W][C/2]*/, int B,
                                                     int H, int W, const uint4* __restrict__ wqk = nullptr /*[2][2][64]*/, float* __restrict__ partial = nullptr /*[blocks][34][32]*/) {
    constexpr int KS = C / 8, CO = C / 2, NT = CO / 32, KSS = CO / 8;
    static_assert(!GRAM || C == 64, "the Gram epilogue is for the 32-channel output level");
    extern __shared__ __align__(16) unsigned char smem[];
    uint2* wl = reinterpret_cast<uint2*>(smem);
    float* bl = reinterpret_cast<float*>(smem + (size_t)4 * NT * KS * 64 * sizeof(uint2));
    uint2* wsl = reinterpret_cast<uint2*>(bl + CO);
    uint4* wq = reinterpret_cast<uint4*>(wsl + NT * KSS * 64);  // GRAM only
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4 * NT * KS * 64; i += kT) wl[i] = wpack[i];
    for (int i = tid; i < CO; i += kT) bl[i] = bias[i];
    if (skip)
        for (int i = tid; i < NT * KSS * 64; i += kT) wsl[i] = wskip[i];
    float16_t G;
    float nq = 0.f, nk = 0.f;
    if constexpr (GRAM) {
        for (int i = tid; i < 4 * 64; i += kT) wq[i] = wqk[i];
#pragma unroll
        for (int v = 0; v < 16; ++v) G[v] = 0.f;
    }
    __syncthreads();
    const int xt = (W + 31) / 32;
    const long total = (long)B * H * xt;
    const int p = lane & 31, h = lane >> 5;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < total; tile += (long)gridDim.x * 4) {
        const int x0 = (int)(tile % xt) * 32, y = (int)((tile / xt) % H);
        const long b = tile / ((long)xt * H);
        const int xw = x0 + p;
        half4_t xf[KS];
        {
            const size_t row = (size_t)((b * H + y) * (long)W + (xw < W ? xw : W - 1));
            const uint4* src = reinterpret_cast<const uint4*>(x + row * C + h * (C / 2));
#pragma unroll
            for (int q = 0; q < C / 16; ++q) {
                const half8_t h8 = __builtin_bit_cast(half8_t, src[q]);
                xf[2 * q] = half4_t{h8[0], h8[1], h8[2], h8[3]};
                xf[2 * q + 1] = half4_t{h8[4], h8[5], h8[6], h8[7]};
            }
        }
        if constexpr (C > 64) asm volatile("" ::: "memory");  // weight fragments stay in LDS
#pragma unroll
        for (int tap = 0; tap < 4; ++tap) {
            const size_t opix = ((size_t)(b * 2 * H + 2 * y + (tap >> 1)) * (size_t)(2 * W) + (size_t)(2 * (xw < W ? xw : W - 1) + (tap & 1)));
            half4_t sf[KSS];
            if (skip) {  // lane (p, h): channels [h CO/2, (h + 1) CO/2) of the OUTPUT pixel's skip row
                const uint4* src = reinterpret_cast<const uint4*>(skip + opix * CO + h * (CO / 2));
#pragma unroll
                for (int q = 0; q < CO / 16; ++q) {
                    const half8_t h8 = __builtin_bit_cast(half8_t, src[q]);
                    sf[2 * q] = half4_t{h8[0], h8[1], h8[2], h8[3]};
                    sf[2 * q + 1] = half4_t{h8[4], h8[5], h8[6], h8[7]};
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float16_t d;
#pragma unroll
                for (int v = 0; v < 16; ++v) d[v] = 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s) d = mfma(__builtin_bit_cast(half4_t, wl[((tap * NT + t) * KS + s) * 64 + lane]), xf[s], d);
                if (skip) {
#pragma unroll
                    for (int s = 0; s < KSS; ++s) d = mfma(__builtin_bit_cast(half4_t, wsl[(t * KSS + s) * 64 + lane]), sf[s], d);
                }
                _Float16 o[16];
#pragma unroll
                for (int v = 0; v < 16; ++v) o[v] = (_Float16)(d[v] + bl[32 * t + 16 * h + v]);
                if (xw < W) store_tile16(out + opix * CO + 32 * t + 16 * h, o);
                if constexpr (GRAM) {
                    half8_t o0, o1;
#pragma unroll
                    for (int v = 0; v < 8; ++v) { o0[v] = xw < W ? o[v] : (_Float16)0.f; o1[v] = xw < W ? o[8 + v] : (_Float16)0.f; }  // columns past the edge take no part
                    float16_t dq, dk;
#pragma unroll
                    for (int v = 0; v < 16; ++v) { dq[v] = 0.f; dk[v] = 0.f; }
                    dq = mfma16(o0, __builtin_bit_cast(half8_t, wq[0 * 64 + lane]), dq);  // D[pixel][q channel]
                    dq = mfma16(o1, __builtin_bit_cast(half8_t, wq[1 * 64 + lane]), dq);
                    dk = mfma16(o0, __builtin_bit_cast(half8_t, wq[2 * 64 + lane]), dk);  // D[pixel][k channel]
                    dk = mfma16(o1, __builtin_bit_cast(half8_t, wq[3 * 64 + lane]), dk);
                    unsigned aq[8], ak[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        aq[j] = pack_f16(dq[2 * j], dq[2 * j + 1]);
                        ak[j] = pack_f16(dk[2 * j], dk[2 * j + 1]);
                        fma_mix_lo(nq, aq[j], aq[j]); fma_mix_hi(nq, aq[j], aq[j]);
                        fma_mix_lo(nk, ak[j], ak[j]); fma_mix_hi(nk, ak[j], ak[j]);
                    }
#pragma unroll
                    for (int b2 = 0; b2 < 2; ++b2)
                        G = mfma16(__builtin_bit_cast(half8_t, uint4{ak[4 * b2], ak[4 * b2 + 1], ak[4 * b2 + 2], ak[4 * b2 + 3]}),
                                   __builtin_bit_cast(half8_t, uint4{aq[4 * b2], aq[4 * b2 + 1], aq[4 * b2 + 2], aq[4 * b2 + 3]}), G);
                }
            }
        }
    }
    if constexpr (GRAM) {  // this workgroup's partial: [34][32] = 32 Gram rows, sum q^2, sum k^2 (as k_mst_qkv16); the weights in LDS are dead
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
        float* mine = red + (size_t)wave * 34 * 32;
#pragma unroll
        for (int v = 0; v < 16; ++v) mine[(8 * (v / 4) + 4 * h + (v % 4)) * 32 + p] = G[v];
        const float sq = nq + __shfl_xor(nq, 32), sk = nk + __shfl_xor(nk, 32);
        if (h == 0) { mine[32 * 32 + p] = sq; mine[33 * 32 + p] = sk; }
        __syncthreads();
        for (int i = tid; i < 34 * 32; i += kT) {
            float sum = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) sum += red[(size_t)w4 * 34 * 32 + i];
            partial[(size_t)blockIdx.x * 34 * 32 + i] = sum;
        }
    }
}

// ---- the same pass on v_mfma_f32_32x32x16_f16 (gfx950's double-rate form) ------------------------------------------------------------------
// k_mst_qkv above is BOUND by its K = 8 MFMAs (64 cycles each: 144 per 32 pixels at C = 128 is 70 us of matrix-core time per 4K launch at a
// sixteenth of the pixels, 77 us at C = 64, 93 at C = 32); the K = 16 form does the same products in a quarter of the cycles and leaves
// the pass to its one read of x.  Fragment order (ml/mst_plus_plus.py::pack_qkv16): step q, lane half h, element j <-> input channel
// h C/2 + 8 q + j -- again the order in which a lane holds its contiguous half row, one 16-byte load per step.  A "pixels x channels"
// accumulator (lane = channel, registers v <-> pixels 8 (v / 4) + 4 h + v % 4) packs into TWO K = 16 operands of the Gram MFMA (registers
// 0-7 and 8-15: the K order is free and k and q use the same one).  Norms: sum of squares of the float16-rounded values, taken from the
// packed pair by v_fma_mix_f32 (no conversion back).  NW waves per workgroup: 8 at C = 128, where the 98 KB of weights in LDS allow one
// workgroup per CU and four waves left every SIMD alone with its latencies.
template <int C, int NW>
__global__ __launch_bounds__(64 * NW) void k_mst_qkv16(const __half* __restrict__ x, const uint4* __restrict__ wpack /*[3*HD][C/16][64]*/, size_t n,
                                                      __half* __restrict__ v_out /*[n][C] or NULL*/, float* __restrict__ partial /*[blocks][HD][34][32]*/) {
    constexpr int KS = C / 16, HD = C / 32, NT = 3 * HD;
    extern __shared__ __align__(16) unsigned char smem[];
    uint4* wl = reinterpret_cast<uint4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < NT * KS * 64; i += 64 * NW) wl[i] = wpack[i];
    __syncthreads();
    auto W = [&](int t, int s) { return __builtin_bit_cast(half8_t, wl[(t * KS + s) * 64 + lane]); };
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
    auto load_row = [&](size_t tile, uint4 (&u)[KS]) {  // rows past the end: the last row's bytes, zeroed when they are used
        const size_t row = tile * 32 + p;
        const uint4* src = reinterpret_cast<const uint4*>(x + (row < n ? row : n - 1) * (size_t)C + h * (C / 2));
#pragma unroll
        for (int q = 0; q < KS; ++q) u[q] = src[q];
    };
    constexpr bool PF = C <= 64;  // next tile's row in flight under this tile's MFMAs (C = 128: the registers go to the four Gram accumulators)
    uint4 xn[KS];
    size_t tile = (size_t)blockIdx.x * NW + wave;
    if (PF && tile < ntiles) load_row(tile, xn);
    for (; tile < ntiles; tile += (size_t)gridDim.x * NW) {
        const size_t row = tile * 32 + p;
        half8_t xf[KS];
        if (!PF) load_row(tile, xn);
#pragma unroll
        for (int q = 0; q < KS; ++q) xf[q] = __builtin_bit_cast(half8_t, row < n ? xn[q] : uint4{0u, 0u, 0u, 0u});
        if (PF && tile + (size_t)gridDim.x * NW < ntiles) load_row(tile + (size_t)gridDim.x * NW, xn);
        if constexpr (C > 32) asm volatile("" ::: "memory");  // keep the weight fragments in LDS: hoisted out of this loop they cost 6 C VGPRs
#pragma unroll
        for (int hd = 0; hd < HD; ++hd) {
            float16_t dq, dk;
#pragma unroll
            for (int v = 0; v < 16; ++v) { dq[v] = 0.f; dk[v] = 0.f; }
#pragma unroll
            for (int s = 0; s < KS; ++s) dq = mfma16(xf[s], W(hd, s), dq);        // D[pixel][q channel]
#pragma unroll
            for (int s = 0; s < KS; ++s) dk = mfma16(xf[s], W(HD + hd, s), dk);   // D[pixel][k channel]
            unsigned aq[8], ak[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                aq[j] = pack_f16(dq[2 * j], dq[2 * j + 1]);
                ak[j] = pack_f16(dk[2 * j], dk[2 * j + 1]);
                fma_mix_lo(nq[hd], aq[j], aq[j]); fma_mix_hi(nq[hd], aq[j], aq[j]);
                fma_mix_lo(nk[hd], ak[j], ak[j]); fma_mix_hi(nk[hd], ak[j], ak[j]);
            }
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2)  // G[i][j] += sum_px k[px][i] q[px][j], 16 pixels per instruction
                G[hd] = mfma16(__builtin_bit_cast(half8_t, uint4{ak[4 * b2], ak[4 * b2 + 1], ak[4 * b2 + 2], ak[4 * b2 + 3]}),
                               __builtin_bit_cast(half8_t, uint4{aq[4 * b2], aq[4 * b2 + 1], aq[4 * b2 + 2], aq[4 * b2 + 3]}), G[hd]);
        }
        if (v_out)  // NULL: the attention tail forms v from x itself (avx_mst_attn_tail_x): this pass only reads
#pragma unroll
        for (int hd = 0; hd < HD; ++hd) {
            float16_t dv;
#pragma unroll
            for (int v = 0; v < 16; ++v) dv[v] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) dv = mfma16(W(2 * HD + hd, s), xf[s], dv);  // D[v channel (permuted)][pixel]
            if (row < n) {
                unsigned o[8];
#pragma unroll
                for (int v = 0; v < 8; ++v) o[v] = pack_f16(dv[2 * v], dv[2 * v + 1]);
                uint4* dst = reinterpret_cast<uint4*>(v_out + row * (size_t)C + 32 * hd + 16 * h);
                dst[0] = uint4{o[0], o[1], o[2], o[3]};
                dst[1] = uint4{o[4], o[5], o[6], o[7]};
            }
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
    for (int i = tid; i < HD * 34 * 32; i += 64 * NW) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += red[(size_t)w * HD * 34 * 32 + i];
        partial[(size_t)blockIdx.x * HD * 34 * 32 + i] = s;
    }
}

template <int C, int NW>
int launch_qkv16(avx_ctx* ctx, avx_ws* ws, const void* x, const void* wpack, size_t n, void* v_out, float* gram, float* nq, float* nk, hipStream_t s) {
    constexpr int KS = C / 16, HD = C / 32;
    const size_t wbytes = (size_t)3 * HD * KS * 64 * sizeof(uint4), rbytes = sizeof(float) * NW * HD * 34 * 32;
    const size_t lds = wbytes > rbytes ? wbytes : rbytes;
    const size_t ntiles = (n + 31) / 32;
    long blocks = (long)((ntiles + NW - 1) / NW);
    long per_cu = (long)(160 * 1024 / lds);
    if (per_cu > 16 / NW * 2) per_cu = 16 / NW * 2;  // at most 8 waves per SIMD's worth of workgroups
    if (per_cu < 1) per_cu = 1;
    const long cap = (long)ctx->num_cus * per_cu;
    if (blocks > cap) blocks = cap;
    int rc = avx_ensure_scratch(ctx, ws, sizeof(float) * (size_t)blocks * HD * 34 * 32);
    if (rc) return rc;
    float* partial = (float*)ws->d_scratch;
    auto k = k_mst_qkv16<C, NW>;
    AVX_HIP(ctx, hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64 * NW), lds, s, (const __half*)x, (const uint4*)wpack, n, (__half*)v_out, partial);
    AVX_HIP(ctx, hipGetLastError());
    const int tot = HD * 34 * 32;
    hipLaunchKernelGGL(k_mst_qkv_final, dim3((tot / 4 * 64 + 255) / 256), dim3(256), 0, s, partial, (int)blocks, HD, gram, nq, nk);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
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
    hipLaunchKernelGGL(k_mst_qkv_final, dim3((tot / 4 * 64 + 255) / 256), dim3(256), 0, s, partial, (int)blocks, HD, gram, nq, nk);
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

template <int C, int ROWS, int PXT, bool PF, int MINW>
int launch_ffn2(avx_ctx* ctx, const void* hid, const float* w9, const void* w2pack, const void* res, void* out, int B, int H, int W, hipStream_t s) {
    constexpr int C4 = 4 * C, KS2 = C4 / 8, NT = C / 32, PITCH = C4 * 2 + 16;
    const size_t lds = (size_t)NT * KS2 * 64 * sizeof(uint2) + sizeof(float) * 9 * C4 + (size_t)ROWS * 32 * PXT * PITCH;
    const long total = (long)B * ((H + ROWS - 1) / ROWS) * ((W + 32 * PXT - 1) / (32 * PXT));
    const long cap = (long)ctx->num_cus * MINW;  // resident workgroups per CU (registers / LDS)
    auto k = k_mst_ffn2<C, ROWS, PXT, PF, MINW>;
    AVX_HIP(ctx, hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)(total < cap ? total : cap)), dim3(kT), lds, s, (const __half*)hid, w9, (const uint2*)w2pack, (const __half*)res,
                       (__half*)out, B, H, W);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

}  // namespace

int avx_mst_qkv_final_launch(avx_ctx* ctx, const float* partial, int blocks, int heads, float* gram, float* nq, float* nk, hipStream_t s) {
    const int tot = heads * 34 * 32;
    hipLaunchKernelGGL(k_mst_qkv_final, dim3((tot / 4 * 64 + 255) / 256), dim3(256), 0, s, partial, blocks, heads, gram, nq, nk);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

extern "C" {

int avx_mst_qkv_gram(avx_ctx* ctx, const void* x, const void* wpack, size_t n_pix, int C, void* v_out, float* gram, float* nq, float* nk, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, x && wpack && gram && nq && nk && n_pix > 0, "avx_mst_qkv_gram: NULL pointer or empty tensor");  // v_out may be NULL
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

int avx_mst_qkv_gram16(avx_ctx* ctx, const void* x, const void* wpack16, size_t n_pix, int C, void* v_out, float* gram, float* nq, float* nk, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, x && wpack16 && gram && nq && nk && n_pix > 0, "avx_mst_qkv_gram16: NULL pointer or empty tensor");  // v_out may be NULL
    AVX_REQUIRE(ctx, C == 32 || C == 64 || C == 128, "avx_mst_qkv_gram16: C=%d (32, 64 or 128: 31-channel groups stored 32 wide)", C);
    AVX_REQUIRE(ctx, ((((uintptr_t)x | (uintptr_t)wpack16 | (uintptr_t)v_out)) & 15u) == 0, "avx_mst_qkv_gram16: pointers must be 16-byte aligned");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    avx_ws* ws = avx_workspace(ctx, s);
    if (!ws) return AVX_ERR_NOMEM;
    if (C == 32) return launch_qkv16<32, 4>(ctx, ws, x, wpack16, n_pix, v_out, gram, nq, nk, s);
    if (C == 64) return launch_qkv16<64, 4>(ctx, ws, x, wpack16, n_pix, v_out, gram, nq, nk, s);
    return launch_qkv16<128, 8>(ctx, ws, x, wpack16, n_pix, v_out, gram, nq, nk, s);
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

int avx_mst_dw_gemm_add(avx_ctx* ctx, const void* hidden, const float* w_c9, const void* w2pack, const void* residual, void* out, int B, int H, int W,
                        int C, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, hidden && w_c9 && w2pack && residual && out && B > 0 && H > 0 && W > 0, "avx_mst_dw_gemm_add: NULL pointer or empty tensor");
    AVX_REQUIRE(ctx, C == 32 || C == 64, "avx_mst_dw_gemm_add: C=%d (32 or 64)", C);
    AVX_REQUIRE(ctx, ((((uintptr_t)hidden | (uintptr_t)w2pack | (uintptr_t)residual | (uintptr_t)out)) & 15u) == 0,
                "avx_mst_dw_gemm_add: pointers must be 16-byte aligned");
    AVX_REQUIRE(ctx, hidden != out, "avx_mst_dw_gemm_add: the hidden tensor cannot be the output (the residual can)");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    if (C == 32) return launch_ffn2<32, 2, 2, false, 3>(ctx, hidden, w_c9, w2pack, residual, out, B, H, W, s);
    return launch_ffn2<64, 2, 1, true, 2>(ctx, hidden, w_c9, w2pack, residual, out, B, H, W, s);
}

int avx_mst_rowgemm_add(avx_ctx* ctx, const void* a, const void* wpack, const void* a2, const void* wpack2, const void* add, void* out, size_t rows, int C,
                        void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, a && wpack && out && rows > 0 && (!a2 == !wpack2), "avx_mst_rowgemm_add: NULL pointer or empty tensor");
    AVX_REQUIRE(ctx, C == 32 || C == 64 || C == 128, "avx_mst_rowgemm_add: C=%d (32, 64 or 128)", C);
    AVX_REQUIRE(ctx, ((((uintptr_t)a | (uintptr_t)wpack | (uintptr_t)a2 | (uintptr_t)wpack2 | (uintptr_t)add | (uintptr_t)out)) & 15u) == 0,
                "avx_mst_rowgemm_add: pointers must be 16-byte aligned");
    AVX_REQUIRE(ctx, a != out && a2 != out, "avx_mst_rowgemm_add: a GEMM operand cannot be the output (the addend can)");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    if (C == 32) return launch_rowgemm_add<32>(ctx, a, wpack, a2, wpack2, add, out, rows, s);
    if (C == 64) return launch_rowgemm_add<64>(ctx, a, wpack, a2, wpack2, add, out, rows, s);
    return launch_rowgemm_add<128>(ctx, a, wpack, a2, wpack2, add, out, rows, s);
}

static int attn_pack_impl(avx_ctx* ctx, const float* gram, const float* nq, const float* nk, const float* rescale, const float* wproj_t, int C, void* mpack,
                          int layout, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, gram && nq && nk && rescale && wproj_t && mpack, "avx_mst_attn_pack: NULL pointer");
    AVX_REQUIRE(ctx, C == 32 || C == 64 || C == 128, "avx_mst_attn_pack: C=%d (32, 64 or 128)", C);
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    hipLaunchKernelGGL(k_mst_attn_pack, dim3((unsigned)((layout == 2 ? 4 : 1) * C * C / 256)), dim3(256), 0, s, gram, nq, nk, rescale, wproj_t, C, (__half*)mpack, layout);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

int avx_mst_attn_pack(avx_ctx* ctx, const float* gram, const float* nq, const float* nk, const float* rescale, const float* wproj_t, int C, void* mpack,
                      void* stream) {
    return attn_pack_impl(ctx, gram, nq, nk, rescale, wproj_t, C, mpack, 0, stream);
}

int avx_mst_attn_pack16(avx_ctx* ctx, const float* gram, const float* nq, const float* nk, const float* rescale, const float* wproj_t, int C, void* mpack,
                        void* stream) {
    return attn_pack_impl(ctx, gram, nq, nk, rescale, wproj_t, C, mpack, 1, stream);
}

int avx_mst_attn_pack_mx(avx_ctx* ctx, const float* gram, const float* nq, const float* nk, const float* rescale, const float* wproj_t, int C, void* mpack,
                         void* stream) {
    return attn_pack_impl(ctx, gram, nq, nk, rescale, wproj_t, C, mpack, 2, stream);
}

int avx_mst_conv3x3_add(avx_ctx* ctx, const void* x, const void* wpack, const void* add, void* out, int B, int H, int W, int C, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, x && wpack && out && B > 0 && H > 0 && W > 0, "avx_mst_conv3x3_add: NULL pointer or empty tensor");
    AVX_REQUIRE(ctx, C == 32, "avx_mst_conv3x3_add: C=%d (32: the 31-channel full-resolution convs)", C);
    AVX_REQUIRE(ctx, ((((uintptr_t)x | (uintptr_t)wpack | (uintptr_t)add | (uintptr_t)out)) & 15u) == 0, "avx_mst_conv3x3_add: pointers must be 16-byte aligned");
    AVX_REQUIRE(ctx, x != out, "avx_mst_conv3x3_add: in-place convolution is not possible");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    constexpr int KS = 4, NT = 1;
    const size_t lds = (size_t)9 * NT * KS * 64 * sizeof(uint2);
    const long total = (long)B * H * ((W + 31) / 32);
    long blocks = (total + 3) / 4;
    const long cap = (long)ctx->num_cus * 8;
    if (blocks > cap) blocks = cap;
    auto k = k_mst_conv3x3<32>;
    AVX_HIP(ctx, hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kT), lds, s, (const __half*)x, (const uint2*)wpack, (const __half*)add, (__half*)out, B, H, W);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

int avx_mst_posemb(avx_ctx* ctx, const void* v, const float* w1_c9, const float* w2_c9, const void* residual, const float* bias, void* out, int B, int H, int W,
                   int C, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, v && w1_c9 && w2_c9 && out && B > 0 && H > 0 && W > 0, "avx_mst_posemb: NULL pointer or empty tensor");
    AVX_REQUIRE(ctx, C == 32 || C == 64 || C == 128, "avx_mst_posemb: C=%d (32, 64 or 128)", C);
    AVX_REQUIRE(ctx, ((((uintptr_t)v | (uintptr_t)residual | (uintptr_t)out)) & 15u) == 0, "avx_mst_posemb: pointers must be 16-byte aligned");
    AVX_REQUIRE(ctx, v != out, "avx_mst_posemb: in-place convolution is not possible");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    if (C == 32) return launch_posemb<32, 8, 32>(ctx, v, w1_c9, w2_c9, residual, bias, out, B, H, W, s);
    if (C == 64) return launch_posemb<64, 8, 16>(ctx, v, w1_c9, w2_c9, residual, bias, out, B, H, W, s);
    return launch_posemb<128, 4, 16>(ctx, v, w1_c9, w2_c9, residual, bias, out, B, H, W, s);
}

static int convt2x2_impl(avx_ctx* ctx, const void* x, const void* wpack, const float* bias, const void* skip, const void* wskip, void* out, int B, int H, int W,
                         int C, void* stream);
int avx_mst_convt2x2(avx_ctx* ctx, const void* x, const void* wpack, const float* bias, void* out, int B, int H, int W, int C, void* stream) {
    return convt2x2_impl(ctx, x, wpack, bias, nullptr, nullptr, out, B, H, W, C, stream);
}
int avx_mst_convt2x2_fuse(avx_ctx* ctx, const void* x, const void* wpack, const float* bias, const void* skip, const void* wskip, void* out, int B, int H,
                          int W, int C, void* stream) {
    if (ctx && !(skip && wskip)) return avx_fail(ctx, AVX_ERR_INVALID, "avx_mst_convt2x2_fuse: skip / wskip is NULL");
    if (ctx && ((((uintptr_t)skip | (uintptr_t)wskip)) & 15u)) return avx_fail(ctx, AVX_ERR_INVALID, "avx_mst_convt2x2_fuse: pointers must be 16-byte aligned");
    return convt2x2_impl(ctx, x, wpack, bias, skip, wskip, out, B, H, W, C, stream);
}
// avx_mst_convt2x2_fuse at C = 64 (32 output channels) followed by avx_mst_qkv_gram16 (v_out = NULL) on its output, in one pass; one frame
int avx_mst_convt2x2_fuse_gram(avx_ctx* ctx, const void* x, const void* wpack, const float* bias, const void* skip, const void* wskip, void* out, int H, int W, int C,
                               const void* wqk16, float* gram, float* nq, float* nk, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, x && wpack && bias && skip && wskip && out && wqk16 && gram && nq && nk && H > 0 && W > 0, "avx_mst_convt2x2_fuse_gram: NULL pointer or empty tensor");
    AVX_REQUIRE(ctx, C == 64, "avx_mst_convt2x2_fuse_gram: C=%d (64 input channels: the decoder step back to full resolution)", C);
    AVX_REQUIRE(ctx, ((((uintptr_t)x | (uintptr_t)wpack | (uintptr_t)out | (uintptr_t)skip | (uintptr_t)wskip | (uintptr_t)wqk16)) & 15u) == 0,
                "avx_mst_convt2x2_fuse_gram: pointers must be 16-byte aligned");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    avx_ws* ws = avx_workspace(ctx, s);
    if (!ws) return AVX_ERR_NOMEM;
    const long total = (long)H * ((W + 31) / 32);
    long blocks = (total + 3) / 4;
    const long cap = (long)ctx->num_cus * 4;
    if (blocks > cap) blocks = cap;
    int rc = avx_ensure_scratch(ctx, ws, sizeof(float) * (size_t)blocks * 34 * 32);
    if (rc) return rc;
    float* partial = (float*)ws->d_scratch;
    size_t lds = (size_t)4 * 1 * 8 * 64 * sizeof(uint2) + sizeof(float) * 32 + (size_t)1 * 4 * 64 * sizeof(uint2) + (size_t)4 * 64 * sizeof(uint4);
    if (lds < sizeof(float) * 4 * 34 * 32) lds = sizeof(float) * 4 * 34 * 32;  // the four waves' results land where the weights were
    auto k = k_mst_convt2x2<64, true>;
    AVX_HIP(ctx, hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kT), lds, s, (const __half*)x, (const uint2*)wpack, bias, (const __half*)skip, (const uint2*)wskip, (__half*)out, 1, H, W,
                       (const uint4*)wqk16, partial);
    AVX_HIP(ctx, hipGetLastError());
    return avx_mst_qkv_final_launch(ctx, partial, (int)blocks, 1, gram, nq, nk, s);
}
static int convt2x2_impl(avx_ctx* ctx, const void* x, const void* wpack, const float* bias, const void* skip, const void* wskip, void* out, int B, int H, int W,
                         int C, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, x && wpack && bias && out && B > 0 && H > 0 && W > 0, "avx_mst_convt2x2: NULL pointer or empty tensor");
    AVX_REQUIRE(ctx, C == 64 || C == 128, "avx_mst_convt2x2: C=%d (64 or 128 input channels)", C);
    AVX_REQUIRE(ctx, ((((uintptr_t)x | (uintptr_t)wpack | (uintptr_t)out)) & 15u) == 0, "avx_mst_convt2x2: pointers must be 16-byte aligned");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    const long total = (long)B * H * ((W + 31) / 32);
    long blocks = (total + 3) / 4;
    const long cap = (long)ctx->num_cus * (C == 128 ? 2 : 8);
    if (blocks > cap) blocks = cap;
    if (C == 64) {
        const size_t lds = (size_t)4 * 1 * 8 * 64 * sizeof(uint2) + sizeof(float) * 32 + (size_t)1 * 4 * 64 * sizeof(uint2);
        AVX_HIP(ctx, hipFuncSetAttribute((const void*)k_mst_convt2x2<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_mst_convt2x2<64, false>), dim3((unsigned)blocks), dim3(kT), lds, s, (const __half*)x, (const uint2*)wpack, bias, (const __half*)skip,
                           (const uint2*)wskip, (__half*)out, B, H, W);
    } else {
        const size_t lds = (size_t)4 * 2 * 16 * 64 * sizeof(uint2) + sizeof(float) * 64 + (size_t)2 * 8 * 64 * sizeof(uint2);
        AVX_HIP(ctx, hipFuncSetAttribute((const void*)k_mst_convt2x2<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_mst_convt2x2<128, false>), dim3((unsigned)blocks), dim3(kT), lds, s, (const __half*)x, (const uint2*)wpack, bias, (const __half*)skip,
                           (const uint2*)wskip, (__half*)out, B, H, W);
    }
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

}  // extern "C"
