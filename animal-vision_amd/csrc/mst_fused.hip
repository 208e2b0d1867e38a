// csrc/mst_fused.hip -- MSAB block kernels fused end to end (reference: ml/MST_plus_plus/predict_code/architecture/
// MST_Plus_Plus.py): the activations of a block make ONE HBM round trip per half instead of one per operator.
//
//   avx_mst_ffn_fused   PreNorm + FeedForward + residual (:57-65, :141-158, :184):
//                       LayerNorm -> 1x1 conv (C -> 4C) -> GELU -> depthwise 3x3 -> GELU -> 1x1 conv (4C -> C) -> + x
//                       in one kernel.  The 4C-channel hidden tensor -- 512 of the block's 1,216 B/px at C = 32 -- never
//                       reaches memory: a workgroup owns a 16 x 16 pixel tile and keeps the hidden map of its 18 x 18 halo
//                       region in LDS (the 1-pixel halo is recomputed: 27 % more first-GEMM + GELU work, no HBM traffic).
//
// Schedule (512 threads = 8 waves, one workgroup per CU, persistent over tiles; v_mfma_f32_32x32x16_f16):
//   phase 0  LayerNorm of the 324 halo pixels (lane pair = one pixel, float32 statistics over the 31 real channels of each
//            32-wide group), normalised rows to LDS as float16.  The raw rows were fetched into registers BEFORE the
//            previous tile's phase 2, so their HBM latency is off the critical path.
//   phase 1  first GEMM + GELU: wave w owns hidden-channel tile w % 4 (its W1 fragments stay in registers) and walks the
//            pixel groups w / 4, w / 4 + 2, ...; "channels x pixels" results (a lane ends up with 16 contiguous hidden
//            channels of one pixel: two 16-byte LDS stores); pixels outside the image store zeros (the depthwise conv's
//            zero padding applies to THIS map).
//   phase 2  depthwise 3x3 + GELU + second GEMM + residual: wave w owns output rows 2w, 2w + 1 (32 pixels).  Per K-step a
//            lane (pixel p, half h) convolves hidden channels 8 (2 s + h) ... + 7 of its pixel from nine 16-byte LDS reads
//            (taps as float16 pairs, float32 accumulation: v_fma_mix_f32) -- and those 8 values ARE its B-operand fragment
//            of the second GEMM's MFMA (k = 8 h + j), so the GELU'd map never touches LDS either.
// Hidden channels are processed in passes of 128 (C = 32: one pass; C = 64: two, the second GEMM accumulating across them).
// Rounding: what reaches LDS or memory is float16 (as in the autocast pipeline of predict_torch.py:109); LayerNorm statistics and
// all accumulations are float32, and a GELU takes its argument straight from the float32 accumulator (the reference rounds the
// conv / GEMM result to float16 first: one rounding less here, and two conversions per element less in kernels that are bound
// by exactly this arithmetic; the forward pass is held to the reference's float32 output, tests/test_mstpp.py).
#include <hip/hip_fp16.h>

#include <cstdio>
#include <cstdlib>

#include "avx_internal.h"
#include "mst_common.h"

namespace {

typedef float float4_t __attribute__((ext_vector_type(4)));
#ifndef AVX_FFN32_GM2
#define AVX_FFN32_GM2 1
#endif
constexpr int kFT = 512;              // 8 waves
constexpr int TS = 16;                // output tile side
constexpr int HS = TS + 2;            // halo tile side
#ifndef AVX_FFN_TH
#define AVX_FFN_TH 16
#endif
constexpr int kFfnTH = AVX_FFN_TH;     // rows of a FeedForward tile.  -DAVX_FFN_TH=14 (the 18 x 16 halo region is then exactly nine 32-pixel groups instead of ten and four pixels: 6.5 % fewer LayerNorm / first-GEMM / GELU slots per output pixel) measured 1.8 % SLOWER at 4K: 15 % more tiles, each with its barriers, prologue and epilogue (profiles/r03/ab_ffn_tile_16x14.txt)
static_assert(kFfnTH == 14 || kFfnTH == 16, "FeedForward tile rows");
constexpr int kFfnNGRP = (HS * (kFfnTH + 2) + 31) / 32;
// Hidden channels are processed in passes of HPASS (128 or 64).  Per pixel the hidden tile takes HPASS * 2 + 16 bytes (68 or 36
// dwords: 16-byte reads of 16 consecutive pixels are conflict-free) and a halo ROW is padded to a multiple of 256 bytes, so the
// two output rows of a wave hit disjoint banks.  HPASS = 64 halves the tile: two workgroups fit a CU (4 waves per SIMD), and
// one's barriers and LDS waits are covered by the other's arithmetic.

__device__ __forceinline__ float16_t mfma16(half8_t a, half8_t b, float16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

struct Tile { int x0, y0; long b; };
// Tiles blockIdx.x, + gridDim.x, ... of a (frames x ty x tx) grid WITHOUT a division per tile: the coordinates advance by the step's own
// (dx, dy, db) with carries -- scalar adds and compares.  tile_of(t) = (t % tx, (t / tx) % ty, t / (tx ty)) as 64-bit divisions cost ~150
// vector instructions per tile and wave (there is no scalar divide): a fifth of the attention tail's instruction stream.
struct TileWalk {
    int xi, yi, dx, dy, tx, ty;
    long b, db;
    __device__ __forceinline__ void init(long t0, long step, int tx_, int ty_) {
        tx = tx_; ty = ty_;
        auto uni = [](long v) { return (long)(unsigned)__builtin_amdgcn_readfirstlane((int)v) | (long)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32; };  // the divisions run on the vector unit
        xi = (int)uni(t0 % tx); yi = (int)uni((t0 / tx) % ty); b = uni(t0 / ((long)tx * ty));
        dx = (int)uni(step % tx); dy = (int)uni((step / tx) % ty); db = uni(step / ((long)tx * ty));
    }
    __device__ __forceinline__ void advance() {
        xi += dx;
        const int cx = xi >= tx ? 1 : 0;
        xi -= cx ? tx : 0;
        yi += dy + cx;
        const int cy = yi >= ty ? 1 : 0;
        yi -= cy ? ty : 0;
        b += db + cy;
    }
    __device__ __forceinline__ Tile tile(int side) const { return Tile{xi * side, yi * side, b}; }
};

// DWM (round 3): the depthwise 3x3 conv runs on the MATRIX pipe.  A depthwise conv contracts over taps only, so as a matrix product its weight
// operand is diagonal in the channels; the waste is bounded by letting the 16 rows of a v_mfma_f32_16x16x32_f16 result be 8 channels (one
// "octet": what a lane's 16-byte LDS read holds) x 2 vertically adjacent output rows: the two rows share input rows, so the 3 x 3 taps of both
// are covered by 4 input rows x 3 column shifts = 12 K-slots of 8 channels = THREE MFMAs per 8 channels x 2 rows x 16 pixels (256 outputs;
// 36 v_fma_mix_f32 per lane before).  Slot q = lane / 16 of MFMA i carries input row r = q, column shift i; the weight fragment (dwpack, built
// on the host: ml/mst_plus_plus.py::pack_dw_mfma) of lane (m = 8 s + c, q) holds w[c][r - s][i] at element c when 0 <= r - s <= 2, zero elsewhere.
// Phase 2 becomes 2a + 2b: (2a) wave w owns octet w of the pass (8 w + 64 o2 ... for 128-channel passes two of them), its three weight
// fragments stay in registers, and it walks the 8 row pairs of the tile: 3 LDS reads, 3 MFMAs, GELU of the lane's 4 results, one 8-byte LDS
// store -- IN PLACE: output rows 2p, 2p + 1 land in halo rows 2p, 2p + 1 (columns 0 ... 15), which no later row pair of this octet reads and
// no other wave touches (its bytes are another octet's); (2b) after a barrier the second GEMM reads its B fragments from that map with the
// same 16-byte reads the depthwise conv used to issue.  One ADDRESS register serves all of 2a (row pair, column shift and octet are
// immediates), the taps need no LDS, and the vector unit is left with the GELUs.
// NW: waves per workgroup.  8 everywhere in round 2; 16 (one 1024-thread workgroup per CU, 128 registers per lane) for the 62-channel blocks since round 3: their
// 143 KB of LDS admit one workgroup per CU, and at 8 waves the vector unit saw 2 waves per SIMD -- half of what it needs to issue back to back.  With 16 waves a pass's
// 16 octets map one to a wave in phase 2a, the 44 phase 1 items are 3 + 3 + 3 + 2 per channel tile, and in phase 2b a wave owns a row pair x ONE 32-channel output tile.
template <int C, int HPASS, int MINW, bool DWM, bool STAMP = false, int NW = 8>
__global__ __launch_bounds__(64 * NW, MINW) void k_mst_ffn_fused(const __half* __restrict__ x /*[B][H][W][C]*/, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, const uint4* __restrict__ w1pack /*[4C/32][C/16][64]*/, const __half* __restrict__ taps /*[9][4C]*/,
                                                          const uint4* __restrict__ w2pack /*[C/32][4C/16][64]*/, __half* __restrict__ out /*[B][H][W][C], != x*/, int B,
                                                          int H, int W, const uint4* __restrict__ dwpack /*[4C/8][3][64], DWM only*/, unsigned long long* __restrict__ stamps /*STAMP: [blocks][8 waves][8 segments] cycles*/) {
    // tile = TS (16) columns x FTH rows (16; 14 is the measured-and-rejected alternative, see kFfnTH)
    constexpr int FTH = kFfnTH, HSY = FTH + 2, NHALO = HS * HSY, NGRP = (NHALO + 31) / 32;
    constexpr int HID = 4 * C, NPASS = HID / HPASS, KS1 = C / 16, KS2 = HPASS / 16, NT = C / 32, YPITCH = C * 2 + 16, LNV = C / 16;
    constexpr int HPITCH = HPASS * 2 + 16, RPITCH = (HS * HPITCH + 255) / 256 * 256 /* a halo ROW: HS pixels */, NCT = HPASS / 32;  // channel tiles of 32 per pass
    static_assert(C == 32 || C == 64 || C == 128, "31-, 62- or 124-channel blocks (stored 32 / 64 / 128 wide)");
    constexpr int NTHR = 64 * NW, NGW = (NGRP + NW - 1) / NW;  // threads; halo pixel groups per wave
    constexpr int NTW = NT / (NW / 8), NOCTW = HPASS / (8 * NW);  // 32-channel output tiles per wave (phase 2b / epilogue); octets per wave and pass (phase 2a)
    static_assert((NW == 8 || NW == 16) && NT % (NW / 8) == 0 && (!DWM || HPASS % (8 * NW) == 0) && (DWM || NW == 8), "wave split");
    constexpr bool PREFETCH = C <= 64;  // the next tile's raw rows wait in registers during phase 2 (C = 128: 64 registers that the accumulators need)
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* yt = smem;                                       // [NGRP * 32][YPITCH]  LayerNorm'd rows, float16
    unsigned char* ht = yt + (size_t)NGRP * 32 * YPITCH;             // [HSY rows][RPITCH]: [HS px][HPITCH]  hidden map of the halo region, float16
    __half* tapl = reinterpret_cast<__half*>(ht + (size_t)HSY * RPITCH);  // [9][HID] (not with DWM: no table, no space)
    float* gl = reinterpret_cast<float*>(tapl + (DWM ? 0 : 9 * HID));  // [C] gamma, [C] beta
    constexpr bool W2LDS = NW == 16;  // 128 registers per lane: the second GEMM's fragments of a pass wait in LDS (16 KB), not in 32 registers through phase 2a
    uint4* w2l = reinterpret_cast<uint4*>(gl + 2 * C);               // W2LDS: [NT][KS2][64] this pass's W2 fragments
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 31, h = lane >> 5;
    // STAMP (diagnostic instantiation, AVX_FFN_STAMPS=1): cycles this wave spends per segment -- 0 top barrier, 1 phase 1, 2 barrier, 3 phase 2a (or the
    // whole vector-unit phase 2), 4 barrier, 5 phase 2b, 6 epilogue, 7 LayerNorm + fetch
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    auto stamp = [&](int k) {
        if constexpr (STAMP) {
            const unsigned long long now = __builtin_readcyclecounter();
            seg[k] += now - tlast;
            tlast = now;
        }
    };
    if constexpr (STAMP) tlast = __builtin_readcyclecounter();
    if constexpr (!DWM)
        for (int i = tid; i < 9 * HID; i += NTHR) tapl[i] = taps[i];
    for (int i = tid; i < C; i += NTHR) { gl[i] = gamma[i]; gl[C + i] = beta[i]; }
    const int tx = (W + TS - 1) / TS, ty = (H + FTH - 1) / FTH;
    const long total = (long)B * ty * tx;
    auto tile_of = [&](long t) { return Tile{(int)(t % tx) * TS, (int)((t / tx) % ty) * FTH, t / ((long)tx * ty)}; };  // (TileWalk measured 2 % slower in THIS kernel: the divisions run on the scalar unit beside a saturated vector unit)
    const float cnt = (float)(NT * 31);

    // ---- phase 0, split: fetch the raw rows of this wave's halo pixel groups (wave w: groups w and w + 8) ----
    uint4 raw[NGW][LNV];
    unsigned f_rel[NGW];  // this lane's halo pixel (group wave + 8 gi), as a byte offset from the halo origin (y0 - 1, x0 - 1): scalar base + lane offset when the halo lies inside the frame
#pragma unroll
    for (int gi = 0; gi < NGW; ++gi) {
        const int q = 32 * (wave + NW * gi) + p, qq = q < NHALO ? q : NHALO - 1;
        f_rel[gi] = (unsigned)(((qq / HS) * W + qq % HS) * C + h * (C / 2)) * 2u;
    }
    auto fetch = [&](const Tile& t) {
        if (t.y0 >= 1 && t.y0 + HSY - 1 <= H && t.x0 >= 1 && t.x0 + HS - 1 <= W) {  // all but the frame's border tiles: no clamping, no 64-bit lane arithmetic
            const char* origin = reinterpret_cast<const char*>(x + ((t.b * H + t.y0 - 1) * (size_t)W + t.x0 - 1) * C);
#pragma unroll
            for (int gi = 0; gi < NGW; ++gi) {
                if (wave + NW * gi >= NGRP) break;
                unsigned o = f_rel[gi];
                asm volatile("" : "+v"(o));  // keeps the zero-extension here (hoisted out of the tile loop it becomes a 64-bit lane address)
                const uint4* src = reinterpret_cast<const uint4*>(origin + o);
#pragma unroll
                for (int v = 0; v < LNV; ++v) raw[gi][v] = src[v];
            }
            return;
        }
#pragma unroll
        for (int gi = 0; gi < NGW; ++gi) {
            const int g = wave + NW * gi;
            if (g >= NGRP) break;
            const int q = 32 * g + p, qq = q < NHALO ? q : NHALO - 1;
            int yy = t.y0 - 1 + qq / HS, xx = t.x0 - 1 + qq % HS;
            yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
            xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            const uint4* src = reinterpret_cast<const uint4*>(x + ((t.b * H + yy) * (size_t)W + xx) * C + h * (C / 2));
#pragma unroll
            for (int v = 0; v < LNV; ++v) raw[gi][v] = src[v];
        }
    };
    auto layernorm = [&]() {  // lane (p, h) holds channels [h * C/2, (h + 1) * C/2) of halo pixel 32 g + p
        if constexpr (DWM) {
            // Folded form (round 3): gamma and beta live in the first GEMM's weights -- W1' = W1 diag(gamma), and W1 beta rides in row 31 of W1' (the first group's padding
            // channel), whose operand element is set to 1 here -- so what is left per element is the centring, the square and one scaling: 3.5 instead of 5.5 vector
            // instructions and no table reads.  The sum runs over all 32 slots of a group with v_dot2_f32_f16 (the padding channels of x are zero in the model).
#pragma unroll
            for (int gi = 0; gi < NGW; ++gi) {
                const int g = wave + NW * gi;
                if (g >= NGRP) break;
                typedef _Float16 half2v __attribute__((ext_vector_type(2)));
                constexpr unsigned kOnes = 0x3c003c00u;
                auto is_pad = [&](int i) { return ((h * (C / 2) + i) & 31) == 31; };
                auto word = [&](int i) { const uint4 u = raw[gi][i / 8]; const int k = (i / 2) & 3; return k == 0 ? u.x : k == 1 ? u.y : k == 2 ? u.z : u.w; };
                float s1 = 0.f;
#pragma unroll
                for (int i = 0; i < 8 * LNV; i += 2) s1 = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, word(i)), __builtin_bit_cast(half2v, kOnes), s1, false);
                const float mean = (s1 + __shfl_xor(s1, 32)) / cnt, nmean = -mean;
                float d[8 * LNV], s2 = 0.f;
#pragma unroll
                for (int i = 0; i < 8 * LNV; ++i) {
                    const float t = (i & 1) ? fma_mix_hi_c(word(i), kOnes, nmean) : fma_mix_lo_c(word(i), kOnes, nmean);
                    d[i] = is_pad(i) ? 0.f : t;
                    s2 = __builtin_fmaf(d[i], d[i], s2);
                }
                const float rstd = rsqrtf((s2 + __shfl_xor(s2, 32)) / cnt + eps);
                unsigned char* dst = yt + (size_t)(32 * g + p) * YPITCH + h * C;
#pragma unroll
                for (int v = 0; v < LNV; ++v) {
                    unsigned o[4];
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const int i = 8 * v + j;
                        const float2_t y2 = float2_t{d[i], d[i + 1]} * float2_t{rstd, rstd};
                        const float bq = (h * (C / 2) + i + 1) == 31 ? 1.0f : y2.y;  // channel 31: the bias slot (the other groups' padding channels stay 0: d = 0 there)
                        o[j / 2] = pack_f16(y2.x, bq);
                    }
                    *reinterpret_cast<uint4*>(dst + 16 * v) = uint4{o[0], o[1], o[2], o[3]};
                }
            }
            return;
        }
#pragma unroll
        for (int gi = 0; gi < NGW; ++gi) {
            const int g = wave + NW * gi;
            if (g >= NGRP) break;
            // float16 values enter through v_fma_mix_f32 (x * 1.0 + c): the sum and the centring cost one instruction per element, no conversions
            constexpr unsigned kOnes = 0x3c003c00u;
            auto is_pad = [&](int i) { return ((h * (C / 2) + i) & 31) == 31; };
            auto word = [&](int i) { const uint4 u = raw[gi][i / 8]; const int k = (i / 2) & 3; return k == 0 ? u.x : k == 1 ? u.y : k == 2 ? u.z : u.w; };
            float s1 = 0.f;
#pragma unroll
            for (int i = 0; i < 8 * LNV; ++i) {
                const float t = (i & 1) ? fma_mix_hi_c(word(i), kOnes, s1) : fma_mix_lo_c(word(i), kOnes, s1);
                s1 = is_pad(i) ? s1 : t;
            }
            const float mean = (s1 + __shfl_xor(s1, 32)) / cnt, nmean = -mean;
            float d[8 * LNV], s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 8 * LNV; ++i) {
                const float t = (i & 1) ? fma_mix_hi_c(word(i), kOnes, nmean) : fma_mix_lo_c(word(i), kOnes, nmean);
                d[i] = is_pad(i) ? 0.f : t;
                s2 = __builtin_fmaf(d[i], d[i], s2);
            }
            const float rstd = rsqrtf((s2 + __shfl_xor(s2, 32)) / cnt + eps);
            unsigned char* dst = yt + (size_t)(32 * g + p) * YPITCH + h * C;  // h * (C/2) channels * 2 bytes
#pragma unroll
            for (int v = 0; v < LNV; ++v) {
                unsigned o[4];
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const int i = 8 * v + j, c = h * (C / 2) + i;
                    const float a = is_pad(i) ? 0.f : __builtin_fmaf(d[i] * rstd, gl[c], gl[C + c]);
                    const float bq = is_pad(i + 1) ? 0.f : __builtin_fmaf(d[i + 1] * rstd, gl[c + 1], gl[C + c + 1]);
                    o[j / 2] = pack_f16(a, bq);
                }
                *reinterpret_cast<uint4*>(dst + 16 * v) = uint4{o[0], o[1], o[2], o[3]};
            }
        }
    };

    const int n0 = (wave / 8) * NTW, w8 = wave % 8;  // this wave's first output tile and its row pair (phase 2b, epilogue)
    half8_t w2f[W2LDS ? 1 : NTW * KS2];  // the second GEMM's A fragments of one pass (this wave's output tiles): resident for the whole launch when there is one pass
    auto load_w2 = [&](int pass) {
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int s = 0; s < KS2; ++s) w2f[n * KS2 + s] = __builtin_bit_cast(half8_t, w2pack[((size_t)(n0 + n) * (HID / 16) + pass * KS2 + s) * 64 + lane]);
    };
    if constexpr (NPASS == 1 && !W2LDS) load_w2(0);
    // the first GEMM's A fragments of this wave's channel tile (wave % NCT of the pass): requested at the END of the previous pass's phase 1, so
    // the next phase 1 never waits for them (they were loaded at its top before: an L2 round trip per pass in front of the MFMAs)
    half8_t w1f[KS1];
    auto load_w1 = [&](int pass) {
#pragma unroll
        for (int s = 0; s < KS1; ++s) w1f[s] = __builtin_bit_cast(half8_t, w1pack[((size_t)(pass * NCT + wave % NCT) * KS1 + s) * 64 + lane]);
    };
    constexpr bool W1AHEAD = DWM && C == 64 && NPASS > 1 && NW == 8;  // C = 32 / 128: the fragments' registers are needed in between (they would spill): loaded at the top of phase 1
    if constexpr (W1AHEAD) load_w1(0);
    // phase 1 items of this wave (pixel groups wave / NCT + j * 8 / NCT): where a lane's pixel sits in the hidden tile and in the image
    constexpr int NITEM = (NGRP + NW / NCT - 1) / (NW / NCT);
    int p1_g[NITEM], p1_dst[NITEM], p1_dy[NITEM], p1_dx[NITEM];
#pragma unroll
    for (int j = 0; j < NITEM; ++j) {
        const int g = wave / NCT + j * (NW / NCT), q = 32 * g + p, qc = q < NHALO ? q : 0;
        p1_g[j] = g < NGRP ? g : -1;
        p1_dst[j] = (g < NGRP && q < NHALO) ? (qc / HS) * RPITCH + (qc % HS) * HPITCH + 32 * h : -1;
        p1_dy[j] = qc / HS - 1; p1_dx[j] = qc % HS - 1;
    }
    long tile = blockIdx.x;
    if (tile >= total) return;
    Tile t = tile_of(tile);
    fetch(t);
    __syncthreads();  // tables are in LDS
    layernorm();
    for (;;) {
        const long next = tile + gridDim.x;
        const bool halo_inside = t.y0 >= 1 && t.y0 + HSY - 1 <= H && t.x0 >= 1 && t.x0 + HS - 1 <= W;  // scalar
        float16_t D[NTW];
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int v = 0; v < 16; ++v) D[n][v] = 0.f;
#pragma unroll 1
        for (int pass = 0; pass < NPASS; ++pass) {
            stamp(pass == 0 ? 7 : 5);
            __syncthreads();  // yt complete (pass 0) / ht no longer read by the previous pass's phase 2
            stamp(0);
            if constexpr (W2LDS)  // every wave is past the previous pass's phase 2b: its fragments may be replaced
                for (int i = tid; i < NT * KS2 * 64; i += NTHR) w2l[i] = w2pack[((size_t)(i / (KS2 * 64)) * (HID / 16) + pass * KS2 + (i / 64) % KS2) * 64 + (i & 63)];
            half8_t af[DWM ? NOCTW : 1][3];  // DWM: this wave's depthwise weight fragments of the pass (in flight during phase 1)
            if constexpr (DWM) {
#pragma unroll
                for (int o2 = 0; o2 < NOCTW; ++o2)
#pragma unroll
                    for (int i = 0; i < 3; ++i) af[o2][i] = __builtin_bit_cast(half8_t, dwpack[((size_t)(pass * (HPASS / 8) + NW * o2 + wave) * 3 + i) * 64 + lane]);
            }
            // ---- phase 1: hidden = GELU(W1 y) for hidden channels [HPASS pass + 32 ct, + 32), ct = wave % NCT, pixel groups wave / NCT, + 8 / NCT, ... ----
            {
                const int ct = wave % NCT;
                if constexpr (!W1AHEAD) load_w1(pass);
#pragma unroll
                for (int j = 0; j < NITEM; ++j) {
                    if (p1_g[j] < 0) continue;  // wave-uniform
                    float16_t d;
#pragma unroll
                    for (int v = 0; v < 16; ++v) d[v] = 0.f;
                    const unsigned char* ysrc = yt + (size_t)(32 * p1_g[j] + p) * YPITCH + 16 * h;
#pragma unroll
                    for (int s = 0; s < KS1; ++s) d = mfma16(w1f[s], __builtin_bit_cast(half8_t, *reinterpret_cast<const uint4*>(ysrc + 32 * s)), d);
                    if (p1_dst[j] >= 0) {
                        uint32_t keep = 0xffffffffu;
                        if (!halo_inside) {  // border tiles: zeros outside the image
                            const int yy = t.y0 + p1_dy[j], xx = t.x0 + p1_dx[j];
                            keep = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? 0xffffffffu : 0u;
                        }
                        // GELU of the 16 results, GP1 pairs at a time step-major (gelu_multi): as many as the registers allow without spilling
                        constexpr int GP1 = (C == 64 && NW == 8) ? 4 : (C == 32 ? 2 : 1);  // C = 32 / 128: no register to spare (more than one pair in flight spills)
                        unsigned pk[8];
#pragma unroll
                        for (int v0 = 0; v0 < 8; v0 += GP1) {
                            float2_t gq[GP1];
#pragma unroll
                            for (int v = 0; v < GP1; ++v) gq[v] = float2_t{d[2 * (v0 + v)], d[2 * (v0 + v) + 1]};
                            if constexpr (DWM) gelu_mx<GP1>(gq); else gelu_multi<GP1>(gq);
#pragma unroll
                            for (int v = 0; v < GP1; ++v) pk[v0 + v] = pack_f16(gq[v].x, gq[v].y);
                        }
                        uint4 u0 = uint4{pk[0], pk[1], pk[2], pk[3]}, u1 = uint4{pk[4], pk[5], pk[6], pk[7]};
                        if (!halo_inside) { u0.x &= keep; u0.y &= keep; u0.z &= keep; u0.w &= keep; u1.x &= keep; u1.y &= keep; u1.z &= keep; u1.w &= keep; }
                        unsigned char* dst = ht + p1_dst[j] + 64 * ct;
                        reinterpret_cast<uint4*>(dst)[0] = u0;
                        reinterpret_cast<uint4*>(dst)[1] = u1;
                    }
                }
            }
            stamp(1);
            __syncthreads();  // ht complete
            stamp(2);
            if (PREFETCH && pass == NPASS - 1 && next < total) fetch(tile_of(next));  // next tile's raw rows: in flight during phase 2
            if constexpr (NPASS > 1 && !W2LDS) load_w2(pass);
            if constexpr (DWM) {
                // ---- phase 2a: depthwise 3x3 on the matrix pipe + GELU, in place in the hidden tile (see the kernel's head comment) ----
                constexpr int NOCT = NOCTW;  // octets of this pass per wave
                const int n16 = lane & 15, q = lane >> 4;
                unsigned char* rbase = ht + (size_t)q * RPITCH + (size_t)n16 * HPITCH + 16 * wave;                               // + 2p RPITCH + i HPITCH + 128 o2
                unsigned char* wbase = ht + (size_t)(q >> 1) * RPITCH + (size_t)n16 * HPITCH + 16 * wave + 8 * (q & 1);        // + 2p RPITCH + 128 o2
                // The map is read and written in place, so the compiler keeps every LDS read behind the previous unit's store (it cannot tell the bytes apart):
                // the reads of unit u + 1 are therefore issued by hand right behind unit u's MFMAs, in front of its GELU and store -- one LDS round trip per unit less on the wave's clock.
                // Software pipeline, one unit deep (the wave issues in order, and the compiler keeps every LDS read behind the previous store to the same array):
                //   unit u's three MFMAs | the reads of unit u + 1 (into the operand registers the MFMAs have just consumed) | GELU + store of unit u - 1
                // so an LDS round trip and the matrix pipe's latency both pass under the previous unit's GELU.  sched_barrier pins that order (left alone, the
                // scheduler hoists unit u + 1's first MFMA up against its reads and the wave waits out the LDS latency once per unit).
                constexpr int NU = NOCT * (FTH / 2);
                uint4 bq[3];
                auto rd = [&](int u) {
                    const int o2 = u / (FTH / 2), pr = u % (FTH / 2);
#pragma unroll
                    for (int i = 0; i < 3; ++i) bq[i] = *reinterpret_cast<const uint4*>(rbase + (size_t)(2 * pr) * RPITCH + (size_t)i * HPITCH + 16 * NW * o2);
                };
                auto finish = [&](int u, float4_t a) {
                    const int o2 = u / (FTH / 2), pr = u % (FTH / 2);
                    float2_t gp[2] = {float2_t{a[0], a[1]}, float2_t{a[2], a[3]}};
                    if constexpr (C == 64 || (C == 32 && AVX_FFN32_GM2)) gelu_mx<2>(gp);
                    else { float2_t g0[1] = {gp[0]}, g1[1] = {gp[1]}; gelu_mx<1>(g0); gelu_mx<1>(g1); gp[0] = g0[0]; gp[1] = g1[0]; }
                    *reinterpret_cast<uint2*>(wbase + (size_t)(2 * pr) * RPITCH + 16 * NW * o2) = uint2{pack_f16(gp[0].x, gp[0].y), pack_f16(gp[1].x, gp[1].y)};
                };
#ifndef AVX_FFN_PIPE_ALL
#define AVX_FFN_PIPE_ALL 0
#endif
                constexpr bool PIPE = C == 64 || AVX_FFN_PIPE_ALL;  // C = 32 / 128: not a register to spare (the lagging accumulator spills); measured, same-box A/B
                if constexpr (PIPE) {
                    rd(0);
                    float4_t prev = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const int o2 = u / (FTH / 2);
                        float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int i = 0; i < 3; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[o2][i], __builtin_bit_cast(half8_t, bq[i]), acc, 0, 0, 0);
                        if (u + 1 < NU) rd(u + 1);
                        __builtin_amdgcn_sched_barrier(0);
                        if (u > 0) finish(u - 1, prev);
                        __builtin_amdgcn_sched_barrier(0);
                        prev = acc;
                    }
                    finish(NU - 1, prev);
                } else {
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const int o2 = u / (FTH / 2);
                        rd(u);
                        float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int i = 0; i < 3; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[o2][i], __builtin_bit_cast(half8_t, bq[i]), acc, 0, 0, 0);
                        finish(u, acc);
                    }
                }
                stamp(3);
                __syncthreads();  // the GELU'd depthwise map is complete
                stamp(4);
                if constexpr (W1AHEAD) load_w1(pass + 1 < NPASS ? pass + 1 : 0);  // dead since phase 1; requested here (not there: 8 live registers through phase 2a spill) -- phase 2b and the barrier cover the trip
                // ---- phase 2b: second GEMM, this wave's 32 output pixels (rows 2 wave, 2 wave + 1), B fragments from the map ----
                const int r = 2 * w8 + (p >> 4), c = p & 15;
                const unsigned char* zb = ht + (size_t)r * RPITCH + (size_t)c * HPITCH + 16 * h;
                if (2 * w8 < FTH) {  // wave-uniform (FTH = 14: the eighth row pair does not exist)
#pragma unroll
                for (int s = 0; s < KS2; ++s) {
                    const half8_t bf = __builtin_bit_cast(half8_t, *reinterpret_cast<const uint4*>(zb + 32 * s));
#pragma unroll
                    for (int n = 0; n < NTW; ++n) {
                        if constexpr (W2LDS) D[n] = mfma16(__builtin_bit_cast(half8_t, w2l[((n0 + n) * KS2 + s) * 64 + lane]), bf, D[n]);
                        else D[n] = mfma16(w2f[n * KS2 + s], bf, D[n]);
                    }
                }
                }
            } else
            // ---- phase 2: depthwise 3x3 + GELU + W2, this wave's 32 output pixels (rows 2 wave, 2 wave + 1) ----
            if (2 * wave < FTH) {  // wave-uniform (FTH = 14: the eighth row pair does not exist)
                const int r = 2 * wave + (p >> 4), c = p & 15;
                const unsigned char* hbase = ht + (size_t)r * RPITCH + (size_t)c * HPITCH + 16 * h;  // top-left tap of this pixel, this lane's channel octet
                const unsigned char* tbase = reinterpret_cast<const unsigned char*>(tapl) + (size_t)(pass * HPASS + 8 * h) * 2;
#pragma unroll
                for (int s = 0; s < KS2; ++s) {
                    float acc[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const uint4 hv = *reinterpret_cast<const uint4*>(hbase + (size_t)(tap / 3) * RPITCH + (size_t)(tap % 3) * HPITCH + 32 * s);
                        const uint4 wv = *reinterpret_cast<const uint4*>(tbase + (size_t)tap * HID * 2 + 32 * s);
                        fma_mix_lo(acc[0], hv.x, wv.x); fma_mix_hi(acc[1], hv.x, wv.x);
                        fma_mix_lo(acc[2], hv.y, wv.y); fma_mix_hi(acc[3], hv.y, wv.y);
                        fma_mix_lo(acc[4], hv.z, wv.z); fma_mix_hi(acc[5], hv.z, wv.z);
                        fma_mix_lo(acc[6], hv.w, wv.w); fma_mix_hi(acc[7], hv.w, wv.w);
                    }
                    half8_t bf;
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const float2_t gv = gelu_fast2(float2_t{acc[j], acc[j + 1]});
                        bf[j] = (_Float16)gv.x;
                        bf[j + 1] = (_Float16)gv.y;
                    }
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        D[n] = mfma16(w2f[n * KS2 + s], bf, D[n]);
                }
            }
        }
        stamp(5);
        // ---- epilogue: + x (the block's residual, :184), float16, 32 contiguous bytes per lane and output tile ----
        {
            const int yo = t.y0 + 2 * w8 + (p >> 4), xo = t.x0 + (p & 15);
            if (2 * w8 < FTH && yo < H && xo < W) {
                const size_t off = ((t.b * H + yo) * (size_t)W + xo) * C + 16 * h + 32 * n0;
#pragma unroll
                for (int n = 0; n < NTW; ++n) {
                    const uint4 xa = reinterpret_cast<const uint4*>(x + off + 32 * n)[0], xb = reinterpret_cast<const uint4*>(x + off + 32 * n)[1];
                    const unsigned xw[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
                    unsigned ow[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {  // D + x with the float16 residual taken as it is (v_fma_mix_f32, x * 1.0 + D: the same sum, no conversion instruction)
                        float s0 = D[n][2 * j], s1 = D[n][2 * j + 1];
                        fma_mix_lo(s0, xw[j], 0x3c003c00u); fma_mix_hi(s1, xw[j], 0x3c003c00u);
                        ow[j] = pack_f16(s0, s1);
                    }
                    reinterpret_cast<uint4*>(out + off + 32 * n)[0] = uint4{ow[0], ow[1], ow[2], ow[3]};
                    reinterpret_cast<uint4*>(out + off + 32 * n)[1] = uint4{ow[4], ow[5], ow[6], ow[7]};
                }
            }
        }
        stamp(6);
        if (next >= total) break;
        tile = next;
        t = tile_of(tile);
        if (!PREFETCH) fetch(t);
        layernorm();  // yt is free: every wave is past the last pass's first barrier, after which nobody reads it
    }
    if constexpr (STAMP) {
        if (lane == 0 && stamps)
#pragma unroll
            for (int k = 0; k < 8; ++k) stamps[((size_t)blockIdx.x * NW + wave) * 8 + k] = seg[k];
    }
}


// ---- attention tail: pos_emb(v) + v @ M + bias + x in one pass (MS_MSA :104-106, :132-137; MSAB :183) ------------------------
// After the Gram pass the first half of a block is out = proj(attn @ v) + pos_emb(v) + x with proj(attn @ v) == v @ M + bias
// (M = blockdiag(attn^T) W_proj^T, one C x C matrix per frame, avx_mst_attn_pack16) and pos_emb = depthwise 3x3 -> GELU ->
// depthwise 3x3.  One workgroup per 16 x 16 pixel tile: v on the 20 x 20 halo region goes to LDS once (zeros outside the image),
// the GELU'd first conv on 18 x 18 stays in LDS, the second conv, the C x C product (MFMA, B operand = the v row already in LDS),
// bias and residual meet in registers: v and x are read once, out is written once (192 B/px at C = 32 where the pos_emb kernel
// plus the projection GEMM moved 384).  A lane (pixel p, half h) owns channels [h C/2, (h + 1) C/2) of its pixel in BOTH the
// second conv and the MFMA result (the K order and the output-row order of the M fragments are chosen for that).
constexpr int VS = TS + 4, MS = TS + 2;

template <int C, int MINW>
__global__ __launch_bounds__(kFT, MINW) void k_mst_attn_tail(const __half* __restrict__ v /*[B][H][W][C]*/, const __half* __restrict__ x /*[B][H][W][C]*/,
                                                             const uint4* __restrict__ mpack /*[C/32][C/16][64]*/, const __half* __restrict__ taps1 /*[9][C]*/,
                                                             const __half* __restrict__ taps2 /*[9][C]*/, const float* __restrict__ bias /*[C]*/,
                                                             __half* __restrict__ out, int B, int H, int W, const uint4* __restrict__ wvpack /*[C/32][C/16][64] or NULL*/) {
    constexpr int NO = C / 8, NS = C / 16, NT = C / 32, PP = C * 2 + 16;
    constexpr int VRP = (VS * PP + 255) / 256 * 256, MRP = VRP;  // one row pitch for both maps (the second region is sized for the x tile anyway): a phase B item's source and destination offsets coincide
    constexpr int NFILL = (VS * VS * NO + kFT - 1) / kFT;
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* vt = smem;                                  // [VS][VRP]: [VS px][PP]  v on the halo region
    unsigned char* mt = vt + (size_t)VS * VRP;                 // [MS][MRP]: [MS px][PP]  gelu(dw1(v)); with wvpack: first x on the halo region, [VS][VRP]
    // wvpack != NULL: v is not read but formed here, v = float16(x W_v^T) on the halo region (to_v has no bias, so the zero rows outside
    // the image stay zero: the convs' padding).  The staged tile is then x; it lands in the SECOND region (sized for it), the product goes
    // to vt, and once every wave is past that the region is mt's.  The Gram pass then writes no v at all: 128 B/px less per block.
    const bool fromx = wvpack != nullptr;
    __half* t2l = reinterpret_cast<__half*>(mt + (size_t)VS * VRP);  // [9][C] second conv's taps (behind the larger of the two uses of the region)
    __half* t1l = t2l + 9 * C;                                 // [9][C] first conv's taps
    float* bl = reinterpret_cast<float*>(t1l + 9 * C);         // [C]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 31, h = lane >> 5;
    for (int i = tid; i < 9 * C; i += kFT) { t2l[i] = taps2[i]; t1l[i] = taps1[i]; }
    for (int i = tid; i < C; i += kFT) bl[i] = bias ? bias[i] : 0.f;
    const int tx = (W + TS - 1) / TS, ty = (H + TS - 1) / TS;
    const long total = (long)B * ty * tx;
    // phase B role: a wave keeps ONE channel octet (its first-conv taps are wave-uniform LDS reads: broadcasts) and walks pixels of the 18 x 18 map
    const int oct = wave % NO, part = wave / NO, nparts = 8 / NO, per = (MS * MS + nparts - 1) / nparts;
    half8_t mf[NT * NS];
#pragma unroll
    for (int i = 0; i < NT * NS; ++i) mf[i] = __builtin_bit_cast(half8_t, mpack[(size_t)i * 64 + lane]);

    // which halo pixel / 16-byte part a thread stages, and where it lands in LDS, does not depend on the tile: computed once
    uint4 pre[NFILL];
    int f_lds[NFILL];       // LDS byte offset (-1: none)
    unsigned f_rel[NFILL];  // global BYTE offset relative to the halo origin (y0 - 2, x0 - 2): unsigned 32-bit, so the load is scalar base + lane offset
#pragma unroll
    for (int k = 0; k < NFILL; ++k) {
        const int it = tid + k * kFT, itc = it < VS * VS * NO ? it : 0, q = itc / NO, part16 = itc % NO;
        f_lds[k] = it < VS * VS * NO ? (q / VS) * VRP + (q % VS) * PP + 16 * part16 : -1;
        f_rel[k] = (unsigned)(((q / VS) * W + q % VS) * C + 8 * part16) * 2u;
    }
    auto fetch = [&](const Tile& t) {
        const __half* src = fromx ? x : v;
        if (t.y0 >= 2 && t.y0 + VS - 2 <= H && t.x0 >= 2 && t.x0 + VS - 2 <= W) {  // halo inside the frame (all but the border tiles): scalar origin + fixed lane offsets
            const char* origin = reinterpret_cast<const char*>(src + ((t.b * H + t.y0 - 2) * (size_t)W + t.x0 - 2) * C);
#pragma unroll
            for (int k = 0; k < NFILL; ++k) {
                unsigned o = f_rel[k];
                asm volatile("" : "+v"(o));  // keeps the zero-extension here: hoisted out of the tile loop it becomes a 64-bit lane address (two VGPRs per load, spilled)
                pre[k] = *reinterpret_cast<const uint4*>(origin + o);
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < NFILL; ++k) {  // border tiles: where the element sits is read back from its LDS offset (registers are what this kernel is short of)
            const int fl = f_lds[k] < 0 ? 0 : f_lds[k], row = fl / VRP, col = (fl - row * VRP) / PP, f_off = (fl - row * VRP - col * PP) / 2;
            const int yy = t.y0 + row - 2, xx = t.x0 + col - 2;
            const bool ok = f_lds[k] >= 0 && yy >= 0 && yy < H && xx >= 0 && xx < W;
            const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy), xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            uint4 r = *reinterpret_cast<const uint4*>(src + ((t.b * H + yc) * (size_t)W + xc) * C + f_off);
            r.x = ok ? r.x : 0u; r.y = ok ? r.y : 0u; r.z = ok ? r.z : 0u; r.w = ok ? r.w : 0u;
            pre[k] = r;
        }
    };
    auto fill = [&]() {
        unsigned char* dst = fromx ? mt : vt;
#pragma unroll
        for (int k = 0; k < NFILL; ++k)
            if (f_lds[k] >= 0) *reinterpret_cast<uint4*>(dst + f_lds[k]) = pre[k];
    };
    // phase B items of this lane (pixels of the 18 x 18 map: part * per + lane + 64 i): LDS offsets and image offsets, once
    constexpr int NB = (((MS * MS + (8 / NO) - 1) / (8 / NO)) + 63) / 64;
    int b_src[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int q = part * per + lane + 64 * i;
        const bool ok = q < (part + 1) * per && q < MS * MS;
        const int qc = ok ? q : 0, my = qc / MS, mx = qc % MS;
        b_src[i] = ok ? my * VRP + mx * PP + 16 * oct : -1;  // in vt: the window's first tap; in mt: the pixel itself
    }
    long tile = blockIdx.x;
    if (tile >= total) return;
    TileWalk walk, ahead;  // this tile, the next one
    walk.init(tile, gridDim.x, tx, ty);
    ahead = walk;
    ahead.advance();
    Tile t = walk.tile(TS);
    fetch(t);
    for (;;) {
        fill();
        __syncthreads();  // vt (or the x tile) complete (and the tables, first time round)
        if (fromx) {
            // ---- phase A: v = float16(x W_v^T) on the 20 x 20 halo region: 32-pixel groups x 32-channel tiles over the waves ----
            constexpr int NG = (VS * VS + 31) / 32;
            for (int task = wave; task < NG * NT; task += kFT / 64) {
                const int g = task / NT, nt = task % NT;
                const int q = 32 * g + p, qc = q < VS * VS ? q : VS * VS - 1;
                const int lo = (qc / VS) * VRP + (qc % VS) * PP;
                float16_t d;
#pragma unroll
                for (int vv = 0; vv < 16; ++vv) d[vv] = 0.f;
#pragma unroll
                for (int s2 = 0; s2 < NS; ++s2)
                    d = mfma16(__builtin_bit_cast(half8_t, wvpack[(size_t)(nt * NS + s2) * 64 + lane]),
                               __builtin_bit_cast(half8_t, *reinterpret_cast<const uint4*>(mt + lo + 32 * s2 + 16 * h)), d);
                if (q < VS * VS) {
                    half8_t o0, o1;
#pragma unroll
                    for (int vv = 0; vv < 8; ++vv) { o0[vv] = (_Float16)d[vv]; o1[vv] = (_Float16)d[8 + vv]; }
                    reinterpret_cast<uint4*>(vt + lo + 64 * nt + 32 * h)[0] = __builtin_bit_cast(uint4, o0);
                    reinterpret_cast<uint4*>(vt + lo + 64 * nt + 32 * h)[1] = __builtin_bit_cast(uint4, o1);
                }
            }
            __syncthreads();  // vt complete; the x tile is dead: its region becomes mt
        }
        // ---- phase B: mid = gelu(dw1(v)) on the 18 x 18 region, zero outside the image ----
        constexpr bool HOIST = C == 64;  // two waves per SIMD: the nine tap vectors of this wave's octet are read once per tile, not once per item (a third of phase B's LDS reads)
        uint4 wt1[HOIST ? 9 : 1];
        if constexpr (HOIST) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) wt1[tap] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(t1l) + (size_t)tap * C * 2 + 16 * oct);
        }
        const bool mid_inside = t.y0 >= 1 && t.y0 + MS - 1 <= H && t.x0 >= 1 && t.x0 + MS - 1 <= W;  // scalar: no per-pixel test, no masking below
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (b_src[i] < 0) continue;
            const unsigned char* src = vt + b_src[i];
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const uint4 hv = *reinterpret_cast<const uint4*>(src + (tap / 3) * VRP + (tap % 3) * PP);
                const uint4 wv = HOIST ? wt1[HOIST ? tap : 0] : *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(t1l) + (size_t)tap * C * 2 + 16 * oct);
                fma_mix_lo(acc[0], hv.x, wv.x); fma_mix_hi(acc[1], hv.x, wv.x);
                fma_mix_lo(acc[2], hv.y, wv.y); fma_mix_hi(acc[3], hv.y, wv.y);
                fma_mix_lo(acc[4], hv.z, wv.z); fma_mix_hi(acc[5], hv.z, wv.z);
                fma_mix_lo(acc[6], hv.w, wv.w); fma_mix_hi(acc[7], hv.w, wv.w);
            }
            bool inside = true;
            if (!mid_inside) {  // border tiles only
                const int my = b_src[i] / MRP, mx = (b_src[i] - my * MRP) / PP, yy = t.y0 + my - 1, xx = t.x0 + mx - 1;
                inside = yy >= 0 && yy < H && xx >= 0 && xx < W;
            }
            half8_t o;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const float2_t gv = gelu_fast2(float2_t{acc[j], acc[j + 1]});
                o[j] = (_Float16)gv.x;
                o[j + 1] = (_Float16)gv.y;
            }
            uint4 ov = __builtin_bit_cast(uint4, o);
            if (!mid_inside) {
                const uint32_t keep = inside ? 0xffffffffu : 0u;  // the second conv's zero padding applies to THIS map
                ov.x &= keep; ov.y &= keep; ov.z &= keep; ov.w &= keep;
            }
            *reinterpret_cast<uint4*>(mt + b_src[i]) = ov;
        }
        __syncthreads();  // mt complete
        const long next = tile + gridDim.x;
        if (next < total) fetch(ahead.tile(TS));  // in flight during phase C
        // ---- phase C: dw2(mid) + v @ M + bias + x for this wave's 32 pixels (rows 2 wave, 2 wave + 1) ----
        {
            const int r = 2 * wave + (p >> 4), c = p & 15;
            const int yo = t.y0 + r, xo = t.x0 + c;
            const bool live = yo < H && xo < W;
            const size_t off = ((t.b * H + (live ? yo : 0)) * (size_t)W + (live ? xo : 0)) * C + h * (C / 2);
            uint4 xr[2 * NT];
#pragma unroll
            for (int i = 0; i < 2 * NT; ++i) xr[i] = reinterpret_cast<const uint4*>(x + off)[i];
            float16_t D[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int vv = 0; vv < 16; ++vv) D[n][vv] = 0.f;
            float pe[NS][8];
            const unsigned char* msrc = mt + (size_t)r * MRP + (size_t)c * PP + (size_t)h * C;        // octet h * NS + s: + 16 s bytes
            const unsigned char* vsrc = vt + (size_t)(r + 2) * VRP + (size_t)(c + 2) * PP + (size_t)h * C;
            const unsigned char* tsrc = reinterpret_cast<const unsigned char*>(t2l) + (size_t)h * C;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
#pragma unroll
                for (int j = 0; j < 8; ++j) pe[s][j] = 0.f;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const uint4 hv = *reinterpret_cast<const uint4*>(msrc + (size_t)(tap / 3) * MRP + (size_t)(tap % 3) * PP + 16 * s);
                    const uint4 wv = *reinterpret_cast<const uint4*>(tsrc + (size_t)tap * C * 2 + 16 * s);
                    fma_mix_lo(pe[s][0], hv.x, wv.x); fma_mix_hi(pe[s][1], hv.x, wv.x);
                    fma_mix_lo(pe[s][2], hv.y, wv.y); fma_mix_hi(pe[s][3], hv.y, wv.y);
                    fma_mix_lo(pe[s][4], hv.z, wv.z); fma_mix_hi(pe[s][5], hv.z, wv.z);
                    fma_mix_lo(pe[s][6], hv.w, wv.w); fma_mix_hi(pe[s][7], hv.w, wv.w);
                }
                const half8_t bv = __builtin_bit_cast(half8_t, *reinterpret_cast<const uint4*>(vsrc + 16 * s));
#pragma unroll
                for (int n = 0; n < NT; ++n) D[n] = mfma16(mf[n * NS + s], bv, D[n]);
            }
            if (live) {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const float* bb = bl + h * (C / 2) + 16 * n;
                    // ((D + float16(pe)) + bias) + x, the float16 operands taken as they are by v_fma_mix_f32 (a * 1.0 + c: the same sum, no conversion instructions)
                    constexpr unsigned kOnes = 0x3c003c00u;
                    const uint4 xa = xr[2 * n], xb = xr[2 * n + 1];
                    const unsigned xw[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
                    unsigned ow[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {  // output channels 2 j, 2 j + 1 of this lane's 16
                        const float* pj = j < 4 ? &pe[2 * n][2 * j] : &pe[2 * n + 1][2 * j - 8];
                        const unsigned pk = pack_f16(pj[0], pj[1]);  // pos_emb's own float16 rounding
                        float s0 = D[n][2 * j], s1 = D[n][2 * j + 1];
                        fma_mix_lo(s0, pk, kOnes); fma_mix_hi(s1, pk, kOnes);
                        s0 += bb[2 * j]; s1 += bb[2 * j + 1];
                        fma_mix_lo(s0, xw[j], kOnes); fma_mix_hi(s1, xw[j], kOnes);
                        ow[j] = pack_f16(s0, s1);
                    }
                    reinterpret_cast<uint4*>(out + off)[2 * n] = uint4{ow[0], ow[1], ow[2], ow[3]};
                    reinterpret_cast<uint4*>(out + off)[2 * n + 1] = uint4{ow[4], ow[5], ow[6], ow[7]};
                }
            }
        }
        if (next >= total) break;
        tile = next;
        walk = ahead;
        ahead.advance();
        t = walk.tile(TS);
        __syncthreads();  // everyone is done reading vt / mt
    }
}

template <int C, int MINW>
int launch_attn_tail(avx_ctx* ctx, const void* v, const void* x, const void* mpack, const void* taps1, const void* taps2, const float* bias, void* out, int B,
                     int H, int W, hipStream_t s, const void* wvpack) {
    constexpr int PP = C * 2 + 16, VRP = (VS * PP + 255) / 256 * 256, MRP = VRP;
    const size_t lds = (size_t)2 * VS * VRP + (size_t)2 * 9 * C * 2 + sizeof(float) * C;  // second region sized for the x tile (>= mt)
    (void)MRP;
    const long total = (long)B * ((H + TS - 1) / TS) * ((W + TS - 1) / TS);
    const long cap = (long)ctx->num_cus * (MINW / 2);
    auto k = k_mst_attn_tail<C, MINW>;
    AVX_HIP(ctx, hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)(total < cap ? total : cap)), dim3(kFT), lds, s, (const __half*)v, (const __half*)x, (const uint4*)mpack,
                       (const __half*)taps1, (const __half*)taps2, bias, (__half*)out, B, H, W, (const uint4*)wvpack);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}


// ---- attention tail with BOTH depthwise convs and the projection on the matrix pipe (round 3) ----------------------------------------------
// out = dw2(gelu(dw1(v))) + v @ M + bias + x with v = float16(x W_v^T) formed on the tile's halo (MS_MSA :96, :104-106, :132-137; MSAB :183), as
// k_mst_attn_tail above, but the vector unit is left with the GELU and the final sums:
//   * a depthwise 3x3 conv is three v_mfma_f32_16x16x32_f16 per "unit" = 8 channels (an octet: one 16-byte LDS read) x 2 vertically adjacent
//     output rows x 16 pixels: the K slots are the 4 input rows the row pair needs x 3 column shifts, the weight fragment is diagonal in the
//     channels (pack_dw_mfma; see k_mst_ffn_fused);
//   * v @ M lands in the SAME accumulator: its fragment is block diagonal in the two rows -- slot q of step t carries row q & 1, input
//     channels 16 t + 8 (q >> 1) ... + 7, and lane (m = 8 s + c, q) of the weight holds M[.][8 o + c] when (q & 1) == s (avx_mst_attn_pack_mx);
//     pos_emb + projection meet in float32 without the float16 rounding of pos_emb's result;
//   * tiles are 14 pixels wide: the first conv's output (needed one pixel beyond the tile on every side) is then exactly 16 wide -- one MFMA
//     column block, no GELU on columns nobody reads -- and v is 18 wide; TR rows tall (v: TR + 4, mid: TR + 2 rows);
//   * depthwise convs do not mix channels, so after v is complete a wave needs NOBODY ELSE: wave w owns octet w % NOCT (and, where two waves
//     share an octet, half of the tile's rows: the one mid row pair both halves need is computed by both), runs conv1 -> GELU -> mid (its own
//     bytes of the LDS map) -> conv2 + projection for its units back to back and stores its 8 bytes per lane and pixel straight to memory with
//     the residual (requested before the first unit) added: two barriers per tile (v complete; v free), no output staging;
//   * v is formed from x as it arrives: a lane's prefetched half row of its halo pixel IS its B operand (W_v packed in that K order,
//     pack_fragments16(halfrow=True)); the x tile never sits in LDS.
// NW: waves per workgroup (8; 16 for the 62-channel blocks, whose LDS footprint admits one workgroup per CU: two waves per octet as at C = 32, four waves per SIMD)
template <int C, int TR, int MINW, bool STAMP = false, int NW = 8>
__global__ __launch_bounds__(64 * NW, MINW) void k_mst_attn_tail_mx(const __half* __restrict__ x /*[B][H][W][C]*/, const uint4* __restrict__ wvpack /*[C/32][C/16][64], half-row K order*/,
                                                                const uint4* __restrict__ mpack /*[C/8][C/16][64]*/, const uint4* __restrict__ dw1 /*[C/8][3][64]*/,
                                                                const uint4* __restrict__ dw2 /*[C/8][3][64]*/, const float* __restrict__ bias /*[C] or NULL*/,
                                                                __half* __restrict__ out, int B, int H, int W, unsigned long long* __restrict__ stamps /*STAMP: [blocks][8][8]*/) {
    constexpr int TW = 14, VW = 18, VR = TR + 4, MR = TR + 2, NOCT = C / 8, NK = C / 16, NT = C / 32, NS = C / 16, LNV = C / 16;
    constexpr int PP = C * 2 + 16, RP = (VW * PP + 255) / 256 * 256, MP = 16 * PP;  // pixel pitch; row pitch of v (18 px) and of mid (16 px): multiples of 256 bytes
    constexpr int NTHR = 64 * NW, NG = (VR * VW + 31) / 32, NGW = (NG + NW - 1) / NW;                     // 32-pixel groups of the halo region; per wave
    constexpr int WPO = NOCT >= NW ? 1 : NW / NOCT, OPW = NOCT > NW ? NOCT / NW : 1;  // waves per octet; octets per wave (C = 128: two, taken one after the other, fragments loaded per octet)
    constexpr int NOUT = (TR / 2) / WPO, NMID = NOUT + 1;                             // output / mid row pairs per wave and octet
    static_assert(C == 32 || C == 64 || C == 128, "31-, 62- or 124-channel blocks");
    static_assert(MP % 256 == 0 && (NW % NOCT == 0 || NOCT % NW == 0) && (TR / 2) % WPO == 0, "pitches / row split");
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* vt = smem;                                            // [VR][RP]: v on the halo region (zeros outside the image)
    unsigned char* mt = vt + (size_t)VR * RP;                            // [MR][MP]: mid = gelu(dw1(v)), 16 columns; an octet's bytes belong to its wave(s)
    uint4* wvl = reinterpret_cast<uint4*>(mt + (size_t)MR * MP);         // [NT * NS][64]: W_v's fragments (also absorbs the reads of the two unused columns past the last mid row)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 31, h = lane >> 5, n16 = lane & 15, q = lane >> 4;
    // STAMP (AVX_TAIL_STAMPS=1): cycles per segment and wave -- 0 v from x (incl. waiting for the prefetch), 1 barrier, 2 conv1 + GELU, 3 conv2 + projection + stores, 4 barrier
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    auto stamp = [&](int kk) {
        if constexpr (STAMP) {
            const unsigned long long now = __builtin_readcyclecounter();
            seg[kk] += now - tlast;
            tlast = now;
        }
    };
    if constexpr (STAMP) tlast = __builtin_readcyclecounter();
    const int tx = (W + TW - 1) / TW, ty = (H + TR - 1) / TR;
    const long total = (long)B * ty * tx;
    for (int i = tid; i < NT * NS * 64; i += NTHR) wvl[i] = wvpack[i];
    const int o = wave % NOCT, rpo = (wave / NOCT) * NOUT;  // this wave's (first) octet and first row pair
    half8_t a1[3], a2[3], am[NK];
    float bs[4];
    auto load_frags = [&](int oc) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            a1[i] = __builtin_bit_cast(half8_t, dw1[((size_t)oc * 3 + i) * 64 + lane]);
            a2[i] = __builtin_bit_cast(half8_t, dw2[((size_t)oc * 3 + i) * 64 + lane]);
        }
#pragma unroll
        for (int t2 = 0; t2 < NK; ++t2) am[t2] = __builtin_bit_cast(half8_t, mpack[((size_t)oc * NK + t2) * 64 + lane]);
#pragma unroll
        for (int j = 0; j < 4; ++j) bs[j] = bias ? bias[8 * oc + 4 * (q & 1) + j] : 0.f;
    };
    load_frags(o);
    // One octet per wave: the fragments above stay in registers for the whole launch.  Used once here: the compiler's wait for their loads then sits in front of the
    // tile loop -- left to their first use INSIDE the loop, the in-order load counter would drain every tile's prefetches at that point.
#pragma unroll
    for (int i = 0; i < 3; ++i) asm volatile("" ::"v"(a1[i]), "v"(a2[i]));
#pragma unroll
    for (int t2 = 0; t2 < NK; ++t2) asm volatile("" ::"v"(am[t2]));
    asm volatile("" ::"v"(bs[0]), "v"(bs[1]), "v"(bs[2]), "v"(bs[3]));
    // lane-constant LDS addresses: row pair, column shift and K step are immediates
    const unsigned char* rbB = vt + (size_t)q * RP + (size_t)n16 * PP + 16 * o;                              // first conv's B operand: + 2 rp RP + i PP
    unsigned char* wbB = mt + (size_t)(q >> 1) * MP + (size_t)n16 * PP + 16 * o + 8 * (q & 1);              // mid: + 2 rp MP
    const unsigned char* rbC = mt + (size_t)q * MP + (size_t)n16 * PP + 16 * o;                              // second conv's B operand: + 2 rp MP + i PP
    const unsigned char* rbG = vt + (size_t)(2 + (q & 1)) * RP + (size_t)(n16 + 2) * PP + 16 * (q >> 1);    // projection's B operand: + 2 rp RP + 32 t
    // this wave's halo pixel groups (wave, wave + 8): where a lane's pixel sits in the frame (relative to the halo origin) and in the v map
    unsigned f_rel[NGW];
    int v_dst[NGW], p_row[NGW], p_col[NGW];
#pragma unroll
    for (int gi = 0; gi < NGW; ++gi) {
        const int g = wave + NW * gi, qq = 32 * g + p, qc = qq < VR * VW ? qq : VR * VW - 1;
        p_row[gi] = qc / VW; p_col[gi] = qc % VW;
        f_rel[gi] = (unsigned)((p_row[gi] * W + p_col[gi]) * C + h * (C / 2)) * 2u;
        v_dst[gi] = (g < NG && qq < VR * VW) ? p_row[gi] * RP + p_col[gi] * PP + 32 * h : -1;
    }
    uint4 pre[NGW][LNV];  // a lane's half row of its halo pixels: fetched a tile ahead, consumed as MFMA operands
    auto fetch = [&](const Tile& t) {
        if (t.y0 >= 2 && t.y0 + VR - 2 <= H && t.x0 >= 2 && t.x0 + VW - 2 <= W) {  // halo inside the frame: scalar origin + fixed lane offsets
            const char* origin = reinterpret_cast<const char*>(x + ((t.b * H + t.y0 - 2) * (size_t)W + t.x0 - 2) * C);
#pragma unroll
            for (int gi = 0; gi < NGW; ++gi) {
                if (wave + NW * gi >= NG) break;
                unsigned of = f_rel[gi];
                asm volatile("" : "+v"(of));
                const uint4* src = reinterpret_cast<const uint4*>(origin + of);
#pragma unroll
                for (int v = 0; v < LNV; ++v) pre[gi][v] = src[v];
            }
            return;
        }
#pragma unroll
        for (int gi = 0; gi < NGW; ++gi) {
            if (wave + NW * gi >= NG) break;
            const int yy = t.y0 - 2 + p_row[gi], xx = t.x0 - 2 + p_col[gi];
            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;  // zeros outside the image: v = 0 there (to_v has no bias), the convs' padding
            const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy), xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            const uint4* src = reinterpret_cast<const uint4*>(x + ((t.b * H + yc) * (size_t)W + xc) * C + h * (C / 2));
#pragma unroll
            for (int v = 0; v < LNV; ++v) {
                uint4 r = src[v];
                r.x = ok ? r.x : 0u; r.y = ok ? r.y : 0u; r.z = ok ? r.z : 0u; r.w = ok ? r.w : 0u;
                pre[gi][v] = r;
            }
        }
    };
    long tile = blockIdx.x;
    if (tile >= total) return;
    TileWalk walk, ahead;
    walk.init(tile, gridDim.x, tx, ty);
    ahead = walk;
    ahead.advance();
    auto tile_at = [&](const TileWalk& wk) { return Tile{wk.xi * TW, wk.yi * TR, wk.b}; };
    Tile t = tile_at(walk);
    fetch(t);
    __syncthreads();  // W_v's fragments are in LDS
    for (;;) {
        // ---- phase A: v = float16(x W_v^T) for this wave's halo pixel groups, straight from the prefetched rows ----
#pragma unroll
        for (int gi = 0; gi < NGW; ++gi) {
            if (wave + NW * gi >= NG) break;  // wave-uniform
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float16_t d;
#pragma unroll
                for (int vv = 0; vv < 16; ++vv) d[vv] = 0.f;
#pragma unroll
                for (int s2 = 0; s2 < NS; ++s2) d = mfma16(__builtin_bit_cast(half8_t, wvl[(nt * NS + s2) * 64 + lane]), __builtin_bit_cast(half8_t, pre[gi][s2]), d);
                if (v_dst[gi] >= 0) {
                    half8_t w0, w1;
#pragma unroll
                    for (int vv = 0; vv < 8; ++vv) { w0[vv] = (_Float16)d[vv]; w1[vv] = (_Float16)d[8 + vv]; }
                    reinterpret_cast<uint4*>(vt + v_dst[gi] + 64 * nt)[0] = __builtin_bit_cast(uint4, w0);
                    reinterpret_cast<uint4*>(vt + v_dst[gi] + 64 * nt)[1] = __builtin_bit_cast(uint4, w1);
                }
            }
        }
        stamp(0);
        __syncthreads();  // v complete
        stamp(1);
        const long next = tile + gridDim.x;
        if (next < total) fetch(tile_at(ahead));  // the next tile's halo rows: in flight while this wave works through its units
#pragma unroll
        for (int kk = 0; kk < OPW; ++kk) {
        const int oc = o + NW * kk, ob = 16 * NW * kk;  // this pass's octet; its byte offset from the lane-constant addresses (which carry octet o)
        if constexpr (OPW > 1) load_frags(oc);
        // the residual's pieces of x for this wave's output units (8 bytes per lane and unit): requested now, added at the stores
        uint2 xr[NOUT];
        size_t e_off[NOUT];
#pragma unroll
        for (int u = 0; u < NOUT; ++u) {
            const int yo = t.y0 + 2 * (rpo + u) + (q >> 1), xo = t.x0 + n16;
            const bool live = n16 < TW && yo < H && xo < W;
            e_off[u] = live ? ((t.b * H + yo) * (size_t)W + xo) * C + 8 * oc + 4 * (q & 1) : ~(size_t)0;
            xr[u] = live ? *reinterpret_cast<const uint2*>(x + e_off[u]) : uint2{0, 0};
        }
        // ---- conv1 + GELU -> mid, this wave's octet, mid row pairs rpo ... rpo + NOUT (zero outside the image: the second conv's padding applies to THIS map) ----
        {
            const bool mid_inside = t.y0 >= 1 && t.y0 + MR - 1 <= H && t.x0 >= 1 && t.x0 + 15 <= W;  // scalar
            // software pipeline, one unit deep (as k_mst_ffn_fused's phase 2a): unit u's MFMAs | reads of unit u + 1 | GELU + store of unit u - 1
            uint4 bq[3];
            auto rd = [&](int u) {
#pragma unroll
                for (int i = 0; i < 3; ++i) bq[i] = *reinterpret_cast<const uint4*>(rbB + (size_t)(2 * (rpo + u)) * RP + i * PP + ob);
            };
            auto finish = [&](int u, float4_t a) {
                float2_t gp[2] = {float2_t{a[0], a[1]}, float2_t{a[2], a[3]}};
                if constexpr (C >= 64) gelu_mx<2>(gp);
                else { float2_t g0[1] = {gp[0]}, g1[1] = {gp[1]}; gelu_mx<1>(g0); gelu_mx<1>(g1); gp[0] = g0[0]; gp[1] = g1[0]; }
                uint2 ov = uint2{pack_f16(gp[0].x, gp[0].y), pack_f16(gp[1].x, gp[1].y)};
                if (!mid_inside) {
                    const int yy = t.y0 - 1 + 2 * (rpo + u) + (q >> 1), xx = t.x0 - 1 + n16;
                    const unsigned keep = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? 0xffffffffu : 0u;
                    ov.x &= keep; ov.y &= keep;
                }
                *reinterpret_cast<uint2*>(wbB + (size_t)(2 * (rpo + u)) * MP + ob) = ov;
            };
            rd(0);
            float4_t prev = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < NMID; ++u) {
                float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 3; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[i], __builtin_bit_cast(half8_t, bq[i]), acc, 0, 0, 0);
                if (u + 1 < NMID) rd(u + 1);
                __builtin_amdgcn_sched_barrier(0);
                if (u > 0) finish(u - 1, prev);
                __builtin_amdgcn_sched_barrier(0);
                prev = acc;
            }
            finish(NMID - 1, prev);
        }
        stamp(2);
        // ---- conv2(mid) + v @ M + bias + x for this wave's output units, stored from registers ----
        {
            uint4 bq[3], bg[NK];
            auto rd = [&](int u) {
#pragma unroll
                for (int i = 0; i < 3; ++i) bq[i] = *reinterpret_cast<const uint4*>(rbC + (size_t)(2 * (rpo + u)) * MP + i * PP + ob);
#pragma unroll
                for (int t2 = 0; t2 < NK; ++t2) bg[t2] = *reinterpret_cast<const uint4*>(rbG + (size_t)(2 * (rpo + u)) * RP + 32 * t2);
            };
            rd(0);
#pragma unroll
            for (int u = 0; u < NOUT; ++u) {
                float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 3; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[i], __builtin_bit_cast(half8_t, bq[i]), acc, 0, 0, 0);
#pragma unroll
                for (int t2 = 0; t2 < NK; ++t2) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(am[t2], __builtin_bit_cast(half8_t, bg[t2]), acc, 0, 0, 0);
                if (u + 1 < NOUT) rd(u + 1);
                if (e_off[u] != ~(size_t)0) {  // ((pos_emb + projection) + bias) + x: the float16 residual taken as it is (v_fma_mix_f32: x * 1.0 + sum)
                    constexpr unsigned kOnes = 0x3c003c00u;
                    float s0 = acc[0] + bs[0], s1 = acc[1] + bs[1], s2 = acc[2] + bs[2], s3 = acc[3] + bs[3];
                    fma_mix_lo(s0, xr[u].x, kOnes); fma_mix_hi(s1, xr[u].x, kOnes);
                    fma_mix_lo(s2, xr[u].y, kOnes); fma_mix_hi(s3, xr[u].y, kOnes);
                    *reinterpret_cast<uint2*>(out + e_off[u]) = uint2{pack_f16(s0, s1), pack_f16(s2, s3)};
                }
            }
        }
        }  // octets of this wave
        stamp(3);
        if (next >= total) break;
        tile = next;
        walk = ahead;
        ahead.advance();
        t = tile_at(walk);
        __syncthreads();  // every wave is done reading v: the next tile's may be written
        stamp(4);
    }
    if constexpr (STAMP) {
        if (lane == 0 && stamps)
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) stamps[((size_t)blockIdx.x * NW + wave) * 8 + kk] = seg[kk];
    }
}

template <int C, int TR, int MINW, int NW = 8>
int launch_attn_tail_mx(avx_ctx* ctx, const void* x, const void* wvpack, const void* mpack, const void* dw1, const void* dw2, const float* bias, void* out, int B, int H, int W,
                        hipStream_t s) {
    constexpr int PP = C * 2 + 16, RP = (18 * PP + 255) / 256 * 256;
    const size_t lds = (size_t)(TR + 4) * RP + (size_t)(TR + 2) * 16 * PP + (size_t)(C / 32) * (C / 16) * 1024;
    const long total = (long)B * ((H + TR - 1) / TR) * ((W + 13) / 14);
    const long cap = (long)ctx->num_cus * (NW == 16 ? 1 : MINW / 2);
    const unsigned blocks = (unsigned)(total < cap ? total : cap);
    if (getenv("AVX_TAIL_STAMPS")) {  // diagnostic: per-segment cycles of every wave, summed and printed
        auto ks = k_mst_attn_tail_mx<C, TR, MINW, true, NW>;
        unsigned long long* d_st = nullptr;
        const size_t n = (size_t)blocks * NW * 8;
        AVX_HIP(ctx, hipMalloc((void**)&d_st, n * sizeof(unsigned long long)));
        AVX_HIP(ctx, hipFuncSetAttribute((const void*)ks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(ks, dim3(blocks), dim3(64 * NW), lds, s, (const __half*)x, (const uint4*)wvpack, (const uint4*)mpack, (const uint4*)dw1, (const uint4*)dw2, bias, (__half*)out, B, H, W, d_st);
        AVX_HIP(ctx, hipStreamSynchronize(s));
        unsigned long long* h_st = (unsigned long long*)malloc(n * sizeof(unsigned long long));
        AVX_HIP(ctx, hipMemcpy(h_st, d_st, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double tot[8] = {0}, all = 0;
        for (size_t i = 0; i < n; ++i) { tot[i & 7] += (double)h_st[i]; all += (double)h_st[i]; }
        const char* nm[8] = {"v from x", "bar1", "conv1+gelu", "conv2+proj+store", "bar2", "-", "-", "-"};
        fprintf(stderr, "[tail stamps C=%d] %u blocks, %ld tiles, mean cycles per wave %.0f:", C, blocks, total, all / (blocks * (double)NW));
        for (int k2 = 0; k2 < 8; ++k2) fprintf(stderr, "  %s %.1f%%", nm[k2], 100 * tot[k2] / all);
        fprintf(stderr, "\n");
        free(h_st);
        (void)hipFree(d_st);
        return AVX_OK;
    }
    auto k = k_mst_attn_tail_mx<C, TR, MINW, false, NW>;
    AVX_HIP(ctx, hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * NW), lds, s, (const __half*)x, (const uint4*)wvpack, (const uint4*)mpack, (const uint4*)dw1,
                       (const uint4*)dw2, bias, (__half*)out, B, H, W, (unsigned long long*)nullptr);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}


// ---- dense 3x3 conv (31 -> 31 channels, zero padding 1) [+ add] through an LDS halo tile: MST.embedding / mapping + x / conv_out + x ----
// (:199, :228, :277).  The register-only form (csrc/mst_mfma.hip::k_mst_conv3x3) fetches every tap's row from L2 per wave -- nine
// reads of each pixel, 470 MB of fetch traffic per 1080p launch against 133 MB of input; here a workgroup stages its 18 x 18 halo
// region once (next tile's rows in flight during the MFMAs) and the nine taps are 16-byte LDS reads.  Implicit GEMM, K = 9 x 32.
// GRAM: the MSAB block that follows the conv starts with the Gram pass over exactly this output (q = out W_q^T, k = out W_k^T, k^T q and the column
// norms over all pixels, MS_MSA :118-129).  The lane that has just rounded its pixel's 16 channels to float16 holds them in the K = 16 operand order
// of avx_mst_qkv_gram16 (channel 16 h + 8 s + j), so the pass runs here on registers -- four projection MFMAs, two Gram MFMAs per 32 pixels --
// and the 531 MB re-read of the tensor (121 us per 4K launch) does not happen.  Per-workgroup partial sums, reduced by k_mst_qkv_final as there.
// SPEC (round 3): the conv is conv_out + x (MST_Plus_Plus.py:289-292) and what follows is the spectral integration of the cube it would write
// (honeybee.py:126-135: radiance = cube * illuminant, three cone catches = 31 -> 3 GEMV per pixel).  The lane that has just rounded its pixel's 16 bands to float16
// holds half of that pixel's cube row: the catches are formed here -- the SAME float32 FMA chain over the bands 0 ... 31 as csrc/uv.hip::k_spectral_nhwc_h runs
// (bands 0-15 in lane half 0, handed to lane half 1 for bands 16-31), so the planes are bit-identical to the two-kernel route -- written as three float32 planes
// of the cropped frame, with the per-workgroup min / max / sum the von Kries step needs (Stat3 as in csrc/uv.hip).  The cube is never written: 531 MB less to
// write and 531 MB less to read per 4K frame, one launch less.
struct SpecStat3 { float mn, mx; double sum; };  // == csrc/uv.hip::Stat3
template <int MINW, bool GRAM, bool SPEC = false>
__global__ __launch_bounds__(kFT, MINW) void k_mst_conv3x3_lds(const __half* __restrict__ x /*[B][H][W][32]*/, const uint4* __restrict__ wpack /*[9][2][64]*/,
                                                               const __half* __restrict__ add /*or NULL*/, __half* __restrict__ out, int B, int H, int W,
                                                               const uint4* __restrict__ wqk /*[2][2][64]: pack_qkv16's q and k tiles*/, float* __restrict__ partial /*[blocks][34][32]*/,
                                                               const float* __restrict__ specw = nullptr /*SPEC: [32 bands][4]: the three catches' weights, band-major*/,
                                                               float* __restrict__ planes = nullptr /*SPEC: [3][Hc * Wc]*/, SpecStat3* __restrict__ spart = nullptr /*SPEC: [blocks][3]*/,
                                                               int crop_t = 0, int crop_l = 0, int Hc = 0, int Wc = 0) {
    constexpr int C = 32, PP = C * 2 + 16, RP = (HS * PP + 255) / 256 * 256, NFILL = (HS * HS * 4 + kFT - 1) / kFT;
    __shared__ __align__(16) unsigned char xt[HS * RP];
    __shared__ uint4 wl[18 * 64];  // the 18 A fragments (9 taps x 2 K-steps): in registers they cost 72 VGPRs and a wave of occupancy
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 31, h = lane >> 5;
    for (int i = tid; i < 18 * 64; i += kFT) wl[i] = wpack[i];
    __shared__ uint4 wq[GRAM ? 4 * 64 : 1];
    float16_t G;
    float nq = 0.f, nk = 0.f;
    if constexpr (GRAM) {
        if (tid < 4 * 64) wq[tid] = wqk[tid];
#pragma unroll
        for (int v = 0; v < 16; ++v) G[v] = 0.f;
    }
    float smn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, smx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    double ssum[3] = {0.0, 0.0, 0.0};
    const int tx = (W + TS - 1) / TS, ty = (H + TS - 1) / TS;
    const long total = (long)B * ty * tx;
    uint4 pre[NFILL];
    unsigned f_rel[NFILL];  // this thread's halo elements as byte offsets from the halo origin (y0 - 1, x0 - 1)
#pragma unroll
    for (int k = 0; k < NFILL; ++k) {
        const int it = tid + k * kFT, itc = it < HS * HS * 4 ? it : 0, q = itc >> 2, part = itc & 3;
        f_rel[k] = (unsigned)(((q / HS) * W + q % HS) * C + 8 * part) * 2u;
    }
    auto fetch = [&](const Tile& t) {
        if (t.y0 >= 1 && t.y0 + HS - 1 <= H && t.x0 >= 1 && t.x0 + HS - 1 <= W) {  // halo inside the frame: scalar origin + fixed lane offsets
            const char* origin = reinterpret_cast<const char*>(x + ((t.b * H + t.y0 - 1) * (size_t)W + t.x0 - 1) * C);
#pragma unroll
            for (int k = 0; k < NFILL; ++k) {
                unsigned o = f_rel[k];
                asm volatile("" : "+v"(o));  // keeps the zero-extension here (see k_mst_attn_tail)
                pre[k] = *reinterpret_cast<const uint4*>(origin + o);
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < NFILL; ++k) {
            const int it = tid + k * kFT, q = it >> 2, part = it & 3;
            const int yy = t.y0 - 1 + q / HS, xx = t.x0 - 1 + q % HS;
            const bool ok = it < HS * HS * 4 && yy >= 0 && yy < H && xx >= 0 && xx < W;
            const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy), xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            uint4 r = *reinterpret_cast<const uint4*>(x + ((t.b * H + yc) * (size_t)W + xc) * C + 8 * part);
            r.x = ok ? r.x : 0u; r.y = ok ? r.y : 0u; r.z = ok ? r.z : 0u; r.w = ok ? r.w : 0u;
            pre[k] = r;
        }
    };
    long tile = blockIdx.x;
    if (tile >= total) return;
    TileWalk walk, ahead;  // this tile, the next one
    walk.init(tile, gridDim.x, tx, ty);
    ahead = walk;
    ahead.advance();
    Tile t = walk.tile(TS);
    fetch(t);
    for (;;) {
#pragma unroll
        for (int k = 0; k < NFILL; ++k) {
            const int it = tid + k * kFT, q = it >> 2, part = it & 3;
            if (it < HS * HS * 4) *reinterpret_cast<uint4*>(xt + (size_t)(q / HS) * RP + (size_t)(q % HS) * PP + 16 * part) = pre[k];
        }
        __syncthreads();
        const long next = tile + gridDim.x;
        if (next < total) fetch(ahead.tile(TS));
        {
            const int r = 2 * wave + (p >> 4), c = p & 15;
            const int yo = t.y0 + r, xo = t.x0 + c;
            const bool live = yo < H && xo < W;
            const size_t off = ((t.b * H + (live ? yo : 0)) * (size_t)W + (live ? xo : 0)) * C + 16 * h;
            uint4 ar[2] = {uint4{0, 0, 0, 0}, uint4{0, 0, 0, 0}};
            if (add) { ar[0] = reinterpret_cast<const uint4*>(add + off)[0]; ar[1] = reinterpret_cast<const uint4*>(add + off)[1]; }
            float16_t d;
#pragma unroll
            for (int v = 0; v < 16; ++v) d[v] = 0.f;
            const unsigned char* src = xt + (size_t)r * RP + (size_t)c * PP + 16 * h;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                    d = mfma16(__builtin_bit_cast(half8_t, wl[(2 * tap + s2) * 64 + lane]), __builtin_bit_cast(half8_t, *reinterpret_cast<const uint4*>(src + (size_t)(tap / 3) * RP + (size_t)(tap % 3) * PP + 32 * s2)), d);
            half8_t o0, o1;
            {
                const half8_t r0 = __builtin_bit_cast(half8_t, ar[0]), r1 = __builtin_bit_cast(half8_t, ar[1]);
#pragma unroll
                for (int v = 0; v < 8; ++v) { o0[v] = (_Float16)(d[v] + (float)r0[v]); o1[v] = (_Float16)(d[8 + v] + (float)r1[v]); }
            }
            if constexpr (SPEC) {
                // catches of this pixel: acc_k = fma(band_b, w[b][k], acc_k) for b = 0 ... 31 in that order (uv.hip's chain), the first 16 bands in lane half 0
                float a3[3] = {0.f, 0.f, 0.f};
                auto chain = [&]() {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const float v = (float)(j < 8 ? o0[j] : o1[j - 8]);
                        const float* wb = specw + (16 * h + j) * 4;
#pragma unroll
                        for (int k = 0; k < 3; ++k) a3[k] = __builtin_fmaf(v, wb[k], a3[k]);
                    }
                };
                if (h == 0) chain();
#pragma unroll
                for (int k = 0; k < 3; ++k) { const float lo = __shfl_xor(a3[k], 32); a3[k] = h ? lo : a3[k]; }
                if (h == 1) chain();
                const int yc = yo - crop_t, xc = xo - crop_l;
                if (h == 1 && live && yc >= 0 && yc < Hc && xc >= 0 && xc < Wc) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        planes[(size_t)k * Hc * Wc + (size_t)yc * Wc + xc] = a3[k];
                        smn[k] = fminf(smn[k], a3[k]); smx[k] = fmaxf(smx[k], a3[k]); ssum[k] += (double)a3[k];
                    }
                }
            } else if (live) {
                reinterpret_cast<uint4*>(out + off)[0] = __builtin_bit_cast(uint4, o0);
                reinterpret_cast<uint4*>(out + off)[1] = __builtin_bit_cast(uint4, o1);
            }
            if constexpr (GRAM) {
                if (!live) { o0 = half8_t{0, 0, 0, 0, 0, 0, 0, 0}; o1 = o0; }  // pixels past the frame's edge take no part in the sums
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
        if (next >= total) break;
        tile = next;
        walk = ahead;
        ahead.advance();
        t = walk.tile(TS);
        __syncthreads();  // everyone is done reading xt
    }
    if constexpr (SPEC) {  // this workgroup's statistics of the three planes (every workgroup writes its slot: k_finalize_stats reads them all)
        __syncthreads();
        float* redf = reinterpret_cast<float*>(xt);              // [8 waves][3][2]
        double* redd = reinterpret_cast<double*>(xt + 256);      // [8 waves][3]
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float a = smn[k], b2 = smx[k];
            double c2 = ssum[k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { a = fminf(a, __shfl_xor(a, o)); b2 = fmaxf(b2, __shfl_xor(b2, o)); c2 += __shfl_xor(c2, o); }
            if (lane == 0) { redf[(wave * 3 + k) * 2] = a; redf[(wave * 3 + k) * 2 + 1] = b2; redd[wave * 3 + k] = c2; }
        }
        __syncthreads();
        if (tid < 3) {
            float a = 3.4e38f, b2 = -3.4e38f;
            double c2 = 0.0;
            for (int w8 = 0; w8 < 8; ++w8) { a = fminf(a, redf[(w8 * 3 + tid) * 2]); b2 = fmaxf(b2, redf[(w8 * 3 + tid) * 2 + 1]); c2 += redd[w8 * 3 + tid]; }
            spart[(size_t)blockIdx.x * 3 + tid] = SpecStat3{a, b2, c2};
        }
    }
    if constexpr (GRAM) {  // this workgroup's partial: [34][32] = 32 Gram rows (i = k channel, j = q channel), sum q^2, sum k^2 (as k_mst_qkv16)
        static_assert(sizeof(xt) >= sizeof(float) * 4 * 34 * 32, "the halo tile's LDS holds four waves' results at a time");
        float* red = reinterpret_cast<float*>(xt);
        const float sq = nq + __shfl_xor(nq, 32), sk = nk + __shfl_xor(nk, 32);
        float sum[3] = {0.f, 0.f, 0.f};  // entries tid, tid + 512, tid + 1024 of the 1,088
#pragma unroll
        for (int half = 0; half < 2; ++half) {  // waves 0-3, then waves 4-7
            __syncthreads();
            if ((wave >> 2) == half) {
                float* mine = red + (size_t)(wave & 3) * 34 * 32;
#pragma unroll
                for (int v = 0; v < 16; ++v) mine[(8 * (v / 4) + 4 * h + (v % 4)) * 32 + p] = G[v];
                if (h == 0) { mine[32 * 32 + p] = sq; mine[33 * 32 + p] = sk; }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int i = tid + q * kFT;
                if (i < 34 * 32)
#pragma unroll
                    for (int w4 = 0; w4 < 4; ++w4) sum[q] += red[(size_t)w4 * 34 * 32 + i];
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int i = tid + q * kFT;
            if (i < 34 * 32) partial[(size_t)blockIdx.x * 34 * 32 + i] = sum[q];
        }
    }
}


// ---- encoder downsampling: Conv2d(C -> 2C, kernel 4, stride 2, padding 1, no bias), C = 32 and 64 as stored (MST.encoder_layers[i][1], :206-208) ----
// Implicit GEMM, K = 16 taps x C channels.  A workgroup owns OH x 16 output pixels; their (2 OH + 2) x 34 input pixels (zeros outside the
// image) are staged once in LDS; a wave keeps ITS 32 weight fragments (32 output channels x 512 of K) in 128 VGPRs for the whole launch
// -- one workgroup per CU, two waves per SIMD, so there is room -- which leaves one 16-byte LDS read (the pixel operand) per MFMA, issued
// six ahead of the MFMA that takes it.  The halo tile comes in by LDS-direct loads (global_load_lds_dwordx4: lane l's 16 bytes land at
// M0 + 16 l, no VGPR, no ds_write): no prefetch registers, no LDS-write cycles (a ds_write_b128 is 13), and the loads of the next tiles
// (C = 32: three buffers, two tiles ahead) are in flight under the MFMAs of this one -- with one workgroup per CU that depth is what hides
// the HBM latency.  Replaces MIOpen's implicit GEMM (4K: 471 -> 209 us at C = 32, 319 -> 156 us at C = 64, alone on the GPU) and the
// zero fill its split-K form needs.
//   layout   slot (16 bytes) of part j of tile pixel (row, col) = row * RPS + col * NP + (j ^ swz(col)); RPS = 34 NP + 2 (C = 32) / + 4
//            (C = 64) slots and swz(col) = (col >> 2) & 3 / (col >> 1) & 7 make every 16-lane group of the operand's ds_read_b128 --
//            eight pixels two columns apart on each of two tile rows -- cover the 16 slots of the 256-byte bank row exactly once.
//   fill     slot L of the buffer is loaded by lane L % 64 of load L / 64; padding pixels and tiles past the end read 16 zero
//            bytes (ctx->d_zero; pad slots, which nobody reads, whatever is cheapest), so every wave issues the same number of loads per tile and s_waitcnt vmcnt(n) can count them.  The loads are
//            inline asm (the compiler waits for vmcnt(0) in front of every LDS read that follows a load it knows writes the LDS).
//   C = 64   the wave's 128-VGPR weight budget holds 32 output channels x half of K: waves 4-7 take kernel rows 2-3 and hand their partial
//            sums to waves 0-3 through the LDS (one barrier per 32-pixel group).
template <int C>
struct DownGeo {
    static constexpr int OH = C == 32 ? 8 : 4, OW = 16, IH = 2 * OH + 2, IW = 2 * OW + 2;
    static constexpr int NP = C / 8;                                // 16-byte parts of a pixel
    static constexpr int RPS = IW * NP + (C == 32 ? 2 : 4);        // row pitch, slots
    static constexpr int NI = ((IH * RPS + 63) / 64 + 7) / 8 * 8;  // loads per tile, the same count in each of the 8 waves
    static constexpr int PER_WAVE = NI / 8, BUF = NI * 1024, NBUF = C == 32 ? 3 : 2, GROUPS = OH / 2;
    static constexpr int PART = C == 32 ? 0 : 2 * 4 * 4096;        // [group parity][output tile][lane-major float16_t]
    __device__ static int swz(int col) { return C == 32 ? (col >> 2) & 3 : (col >> 1) & 7; }
};
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"  // "m0 is reserved": nothing of the compiler's lives in M0 across these kernels (no indirect register indexing, no LDS-load builtin)
__device__ __forceinline__ void lds_dma16(const void* g, unsigned lds_base /*the same in every lane*/) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(lds_base)), "v"(g) : "memory", "m0");
}
__device__ __forceinline__ void lds_dma16(const void* base /*the same in every lane*/, unsigned ofs, unsigned lds_base) {  // base + ofs: no 64-bit lane arithmetic
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(__builtin_amdgcn_readfirstlane(lds_base)), "v"(ofs), "s"(base) : "memory", "m0");
}
#pragma clang diagnostic pop
template <int C>
__global__ __launch_bounds__(kFT, 2) void k_mst_down4x4_dma(const __half* __restrict__ x /*[B][H][W][C]*/, const uint4* __restrict__ wpack /*[16][2C/32][C/16][64]*/,
                                                            __half* __restrict__ out /*[B][H/2][W/2][2C]*/, const void* __restrict__ zero, int B, int H, int W) {
    using G = DownGeo<C>;
    constexpr int CO = 2 * C, NT = CO / 32, KS = C / 16;
    __shared__ __align__(1024) unsigned char lds[G::NBUF * G::BUF + G::PART];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6) /*scalar: what depends on it stays off the VALU*/, p = lane & 31, h = lane >> 5;
    const int g0 = C == 32 ? wave & 3 : 0, nt = C == 32 ? wave >> 2 : wave & 3, kh = C == 32 ? 0 : wave >> 2;
    half8_t wf[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int tap = C == 32 ? i >> 1 : 8 * kh + (i >> 2), s2 = C == 32 ? i & 1 : i & 3;
        wf[i] = __builtin_bit_cast(half8_t, wpack[((size_t)(tap * NT + nt) * KS + s2) * 64 + lane]);
    }
    const int Ho = H / 2, Wo = W / 2;
    const int tx = (Wo + G::OW - 1) / G::OW, ty = (Ho + G::OH - 1) / G::OH;
    const unsigned total = (unsigned)B * ty * tx;
    struct T2 { int ox0, oy0, b; };
    auto tile_at = [&](const TileWalk& w) { return T2{w.xi * G::OW, w.yi * G::OH, (int)w.b}; };  // coordinates advance with the walk: no division per tile
    // this lane's slots of a tile: tile-relative pixel and byte offset, fixed for the launch
    int rel[G::PER_WAVE], rc[G::PER_WAVE];
#pragma unroll
    for (int k = 0; k < G::PER_WAVE; ++k) {
        const int L = (wave + 8 * k) * 64 + lane, row = L / G::RPS, rem = L - row * G::RPS, col = rem / G::NP, j = (rem - col * G::NP) ^ G::swz(col);
        const bool inside = row < G::IH && col < G::IW;
        rel[k] = inside ? (row * W + col) * (C * 2) + 16 * j : 0;
        rc[k] = inside ? row << 8 | col : -1;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds);
    auto fill = [&](unsigned u, const TileWalk& w, int slot) {
        const bool exists = u < total;  // past the end: the walk's coordinates mean nothing and every lane reads zeros
        const T2 t = tile_at(w);
        const int y0 = 2 * t.oy0 - 1, x0 = 2 * t.ox0 - 1;
        const char* origin = reinterpret_cast<const char*>(x) + (((long)t.b * H + y0) * W + x0) * (long)(C * 2);
        const unsigned lb = lds0 + slot * G::BUF + wave * 1024;
        if (exists && y0 >= 0 && y0 + G::IH <= H && x0 >= 0 && x0 + G::IW <= W) {  // no padding in sight (all but the frame's border tiles): scalar base + the lane's fixed offset
#pragma unroll
            for (int k = 0; k < G::PER_WAVE; ++k) lds_dma16(origin /*scalar: the tile index is*/, (unsigned)rel[k], lb + k * 8192);  // pad slots (offset 0) are never read: anything valid will do
        } else {
#pragma unroll
            for (int k = 0; k < G::PER_WAVE; ++k) {
                const int yy = y0 + (rc[k] >> 8), xx = x0 + (rc[k] & 255);
                const bool ok = exists && rc[k] >= 0 && yy >= 0 && yy < H && xx >= 0 && xx < W;
                lds_dma16(ok ? (const void*)(origin + rel[k]) : zero, lb + k * 8192);
            }
        }
    };
    unsigned tile = blockIdx.x;
    if (tile >= total) return;
    TileWalk walk, look;  // this tile; the one NBUF - 1 steps ahead, whose loads are issued next
    walk.init(tile, gridDim.x, tx, ty);
    look = walk;
#pragma unroll
    for (int a = 0; a < G::NBUF - 1; ++a) { fill(tile + a * gridDim.x, look, a); look.advance(); }
    // operand addressing: pixel (row 2 g + (p >> 4), col p & 15) of the group; tap (ky, kx) reads tile pixel (2 row + ky, 2 col + kx)
    const int c2 = 2 * (p & 15);
    int colofs[4];
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) colofs[kx] = (c2 + kx) * G::NP;
    int slot = 0, plain_stores = 0;  // how many of the last tiles in a row stored every pixel (capped at NBUF - 1)
    for (;;) {
        // This tile's loads have landed once no more operations are pending than were issued after them: the look-ahead tiles' loads
        // (NBUF - 2 tiles) and the output stores of the NBUF - 1 tiles before this one -- two per group when the tile lay inside the frame
        // (vmcnt counts loads and stores in issue order), an unknown number (the compiler may branch around them) when it did not.
        if (plain_stores == G::NBUF - 1 && (C == 32 || kh == 0)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::PER_WAVE * (G::NBUF - 2) + 2 * (C == 32 ? 1 : G::GROUPS) * (G::NBUF - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::PER_WAVE * (G::NBUF - 2)) : "memory");
        __syncthreads();  // every wave's share has landed; every wave is done with the buffer filled next
        {
            int nslot = slot + G::NBUF - 1;
            nslot -= nslot >= G::NBUF ? G::NBUF : 0;
            fill(tile + (G::NBUF - 1) * gridDim.x, look, nslot);
            look.advance();
        }
        const T2 t = tile_at(walk);
        const bool whole = t.oy0 + G::OH <= Ho && t.ox0 + G::OW <= Wo;
        const unsigned char* bt = lds + slot * G::BUF;
#pragma unroll 1
        for (int g = g0; g < (C == 32 ? g0 + 1 : G::GROUPS); ++g) {
            const int r = 2 * g + (p >> 4);
            float16_t d, d1;
#pragma unroll
            for (int v = 0; v < 16; ++v) d[v] = 0.f, d1[v] = 0.f;
            const unsigned char* src = bt + (size_t)(2 * r) * (G::RPS * 16);
            auto frag = [&](int i) {  // i-th weight fragment's pixel operand
                const int tap = C == 32 ? i >> 1 : 8 * kh + (i >> 2), s2 = C == 32 ? i & 1 : i & 3, ky = tap >> 2, kx = tap & 3;
                return *reinterpret_cast<const uint4*>(src + ky * (G::RPS * 16) + (colofs[kx] + ((2 * s2 + h) ^ G::swz(c2 + kx))) * 16);
            };
            constexpr int AHEAD = 6;
            uint4 bq[AHEAD];
#pragma unroll
            for (int i = 0; i < AHEAD; ++i) bq[i] = frag(i);
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                if (i & 1) d1 = mfma16(wf[i], __builtin_bit_cast(half8_t, bq[i % AHEAD]), d1);
                else d = mfma16(wf[i], __builtin_bit_cast(half8_t, bq[i % AHEAD]), d);
                if (i + AHEAD < 32) bq[i % AHEAD] = frag(i + AHEAD);
            }
            // the order above IS the schedule: left alone the scheduler sinks every read next to its MFMA (lowest register pressure) and the LDS latency shows 16 times
            __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i + AHEAD < 32) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) d[v] += d1[v];
            if constexpr (C == 64) {  // K halves meet in the LDS
                float4* part = reinterpret_cast<float4*>(lds + G::NBUF * G::BUF + ((g & 1) * 4 + nt) * 4096);
                if (kh == 1) {
#pragma unroll
                    for (int v4 = 0; v4 < 4; ++v4) part[v4 * 64 + lane] = float4{d[4 * v4], d[4 * v4 + 1], d[4 * v4 + 2], d[4 * v4 + 3]};
                }
                __syncthreads();
                if (kh == 1) continue;
#pragma unroll
                for (int v4 = 0; v4 < 4; ++v4) {
                    const float4 q = part[v4 * 64 + lane];
                    d[4 * v4] += q.x; d[4 * v4 + 1] += q.y; d[4 * v4 + 2] += q.z; d[4 * v4 + 3] += q.w;
                }
            }
            const int yo = t.oy0 + r, xo = t.ox0 + (p & 15);
            half8_t o0, o1;
#pragma unroll
            for (int v = 0; v < 8; ++v) { o0[v] = (_Float16)d[v]; o1[v] = (_Float16)d[8 + v]; }
            __half* dst = out + (((size_t)t.b * Ho + yo) * Wo + xo) * CO + 32 * nt + 16 * h;
            if (whole) {  // every lane stores: exactly two stores, counted on above
                reinterpret_cast<uint4*>(dst)[0] = __builtin_bit_cast(uint4, o0);
                reinterpret_cast<uint4*>(dst)[1] = __builtin_bit_cast(uint4, o1);
            } else if (yo < Ho && xo < Wo) {
                reinterpret_cast<uint4*>(dst)[0] = __builtin_bit_cast(uint4, o0);
                reinterpret_cast<uint4*>(dst)[1] = __builtin_bit_cast(uint4, o1);
            }
        }
        plain_stores = whole ? (plain_stores < G::NBUF - 1 ? plain_stores + 1 : plain_stores) : 0;
        if (tile + gridDim.x >= total) break;
        tile += gridDim.x;
        walk.advance();
        slot = slot + 1 == G::NBUF ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the look-ahead loads of tiles past the end still target this workgroup's LDS
}

// ---- uint8 frame -> conv_in output (MST_Plus_Plus.conv_in :275, 3 -> 31 channels, 3x3, zero padding 1) in one kernel --------------------
// Replaces: uint8 -> float32 / 255 -> NCHW -> reflect pad to the predictor's stride (predict_torch.py:171-183) -> float16 -> channels-last
// copy -> a MIOpen convolution (five elementwise launches and a library conv: ~1 ms per 4K frame).  A thread owns one output pixel: its 27
// inputs (float16-rounded like autocast's conv input, reflect-padded coordinates, zeros beyond the padded frame) stay in registers
// and meet the 27 x 32 weights as wave-uniform scalar operands; 32 float32 accumulators, one 64-byte row out.
__global__ __launch_bounds__(256) void k_mst_conv_in_u8(const uint8_t* __restrict__ frame /*[H][W][3]*/, const float* __restrict__ w /*[27][32]: k = (ky*3+kx)*3+c*/,
                                                        __half* __restrict__ out /*[Hp][Wp][32]*/, int H, int W, int pt, int pl, int Hp, int Wp) {
    __shared__ __align__(16) float wl[27 * 32];  // wave-uniform 16-byte LDS reads (broadcasts): as scalar operands the 864 weights spill the SGPR file
    for (int i = threadIdx.x; i < 27 * 32; i += 256) wl[i] = w[i];
    __syncthreads();
    const size_t total = (size_t)Hp * Wp;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / Wp), x = (int)(i - (size_t)y * Wp);
        float in[27];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            const bool ok = yy >= 0 && yy < Hp && xx >= 0 && xx < Wp;  // the conv's zero padding applies to the PADDED frame
            int sy = (ok ? yy : 0) - pt, sx = (ok ? xx : 0) - pl;       // reflect (no edge repeat) back into the frame
            sy = sy < 0 ? -sy : (sy >= H ? 2 * (H - 1) - sy : sy);
            sx = sx < 0 ? -sx : (sx >= W ? 2 * (W - 1) - sx : sx);
            const uint8_t* q = frame + ((size_t)sy * W + sx) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) in[3 * t + c] = ok ? (float)(_Float16)((float)q[c] / 255.0f) : 0.f;
        }
        uint4* dst = reinterpret_cast<uint4*>(out + i * 32);
#pragma unroll 1
        for (int v = 0; v < 4; ++v) {  // 8 output channels at a time: a bounded set of weights in flight (unrolled, the compiler hoists all 864)
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
            for (int k = 0; k < 27; ++k) {
                const float4 wa = *reinterpret_cast<const float4*>(wl + k * 32 + 8 * v), wb = *reinterpret_cast<const float4*>(wl + k * 32 + 8 * v + 4);
                acc[0] = __builtin_fmaf(in[k], wa.x, acc[0]); acc[1] = __builtin_fmaf(in[k], wa.y, acc[1]);
                acc[2] = __builtin_fmaf(in[k], wa.z, acc[2]); acc[3] = __builtin_fmaf(in[k], wa.w, acc[3]);
                acc[4] = __builtin_fmaf(in[k], wb.x, acc[4]); acc[5] = __builtin_fmaf(in[k], wb.y, acc[5]);
                acc[6] = __builtin_fmaf(in[k], wb.z, acc[6]); acc[7] = __builtin_fmaf(in[k], wb.w, acc[7]);
            }
            half8_t o8;
#pragma unroll
            for (int j = 0; j < 8; ++j) o8[j] = (_Float16)acc[j];
            dst[v] = __builtin_bit_cast(uint4, o8);
        }
    }
}

// The same on the matrix cores: K = 27 (+ 5 zero slots) x N = 32 is two v_mfma_f32_32x32x16_f16 per 32 pixels.  The inputs are float16
// already (autocast's conv input) and the weights are the model's float16 values, so the products are exact and only the float32 summation
// order differs from the scalar form above.  The K order of a product is free: lane half h = 0 carries taps 0-4 (15 slots + a zero), h = 1
// taps 5-8 (12 slots + four zeros), so a lane reads FIVE (four) source pixels, three bytes each, whose offsets from the output pixel are
// fixed per lane -- not 16 separately addressed bytes.  float16(u8 / 255) == float16(u8 * (1 / 255)) for all 256 values (checked
// exhaustively): a multiplication, not 27 IEEE divisions per pixel.  The A fragments are built once per thread from the 27 x 32 table in the
// same slot order (output-channel rows permuted so that the lane ends up with 16 contiguous channels: pack_fragments16's order).  1,130
// vector instructions per pixel before, bound by them; now by its 64 B/px of output.
__global__ __launch_bounds__(256) void k_mst_conv_in_u8_mfma(const uint8_t* __restrict__ frame /*[H][W][3]*/, const float* __restrict__ w /*[27][32]*/,
                                                             __half* __restrict__ out /*[Hp][Wp][32]*/, int H, int W, int pt, int pl, int Hp, int Wp) {
    const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
    const int col = 16 * ((p % 8) / 4) + 4 * (p / 8) + p % 4;  // output channel of this lane's A row
    const int tap0 = 5 * h, ntap = 5 - h;                      // this lane half's taps: [tap0, tap0 + ntap)
    half8_t af[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = 8 * s + j, k = 3 * tap0 + q;  // slot q of this half = (tap tap0 + q / 3, channel q % 3)
            af[s][j] = q < 3 * ntap ? (_Float16)w[k * 32 + col] : (_Float16)0.f;
        }
    int dy[5], dx[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int tap = tap0 + (t < ntap ? t : 0);
        dy[t] = tap / 3 - 1; dx[t] = tap % 3 - 1;
    }
    const size_t total = (size_t)Hp * Wp, ngroups = (total + 31) / 32;
    const float inv255 = 1.0f / 255.0f;
    for (size_t g = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); g < ngroups; g += (size_t)gridDim.x * 4) {
        const size_t i = g * 32 + p, ic = i < total ? i : total - 1;
        const int y = (int)(ic / Wp), x = (int)(ic - (size_t)y * Wp);
        float v[16];
        v[15] = 0.f;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int yy = y + dy[t], xx = x + dx[t];
            const bool ok = t < ntap && yy >= 0 && yy < Hp && xx >= 0 && xx < Wp;  // the conv's zero padding applies to the PADDED frame
            int sy = (ok ? yy : 0) - pt, sx = (ok ? xx : 0) - pl;                  // reflect (no edge repeat) back into the frame
            sy = sy < 0 ? -sy : (sy >= H ? 2 * (H - 1) - sy : sy);
            sx = sx < 0 ? -sx : (sx >= W ? 2 * (W - 1) - sx : sx);
            const uint8_t* q = frame + ((size_t)sy * W + sx) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[3 * t + c] = ok ? (float)q[c] * inv255 : 0.f;
        }
        const half8_t b0 = __builtin_bit_cast(half8_t, uint4{pack_f16(v[0], v[1]), pack_f16(v[2], v[3]), pack_f16(v[4], v[5]), pack_f16(v[6], v[7])});
        const half8_t b1 = __builtin_bit_cast(half8_t, uint4{pack_f16(v[8], v[9]), pack_f16(v[10], v[11]), pack_f16(v[12], v[13]), pack_f16(v[14], v[15])});
        float16_t d;
#pragma unroll
        for (int vv = 0; vv < 16; ++vv) d[vv] = 0.f;
        d = mfma16(af[0], b0, d);
        d = mfma16(af[1], b1, d);
        if (i < total) {
            unsigned o[8];
#pragma unroll
            for (int vv = 0; vv < 8; ++vv) o[vv] = pack_f16(d[2 * vv], d[2 * vv + 1]);
            reinterpret_cast<uint4*>(out + i * 32 + 16 * h)[0] = uint4{o[0], o[1], o[2], o[3]};
            reinterpret_cast<uint4*>(out + i * 32 + 16 * h)[1] = uint4{o[4], o[5], o[6], o[7]};
        }
    }
}

}  // namespace

static int ffn_fused_launch(avx_ctx* ctx, const void* x, const float* gamma, const float* beta, float eps, const void* w1pack, const void* taps_9xhid, const void* dwpack,
                            const void* w2pack, void* out, int B, int H, int W, int C, void* stream);

// What the matrix-pipe kernels expect of their weights (csrc/mst_common.h, prescaled GELU): avx_mst_ffn_fused_mx takes W1 / s, the depthwise weights as they are
// and W2 * s; avx_mst_attn_tail_mx takes pos_emb's first depthwise weights / s and the second * s.  s = 4 (1 when built with -DAVX_GELU_PRE=0).
extern "C" float avx_mst_gelu_prescale(void) { return kGeluPrescale; }

extern "C" int avx_mst_ffn_fused(avx_ctx* ctx, const void* x, const float* gamma, const float* beta, float eps, const void* w1pack, const void* taps_9xhid,
                                 const void* w2pack, void* out, int B, int H, int W, int C, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, taps_9xhid, "avx_mst_ffn_fused: NULL pointer");
    return ffn_fused_launch(ctx, x, gamma, beta, eps, w1pack, taps_9xhid, nullptr, w2pack, out, B, H, W, C, stream);
}

// The same with the depthwise conv on the matrix pipe: dwpack = its weights as v_mfma_f32_16x16x32_f16 A fragments (ml/mst_plus_plus.py::pack_dw_mfma).
extern "C" int avx_mst_ffn_fused_mx(avx_ctx* ctx, const void* x, const float* gamma, const float* beta, float eps, const void* w1pack, const void* dwpack,
                                    const void* w2pack, void* out, int B, int H, int W, int C, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, dwpack && (((uintptr_t)dwpack) & 15u) == 0, "avx_mst_ffn_fused_mx: dwpack NULL or not 16-byte aligned");
    return ffn_fused_launch(ctx, x, gamma, beta, eps, w1pack, nullptr, dwpack, w2pack, out, B, H, W, C, stream);
}

static int ffn_fused_launch(avx_ctx* ctx, const void* x, const float* gamma, const float* beta, float eps, const void* w1pack, const void* taps_9xhid, const void* dwpack,
                            const void* w2pack, void* out, int B, int H, int W, int C, void* stream) {
    AVX_REQUIRE(ctx, x && gamma && beta && w1pack && w2pack && out && B > 0 && H > 0 && W > 0, "avx_mst_ffn_fused: NULL pointer or empty tensor");
    AVX_REQUIRE(ctx, C == 32 || C == 64 || C == 128, "avx_mst_ffn_fused: C=%d (32, 64 or 128: 31-channel groups stored 32 wide)", C);
    AVX_REQUIRE(ctx, ((((uintptr_t)x | (uintptr_t)w1pack | (uintptr_t)w2pack | (uintptr_t)out | (uintptr_t)taps_9xhid)) & 15u) == 0,
                "avx_mst_ffn_fused: pointers must be 16-byte aligned");
    AVX_REQUIRE(ctx, x != out, "avx_mst_ffn_fused: neighbouring tiles read each other's halo: the output cannot be the input");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    const long total = (long)B * ((H + kFfnTH - 1) / kFfnTH) * ((W + TS - 1) / TS);
    const char* hp = getenv("AVX_MST_FFN_HPASS");  // A/B: 128 = one 8-wave workgroup per CU, 64 (C = 32 only) = two
    const bool small = C == 32 && !(hp && atoi(hp) == 128);
    const int hpass = (small || C == 128) ? 64 : 128, hpitch = hpass * 2 + 16, rpitch = (HS * hpitch + 255) / 256 * 256;
    const long cap = (long)ctx->num_cus * (small ? 2 : 1);
    const bool nw16 = C == 64 && dwpack && !(getenv("AVX_MST_FFN_NW") && atoi(getenv("AVX_MST_FFN_NW")) == 8);  // AVX_MST_FFN_NW=8: A/B, the round-2 split
    const size_t lds = (size_t)kFfnNGRP * 32 * (C * 2 + 16) + (size_t)(kFfnTH + 2) * rpitch + (dwpack ? 0 : (size_t)9 * 4 * C * 2) + sizeof(float) * 2 * C + (nw16 ? (size_t)(C / 32) * (hpass / 16) * 1024 : 0);
    const dim3 grid((unsigned)(total < cap ? total : cap));
#define AVX_FFN1(CV, HP, MW, DW, NWV)                                                                                                            \
    {                                                                                                                                            \
        auto k = k_mst_ffn_fused<CV, HP, MW, DW, false, NWV>;                                                                                    \
        AVX_HIP(ctx, hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                 \
        hipLaunchKernelGGL(k, grid, dim3(64 * NWV), lds, s, (const __half*)x, gamma, beta, eps, (const uint4*)w1pack, (const __half*)taps_9xhid, \
                           (const uint4*)w2pack, (__half*)out, B, H, W, (const uint4*)dwpack, (unsigned long long*)nullptr);                     \
    }
#define AVX_FFN(CV, HP, MW) { if (dwpack) AVX_FFN1(CV, HP, MW, true, 8) else AVX_FFN1(CV, HP, MW, false, 8) }
    if (dwpack && getenv("AVX_FFN_STAMPS") && ((C == 32 && small) || C == 64 || C == 128)) {  // diagnostic: per-segment cycles of every wave, summed and printed
        unsigned long long* d_st = nullptr;
        const int nwv = nw16 ? 16 : 8;
        const size_t n = (size_t)grid.x * nwv * 8;
        AVX_HIP(ctx, hipMalloc((void**)&d_st, n * sizeof(unsigned long long)));
#define AVX_FFN_ST(CV, HP, MW, NWV)                                                                                                              \
    {                                                                                                                                            \
        auto k = k_mst_ffn_fused<CV, HP, MW, true, true, NWV>;                                                                                   \
        AVX_HIP(ctx, hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                 \
        hipLaunchKernelGGL(k, grid, dim3(64 * NWV), lds, s, (const __half*)x, gamma, beta, eps, (const uint4*)w1pack, (const __half*)taps_9xhid, \
                           (const uint4*)w2pack, (__half*)out, B, H, W, (const uint4*)dwpack, d_st);                                             \
    }
        if (C == 32) AVX_FFN_ST(32, 64, 4, 8) else if (nw16) AVX_FFN_ST(64, 128, 1, 16) else if (C == 64) AVX_FFN_ST(64, 128, 2, 8) else AVX_FFN_ST(128, 64, 2, 8)
#undef AVX_FFN_ST
        AVX_HIP(ctx, hipStreamSynchronize(s));
        unsigned long long* h_st = (unsigned long long*)malloc(n * sizeof(unsigned long long));
        AVX_HIP(ctx, hipMemcpy(h_st, d_st, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double tot[8] = {0}, w0[8] = {0}, w7[8] = {0};
        for (size_t i = 0; i < n; ++i) { tot[i & 7] += (double)h_st[i]; if (((i >> 3) % nwv) == 0) w0[i & 7] += (double)h_st[i]; if (((i >> 3) % nwv) == (size_t)nwv - 1) w7[i & 7] += (double)h_st[i]; }
        double all = 0;
        for (int k2 = 0; k2 < 8; ++k2) all += tot[k2];
        fprintf(stderr, "[ffn stamps C=%d] %u blocks x %d waves, mean cycles per wave %.0f; share per segment (all waves | first wave | last wave): ", C, grid.x, nwv, all / (grid.x * (double)nwv));
        const char* nm[8] = {"bar1", "phase1", "bar2", "phase2a", "bar3", "phase2b", "epilogue", "ln+fetch"};
        for (int k2 = 0; k2 < 8; ++k2) fprintf(stderr, "%s %.1f%% | %.1f%% | %.1f%%  ", nm[k2], 100 * tot[k2] / all, 100 * w0[k2] * nwv / all, 100 * w7[k2] * nwv / all);
        fprintf(stderr, "\n");
        free(h_st);
        (void)hipFree(d_st);
    } else if (C == 32 && small) AVX_FFN(32, 64, 4) else if (C == 32) AVX_FFN(32, 128, 2)
    else if (nw16) AVX_FFN1(64, 128, 1, true, 16)
    else if (C == 64) AVX_FFN(64, 128, 2) else AVX_FFN(128, 64, 2)
#undef AVX_FFN
#undef AVX_FFN1
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

extern "C" int avx_mst_attn_tail(avx_ctx* ctx, const void* v, const void* x, const void* mpack16, const void* taps1_9xc, const void* taps2_9xc, const float* bias,
                                 void* out, int B, int H, int W, int C, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, v && x && mpack16 && taps1_9xc && taps2_9xc && out && B > 0 && H > 0 && W > 0, "avx_mst_attn_tail: NULL pointer or empty tensor");
    AVX_REQUIRE(ctx, C == 32 || C == 64, "avx_mst_attn_tail: C=%d (32 or 64)", C);
    AVX_REQUIRE(ctx, ((((uintptr_t)v | (uintptr_t)x | (uintptr_t)mpack16 | (uintptr_t)out | (uintptr_t)taps1_9xc | (uintptr_t)taps2_9xc)) & 15u) == 0,
                "avx_mst_attn_tail: pointers must be 16-byte aligned");
    AVX_REQUIRE(ctx, v != out, "avx_mst_attn_tail: tiles read their neighbours' rows of v: the output cannot be v (it may be x)");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    if (C == 32) return launch_attn_tail<32, 4>(ctx, v, x, mpack16, taps1_9xc, taps2_9xc, bias, out, B, H, W, s, nullptr);
    return launch_attn_tail<64, 2>(ctx, v, x, mpack16, taps1_9xc, taps2_9xc, bias, out, B, H, W, s, nullptr);
}

// The same with v formed inside the kernel: v = float16(x W_v^T) (MS_MSA.to_v :96, no bias) on each tile's halo region, wvpack16 = W_v in
// pack_fragments16(transposed) order.  The Gram pass (avx_mst_qkv_gram with v_out = NULL) then only reads x, and v never exists in HBM.
extern "C" int avx_mst_attn_tail_x(avx_ctx* ctx, const void* x, const void* wvpack16, const void* mpack16, const void* taps1_9xc, const void* taps2_9xc,
                                   const float* bias, void* out, int B, int H, int W, int C, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, x && wvpack16 && mpack16 && taps1_9xc && taps2_9xc && out && B > 0 && H > 0 && W > 0, "avx_mst_attn_tail_x: NULL pointer or empty tensor");
    AVX_REQUIRE(ctx, C == 32 || C == 64, "avx_mst_attn_tail_x: C=%d (32 or 64)", C);
    AVX_REQUIRE(ctx, ((((uintptr_t)x | (uintptr_t)wvpack16 | (uintptr_t)mpack16 | (uintptr_t)out | (uintptr_t)taps1_9xc | (uintptr_t)taps2_9xc)) & 15u) == 0,
                "avx_mst_attn_tail_x: pointers must be 16-byte aligned");
    AVX_REQUIRE(ctx, x != out, "avx_mst_attn_tail_x: tiles read their neighbours' rows of x: the output cannot be x");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    if (C == 32) return launch_attn_tail<32, 4>(ctx, x, x, mpack16, taps1_9xc, taps2_9xc, bias, out, B, H, W, s, wvpack16);
    return launch_attn_tail<64, 2>(ctx, x, x, mpack16, taps1_9xc, taps2_9xc, bias, out, B, H, W, s, wvpack16);
}

// The tail with both depthwise convs and the projection on the matrix pipe (k_mst_attn_tail_mx): dw1pack / dw2pack = pos_emb's two depthwise
// weights as pack_dw_mfma fragments, mpack_mx = avx_mst_attn_pack_mx's output.  out != x.
extern "C" int avx_mst_attn_tail_mx(avx_ctx* ctx, const void* x, const void* wvpack16, const void* mpack_mx, const void* dw1pack, const void* dw2pack,
                                    const float* bias, void* out, int B, int H, int W, int C, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, x && wvpack16 && mpack_mx && dw1pack && dw2pack && out && B > 0 && H > 0 && W > 0, "avx_mst_attn_tail_mx: NULL pointer or empty tensor");
    AVX_REQUIRE(ctx, C == 32 || C == 64 || C == 128, "avx_mst_attn_tail_mx: C=%d (32, 64 or 128)", C);
    AVX_REQUIRE(ctx, ((((uintptr_t)x | (uintptr_t)wvpack16 | (uintptr_t)mpack_mx | (uintptr_t)out | (uintptr_t)dw1pack | (uintptr_t)dw2pack)) & 15u) == 0,
                "avx_mst_attn_tail_mx: pointers must be 16-byte aligned");
    AVX_REQUIRE(ctx, x != out, "avx_mst_attn_tail_mx: tiles read their neighbours' rows of x: the output cannot be x");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    if (C == 32) return launch_attn_tail_mx<32, 16, 4>(ctx, x, wvpack16, mpack_mx, dw1pack, dw2pack, bias, out, B, H, W, s);
    if (C == 128) return launch_attn_tail_mx<128, 8, 2>(ctx, x, wvpack16, mpack_mx, dw1pack, dw2pack, bias, out, B, H, W, s);  // 14 x 8 tiles: 137 KB of LDS, two octets per wave
    // 16 waves per workgroup (AVX_MST_TAIL_NW=16) measured no faster than 8 at C = 64 (258 vs 257 us per 4K launch: the kernel is not bound by occupancy): 8 it stays
    if (getenv("AVX_MST_TAIL_NW") && atoi(getenv("AVX_MST_TAIL_NW")) == 16) return launch_attn_tail_mx<64, 16, 1, 16>(ctx, x, wvpack16, mpack_mx, dw1pack, dw2pack, bias, out, B, H, W, s);
    return launch_attn_tail_mx<64, 16, 2>(ctx, x, wvpack16, mpack_mx, dw1pack, dw2pack, bias, out, B, H, W, s);
}

extern "C" int avx_mst_conv3x3_lds(avx_ctx* ctx, const void* x, const void* wpack16, const void* add, void* out, int B, int H, int W, int C, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, x && wpack16 && out && B > 0 && H > 0 && W > 0, "avx_mst_conv3x3_lds: NULL pointer or empty tensor");
    AVX_REQUIRE(ctx, C == 32, "avx_mst_conv3x3_lds: C=%d (32: the 31-channel full-resolution convs)", C);
    AVX_REQUIRE(ctx, ((((uintptr_t)x | (uintptr_t)wpack16 | (uintptr_t)add | (uintptr_t)out)) & 15u) == 0, "avx_mst_conv3x3_lds: pointers must be 16-byte aligned");
    AVX_REQUIRE(ctx, x != out, "avx_mst_conv3x3_lds: in-place convolution is not possible");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    const long total = (long)B * ((H + TS - 1) / TS) * ((W + TS - 1) / TS);
    const long cap = (long)ctx->num_cus * 3;
    hipLaunchKernelGGL((k_mst_conv3x3_lds<6, false>), dim3((unsigned)(total < cap ? total : cap)), dim3(kFT), 0, s, (const __half*)x, (const uint4*)wpack16, (const __half*)add,
                       (__half*)out, B, H, W, (const uint4*)nullptr, (float*)nullptr);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

// conv_out + x with the spectral integration as its epilogue (k_mst_conv3x3_lds<., false, true>): the 31-band cube is never written.  weights_host: 3 x 32 (K x bands,
// illuminant folded in, band 31 = 0); planes_out: 3 x Hc x Wc float32 of the frame cropped at (crop_t, crop_l); partials_out: *n_partials x 3 records {float min, max; double sum}
// (16 bytes each; at most 3 x CUs records) for avx_honeybee_u8 (source 2).  One frame per call.
extern "C" int avx_mst_conv3x3_lds_spectral(avx_ctx* ctx, const void* x, const void* wpack16, const void* add, int H, int W, int C, const float* weights_host, int crop_t,
                                            int crop_l, int Hc, int Wc, float* planes_out, void* partials_out, int* n_partials, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, x && wpack16 && weights_host && planes_out && partials_out && n_partials && H > 0 && W > 0, "avx_mst_conv3x3_lds_spectral: NULL pointer or empty tensor");
    AVX_REQUIRE(ctx, C == 32, "avx_mst_conv3x3_lds_spectral: C=%d (32: the 31-band cube as stored)", C);
    AVX_REQUIRE(ctx, crop_t >= 0 && crop_l >= 0 && Hc > 0 && Wc > 0 && crop_t + Hc <= H && crop_l + Wc <= W, "avx_mst_conv3x3_lds_spectral: the crop window must lie inside the frame");
    AVX_REQUIRE(ctx, ((((uintptr_t)x | (uintptr_t)wpack16 | (uintptr_t)add | (uintptr_t)partials_out)) & 15u) == 0, "avx_mst_conv3x3_lds_spectral: pointers must be 16-byte aligned");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    avx_ws* ws = avx_workspace(ctx, s);
    if (!ws) return AVX_ERR_NOMEM;
    float wT[32 * 4];  // band-major, padded to 4: wave-uniform scalar loads in the kernel
    for (int b = 0; b < 32; ++b)
        for (int k = 0; k < 4; ++k) wT[b * 4 + k] = k < 3 ? weights_host[(size_t)k * 32 + b] : 0.0f;
    float* dwT = nullptr;
    int rc = avx_const_upload(ctx, ws, 2, wT, sizeof(wT), s, (void**)&dwT);
    if (rc) return rc;
    const long total = (long)((H + TS - 1) / TS) * ((W + TS - 1) / TS);
    const long cap = (long)ctx->num_cus * 2;  // 2 workgroups per CU: the epilogue's registers cost the third one's (as the Gram epilogue does)
    const long blocks = total < cap ? total : cap;
    *n_partials = (int)blocks;
    hipLaunchKernelGGL((k_mst_conv3x3_lds<4, false, true>), dim3((unsigned)blocks), dim3(kFT), 0, s, (const __half*)x, (const uint4*)wpack16, (const __half*)add, (__half*)nullptr, 1, H, W,
                       (const uint4*)nullptr, (float*)nullptr, (const float*)dwT, planes_out, (SpecStat3*)partials_out, crop_t, crop_l, Hc, Wc);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

// avx_mst_conv3x3_lds followed by avx_mst_qkv_gram16 (v_out = NULL) on its output, in one pass (B = 1: the Gram matrix is per frame)
extern "C" int avx_mst_conv3x3_lds_gram(avx_ctx* ctx, const void* x, const void* wpack16, const void* add, void* out, int H, int W, int C, const void* wqk16, float* gram,
                                        float* nq, float* nk, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, x && wpack16 && out && wqk16 && gram && nq && nk && H > 0 && W > 0, "avx_mst_conv3x3_lds_gram: NULL pointer or empty tensor");
    AVX_REQUIRE(ctx, C == 32, "avx_mst_conv3x3_lds_gram: C=%d (32: the 31-channel full-resolution convs)", C);
    AVX_REQUIRE(ctx, ((((uintptr_t)x | (uintptr_t)wpack16 | (uintptr_t)add | (uintptr_t)out | (uintptr_t)wqk16)) & 15u) == 0, "avx_mst_conv3x3_lds_gram: pointers must be 16-byte aligned");
    AVX_REQUIRE(ctx, x != out, "avx_mst_conv3x3_lds_gram: in-place convolution is not possible");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    avx_ws* ws = avx_workspace(ctx, s);
    if (!ws) return AVX_ERR_NOMEM;
    const long total = (long)((H + TS - 1) / TS) * ((W + TS - 1) / TS);
    const long cap = (long)ctx->num_cus * 2;  // 2 workgroups per CU: the epilogue's accumulators cost the third one's registers
    const long blocks = total < cap ? total : cap;
    int rc = avx_ensure_scratch(ctx, ws, sizeof(float) * (size_t)blocks * 34 * 32);
    if (rc) return rc;
    float* partial = (float*)ws->d_scratch;
    hipLaunchKernelGGL((k_mst_conv3x3_lds<4, true>), dim3((unsigned)blocks), dim3(kFT), 0, s, (const __half*)x, (const uint4*)wpack16, (const __half*)add, (__half*)out, 1, H, W,
                       (const uint4*)wqk16, partial);
    AVX_HIP(ctx, hipGetLastError());
    return avx_mst_qkv_final_launch(ctx, partial, (int)blocks, 1, gram, nq, nk, s);
}

extern "C" int avx_mst_conv_in_u8(avx_ctx* ctx, const uint8_t* frame_hwc, int H, int W, int pad_top, int pad_bottom, int pad_left, int pad_right,
                                  const float* w_27x32_dev, void* out, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, frame_hwc && w_27x32_dev && out && H > 1 && W > 1, "avx_mst_conv_in_u8: NULL pointer or a frame too small to reflect");
    AVX_REQUIRE(ctx, pad_top >= 0 && pad_bottom >= 0 && pad_left >= 0 && pad_right >= 0 && pad_top < H && pad_bottom < H && pad_left < W && pad_right < W,
                "avx_mst_conv_in_u8: reflect padding must be smaller than the frame");
    AVX_REQUIRE(ctx, ((uintptr_t)out & 15u) == 0, "avx_mst_conv_in_u8: out must be 16-byte aligned");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    const int Hp = H + pad_top + pad_bottom, Wp = W + pad_left + pad_right;
    const size_t total = (size_t)Hp * Wp;
    const size_t want = (total + 255) / 256, cap = (size_t)ctx->num_cus * 16;
    if (getenv("AVX_MST_CONV_IN_VALU"))  // A/B: the scalar-FMA form
        hipLaunchKernelGGL(k_mst_conv_in_u8, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, s, frame_hwc, w_27x32_dev, (__half*)out, H, W, pad_top, pad_left, Hp, Wp);
    else {
        const size_t wantm = (total + 127) / 128;  // 4 waves x 32 pixels per workgroup and trip
        hipLaunchKernelGGL(k_mst_conv_in_u8_mfma, dim3((unsigned)(wantm < cap ? wantm : cap)), dim3(256), 0, s, frame_hwc, w_27x32_dev, (__half*)out, H, W, pad_top,
                           pad_left, Hp, Wp);
    }
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

extern "C" int avx_mst_down4x4(avx_ctx* ctx, const void* x, const void* wpack16, void* out, int B, int H, int W, int C, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, x && wpack16 && out && B > 0 && H > 0 && W > 0, "avx_mst_down4x4: NULL pointer or empty tensor");
    AVX_REQUIRE(ctx, C == 32 || C == 64, "avx_mst_down4x4: C=%d (32 or 64: the encoder steps 31 -> 62 and 62 -> 124 channels as stored)", C);
    AVX_REQUIRE(ctx, H % 2 == 0 && W % 2 == 0, "avx_mst_down4x4: H and W must be even (the model pads frames to multiples of 8)");
    AVX_REQUIRE(ctx, (long)B * H * W * C * 2 < (1l << 31), "avx_mst_down4x4: input larger than 2 GiB");
    AVX_REQUIRE(ctx, ((((uintptr_t)x | (uintptr_t)wpack16 | (uintptr_t)out)) & 15u) == 0, "avx_mst_down4x4: pointers must be 16-byte aligned");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    const long cap = (long)ctx->num_cus;
    if (C == 32) {
        using G = DownGeo<32>;
        const long total = (long)B * ((H / 2 + G::OH - 1) / G::OH) * ((W / 2 + G::OW - 1) / G::OW);
        hipLaunchKernelGGL(k_mst_down4x4_dma<32>, dim3((unsigned)(total < cap ? total : cap)), dim3(kFT), 0, s, (const __half*)x, (const uint4*)wpack16, (__half*)out, (const void*)ctx->d_zero, B, H, W);
    } else {
        using G = DownGeo<64>;
        const long total = (long)B * ((H / 2 + G::OH - 1) / G::OH) * ((W / 2 + G::OW - 1) / G::OW);
        hipLaunchKernelGGL(k_mst_down4x4_dma<64>, dim3((unsigned)(total < cap ? total : cap)), dim3(kFT), 0, s, (const __half*)x, (const uint4*)wpack16, (__half*)out, (const void*)ctx->d_zero, B, H, W);
    }
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}
