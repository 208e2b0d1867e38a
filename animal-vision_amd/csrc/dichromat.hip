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
#include <cstdlib>

#include "dichromat_common.h"

using namespace avxk;

namespace {

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
            float c0, c1, c2;
            if (a.in_f32) {  // float32 frame already in [0,1]: srgb_to_linear (animal_utils.py:5-11) in float32
                const float* pf = reinterpret_cast<const float*>(a.in) + ((size_t)f * a.H * a.W + (size_t)gy * a.W + gx) * 3;
                auto eotf = [](float v) { return v <= 0.04045f ? v / 12.92f : powf((v + 0.055f) / 1.055f, 2.4f); };
                c0 = eotf(pf[0]); c1 = eotf(pf[1]); c2 = eotf(pf[2]);
                seen_gt1 = 1;  // the frame is normalised already: never the all-<=1 branch
            } else {
                const uint8_t* p = fin + ((size_t)gy * a.W + gx) * 3;
                const uint32_t b0 = p[0], b1 = p[1], b2 = p[2];
                seen_gt1 |= (b0 | b1 | b2) >> 1;
                if (DARK) {  // get_normalized_image skips /255 when max <= 1: codes 0/1 are 0.0/1.0
                    c0 = b0 ? 1.0f : 0.0f; c1 = b1 ? 1.0f : 0.0f; c2 = b2 ? 1.0f : 0.0f;
                } else {
                    c0 = lut[b0]; c1 = lut[b1]; c2 = lut[b2];
                }
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


// =================================================================================================
// variant 1: tuned tiled kernel.
//
// One 512-thread workgroup per TW x TH output tile, persistent over tiles in an XCD-aware order
// (neighbouring tiles run on the same XCD, so halo re-reads hit that XCD's L2).  Per tile:
//   phase 0  raw uint8 rows of the haloed tile -> LDS with aligned dword loads (coalesced)
//   phase 1  decode table + colour FMA chain -> three channel planes A[c] in LDS (compute type)
//   per channel c:
//     row pass     one unit = SX consecutive outputs of one row: the SX+2R window is read once with
//                  16-byte LDS reads into registers (lanes walk down rows; odd pitch in 16-byte
//                  units => conflict-free), n taps per output from registers -> plane B
//     column pass  one unit = SY rows x 8 bytes of columns: SY+2R window in registers (lanes walk
//                  along x => conflict-free 8-byte reads), folded symmetric taps, bucketed
//                  threshold quantiser -> uint8 tile in LDS
//   store    uint8 tile -> HBM with 16-byte (or 4-byte / 1-byte on odd widths) stores
// The arithmetic (operation order, fma placement) is exactly the contract at the top of this file.

__host__ __device__ constexpr int round_up_odd_multiple(int v, int m) {
    // smallest p >= v with p % m == 0 and (p / m) odd
    int q = (v + m - 1) / m;
    if ((q & 1) == 0) ++q;
    return q * m;
}

template <typename T, int R, int TW, int TH>
struct TileCfg {
    static constexpr int EPV = 16 / (int)sizeof(T);   // elements per 16-byte LDS vector
    static constexpr int SX = 2 * EPV;                // row-pass outputs per unit
    static constexpr int CX = 8 / (int)sizeof(T);     // column-pass columns per unit (8-byte reads)
    static constexpr int SY = 4;                      // column-pass rows per unit
    static constexpr int AW = TW + 2 * R, AH = TH + 2 * R;
    static constexpr int NWV = (SX + 2 * R + EPV - 1) / EPV;       // 16-byte reads per row window
    static constexpr int PA = round_up_odd_multiple((TW - SX + NWV * EPV) > AW ? (TW - SX + NWV * EPV) : AW, EPV);
    static constexpr int PB = round_up_odd_multiple(TW, EPV);
    static constexpr int RAWP = (AW * 3 + 6) & ~3;    // bytes per raw row (3 bytes of misalignment slack)
    static constexpr int ROW_UNITS = AH * (TW / SX);
    static constexpr int COL_UNITS = (TW / CX) * (TH / SY);
    static constexpr size_t off_thr = 0;                                   // T[256]
    static constexpr size_t off_lut = off_thr + 256 * sizeof(T);           // float[256]
    static constexpr size_t off_coarse = off_lut + 256 * sizeof(float);    // uint8[kCoarseTableBytes]
    static constexpr size_t off_ktab = off_coarse + kCoarseTableBytes;                  // T[64]: taps by distance from the centre
    static constexpr size_t off_A = off_ktab + 64 * sizeof(T);
    static constexpr size_t off_B = off_A + (size_t)3 * AH * PA * sizeof(T);
    static constexpr size_t off_raw = off_B + (size_t)AH * PB * sizeof(T) + 64;  // +64: window overrun of the last B/A row
    static constexpr size_t raw_bytes = (size_t)AH * RAWP > (size_t)TH * TW * 3 ? (size_t)AH * RAWP : (size_t)TH * TW * 3;
    static constexpr size_t lds_bytes = ((off_raw + raw_bytes + 15) / 16) * 16;
    static_assert(TW % SX == 0 && TW % CX == 0 && TH % SY == 0, "tile shape");
    static_assert(lds_bytes <= 160 * 1024, "tile does not fit the 160 KiB LDS of a CU");
};

// XCD-aware persistent schedule: blocks b and b+8 share an XCD (observed round-robin dispatch; used for
// L2 locality only, every tile is still visited exactly once whatever the real placement).
struct TileIter {
    int cur, end, step;
    __device__ TileIter(int total) {
        const int nb = gridDim.x, b = blockIdx.x;
        if (nb >= 8) {
            const int x = b & 7, slot = b >> 3;
            const int nx = (nb - x + 7) >> 3;  // blocks in this XCD group
            const int lo = (int)((long)total * x / 8), hi = (int)((long)total * (x + 1) / 8);
            cur = lo + slot; end = hi; step = nx;
        } else {
            cur = b; end = total; step = nb;
        }
    }
};

// STAMP = true is a DIAGNOSTIC instantiation (AVX_STAMPS=1): lane 0 of wave 0 accumulates s_memtime
// deltas per phase into a.stamps (never into an output); its run time is not representative.
template <typename T, int COLOR, bool DARK, int R, int TW, int TH, int NT, int NFIX, bool STAMP = false>
__global__ __launch_bounds__(NT) void dichromat_tiled_kernel(DichromatArgs a, Taps<T> taps, QuantCoarse qc) {
    using C = TileCfg<T, R, TW, TH>;
    using V16 = typename Vec16<T>::type;
    using V8 = typename Vec8<T>::type;
    constexpr int NWAVES = NT / 64;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* thr = reinterpret_cast<T*>(smem_raw + C::off_thr);
    float* lut = reinterpret_cast<float*>(smem_raw + C::off_lut);
    uint8_t* coarse = smem_raw + C::off_coarse;
    T* A = reinterpret_cast<T*>(smem_raw + C::off_A);
    T* Bm = reinterpret_cast<T*>(smem_raw + C::off_B);
    uint8_t* RAW = smem_raw + C::off_raw;  // raw input rows in phases 0-1, the uint8 output tile afterwards
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: row bookkeeping goes to the SALU
    for (int i = tid; i < 256; i += NT) {
        thr[i] = reinterpret_cast<const T*>(a.enc_thr)[i];
        lut[i] = a.decode_lut[i];
    }
    for (int i = tid; i < kCoarseTableBytes; i += NT) coarse[i] = i < (int)qc.n_keys ? qc.table[i] : (uint8_t)0;
    T* ktab = reinterpret_cast<T*>(smem_raw + C::off_ktab);
    if (tid <= R) ktab[tid] = taps.k[R + tid];  // Gaussian taps are symmetric: k[R-d] == k[R+d] bit for bit
    __syncthreads();
    // Taps live in VGPRs for the whole kernel (read back from LDS so that they are not scalarised:
    // 2R+1 SGPR taps + the arguments overflow the SGPR file and spill through v_readlane).
    T kd[R + 1];
#pragma unroll
    for (int d = 0; d <= R; ++d) kd[d] = ktab[d];
    const int tiles_per_frame = a.tiles_x * a.tiles_y;
    const size_t frame_bytes = (size_t)a.H * a.W * 3;
    const uint8_t* const in_end = a.in + frame_bytes * a.n_frames;
    constexpr int DPR = C::RAWP / 4;                    // dword slots per raw row
    constexpr int RPW = (C::AH + NWAVES - 1) / NWAVES;  // raw rows per wave
    constexpr int DPL = (DPR + 63) / 64;                // dwords per lane per row
    // Raw rows of one tile as aligned dwords in registers, one row per wave-iteration (row address,
    // reflection and alignment are wave-uniform).  Issued for tile t+1 before tile t is computed, so
    // the HBM/L2 latency of the loads hides under a whole tile of arithmetic.
    auto issue_raw_loads = [&](int tile, uint32_t (&rv)[RPW][DPL]) {
        const int f = tile / tiles_per_frame;
        const int t2 = tile - f * tiles_per_frame;
        const int ty = t2 / a.tiles_x, tx = t2 - ty * a.tiles_x;
        const int x0 = tx * TW, y0 = ty * TH;
        const uint8_t* fin = a.in + frame_bytes * f;
        const int xlo = x0 - R > 0 ? x0 - R : 0;
        const int xhi = x0 + TW + R < a.W ? x0 + TW + R : a.W;
        const int row_bytes = (xhi - xlo) * 3;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int ly = wave + r * NWAVES;
            const int gy = reflect101(y0 - R + (ly < C::AH ? ly : 0), a.H);
            const uint8_t* rowp = fin + ((size_t)gy * a.W + xlo) * 3;
            const uint32_t shift = (uint32_t)((uintptr_t)rowp & 3u);
            const int ndw = ly < C::AH ? ((int)shift + row_bytes + 3) >> 2 : 0;
            const uint8_t* base = rowp - shift;
#pragma unroll
            for (int k = 0; k < DPL; ++k) {
                const int d = lane + 64 * k;
                uint32_t v = 0;
                if (d < ndw) {
                    const uint8_t* p = base + (size_t)d * 4;
                    if (p + 4 <= in_end) {
                        v = *reinterpret_cast<const uint32_t*>(p);
                    } else {  // last dword of the whole batch: do not read past the allocation
                        for (int b = 0; b < 4; ++b)
                            if (p + b < in_end) v |= (uint32_t)p[b] << (8 * b);
                    }
                }
                rv[r][k] = v;
            }
        }
    };
    TileIter it(tiles_per_frame * a.n_frames);
    uint32_t rv[RPW][DPL];
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = 0;
    auto stamp = [&](int slot) {
        if constexpr (STAMP) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (slot >= 0) st_acc[slot] += t - st_t;
            st_t = t;
        }
    };
    // The all-<=1 fix-up launch touches memory only for frames whose flag stayed 0.
    auto wanted = [&](int tile) { return !DARK || a.flags[tile / tiles_per_frame] == 0u; };
    if (it.cur < it.end && wanted(it.cur)) issue_raw_loads(it.cur, rv);
    for (int tile = it.cur; tile < it.end; tile += it.step) {
        const int f = tile / tiles_per_frame;
        const bool skip = DARK && a.flags[f] != 0u;  // uniform per workgroup
        const int t2 = tile - f * tiles_per_frame;
        const int ty = t2 / a.tiles_x, tx = t2 - ty * a.tiles_x;
        const int x0 = tx * TW, y0 = ty * TH;
        const uint8_t* fin = a.in + frame_bytes * f;
        uint8_t* fout = a.out + frame_bytes * f;
        const int xlo = x0 - R > 0 ? x0 - R : 0;
        const int xhi = x0 + TW + R < a.W ? x0 + TW + R : a.W;
        const int row_bytes = (xhi - xlo) * 3;
        stamp(-1);
        // ---- phase 0: prefetched raw dwords -> LDS; then prefetch the next tile -----------------
        if (!skip) {
            uint32_t seen = 0;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int ly = wave + r * NWAVES;
                if (ly < C::AH) {
                    const int gy = reflect101(y0 - R + ly, a.H);
                    const uint32_t shift = (uint32_t)((uintptr_t)(fin + ((size_t)gy * a.W + xlo) * 3) & 3u);
                    const int ndw = ((int)shift + row_bytes + 3) >> 2;
#pragma unroll
                    for (int k = 0; k < DPL; ++k) {
                        const int d = lane + 64 * k;
                        if (d < ndw) {
                            reinterpret_cast<uint32_t*>(RAW)[ly * DPR + d] = rv[r][k];
                            if (!DARK) {
                                // bytes of this dword that belong to the row (not the <=3 bytes around it)
                                const int b0 = d * 4 - (int)shift;  // row byte index of dword byte 0
                                uint32_t m = 0xfefefefeu;
                                if (b0 < 0) m &= 0xffffffffu << (8 * (-b0));
                                if (b0 + 4 > row_bytes) m &= 0xffffffffu >> (8 * (b0 + 4 - row_bytes));
                                seen |= rv[r][k] & m;
                            }
                        }
                    }
                }
            }
            if (!DARK && seen) a.flags[f] = 1u;  // benign race: every writer stores the same value
        }
        stamp(0);
        if (tile + it.step < it.end && wanted(tile + it.step)) issue_raw_loads(tile + it.step, rv);
        if (skip) continue;
        __syncthreads();
        stamp(1);
        // ---- phase 1: decode + colour stage -> A planes (one raw row per wave-iteration) --------
        {
            const bool interior_x = (x0 - R >= 0) && (x0 + TW + R <= a.W);
            for (int ly = wave; ly < C::AH; ly += NWAVES) {
                const int gy = reflect101(y0 - R + ly, a.H);
                const uint32_t shift = (uint32_t)((uintptr_t)(fin + ((size_t)gy * a.W + xlo) * 3) & 3u);
                const uint8_t* rawrow = RAW + ly * C::RAWP + shift - xlo * 3;  // pixel gx lives at rawrow + 3*gx
                T* arow = A + (size_t)ly * C::PA;
                for (int lx = lane; lx < C::AW; lx += 64) {
                    int gx = x0 - R + lx;
                    if (!interior_x) gx = reflect101(gx, a.W);
                    const uint8_t* p = rawrow + gx * 3;
                    const uint32_t b0 = p[0], b1 = p[1], b2 = p[2];
                    float c0, c1, c2;
                    if (DARK) {
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
                        const float lm = a.alpha * l + a.one_minus_alpha * m;
                        const double dlm = (double)lm, ds = (double)s;
                        o0 = (T)__builtin_fma(ds, a.Bk[2], __builtin_fma(dlm, a.Bk[1], dlm * a.Bk[0]));
                        o1 = (T)__builtin_fma(ds, a.Bk[5], __builtin_fma(dlm, a.Bk[4], dlm * a.Bk[3]));
                        o2 = (T)__builtin_fma(ds, a.Bk[8], __builtin_fma(dlm, a.Bk[7], dlm * a.Bk[6]));
                    }
                    arow[lx] = o0;
                    arow[(size_t)C::AH * C::PA + lx] = o1;
                    arow[(size_t)2 * C::AH * C::PA + lx] = o2;
                }
            }
        }
        __syncthreads();
        stamp(2);
        uint8_t* OUT = RAW;  // raw rows are dead from here on
#pragma unroll 1
        for (int c = 0; c < 3; ++c) {
            // ---- row pass: A[c] -> Bm --------------------------------------------------------
            for (int u = tid; u < C::ROW_UNITS; u += NT) {
                const int seg = u / C::AH, row = u - seg * C::AH;
                const T* src = A + ((size_t)c * C::AH + row) * C::PA + seg * C::SX;
                T w[C::NWV * C::EPV];
#pragma unroll
                for (int v = 0; v < C::NWV; ++v) {
                    const V16 q = reinterpret_cast<const V16*>(src)[v];
                    if constexpr (C::EPV == 4) { w[4 * v] = q.x; w[4 * v + 1] = q.y; w[4 * v + 2] = q.z; w[4 * v + 3] = q.w; }
                    else { w[2 * v] = q.x; w[2 * v + 1] = q.y; }
                }
                T o[C::SX];
#pragma unroll
                for (int i = 0; i < C::SX; ++i) {
                    T s = w[i] * kd[R];  // tap 0 is at distance R from the centre
#pragma unroll
                    for (int j = 1; j <= 2 * R; ++j) s = fma_t(w[i + j], kd[j < R ? R - j : j - R], s);
                    o[i] = s;
                }
                V16* dst = reinterpret_cast<V16*>(Bm + (size_t)row * C::PB + seg * C::SX);
                if constexpr (C::EPV == 4) { dst[0] = make_float4(o[0], o[1], o[2], o[3]); dst[1] = make_float4(o[4], o[5], o[6], o[7]); }
                else { dst[0] = make_double2(o[0], o[1]); dst[1] = make_double2(o[2], o[3]); }
            }
            __syncthreads();
            stamp(3);
            // ---- column pass + quantise: Bm -> OUT bytes -------------------------------------
            for (int u = tid; u < C::COL_UNITS; u += NT) {
                const int ys = u / (TW / C::CX), cg = u - ys * (TW / C::CX);
                const T* src = Bm + (size_t)(ys * C::SY) * C::PB + cg * C::CX;
                T w[C::SY + 2 * R][C::CX];
#pragma unroll
                for (int j = 0; j < C::SY + 2 * R; ++j) {
                    const V8 q = *reinterpret_cast<const V8*>(src + (size_t)j * C::PB);
                    if constexpr (C::CX == 2) { w[j][0] = q.x; w[j][1] = q.y; }
                    else { w[j][0] = q; }
                }
#pragma unroll
                for (int i = 0; i < C::SY; ++i) {
#pragma unroll
                    for (int x = 0; x < C::CX; ++x) {
                        T s = w[i + R][x] * kd[0];
#pragma unroll
                        for (int j = 1; j <= R; ++j) s = fma_t(w[i + R + j][x] + w[i + R - j][x], kd[j], s);
                        const uint32_t q = quantize_coarse<T, NFIX>(s, thr, coarse, qc.lo_key);
                        OUT[((ys * C::SY + i) * TW + cg * C::CX + x) * 3 + c] = (uint8_t)q;
                    }
                }
            }
            __syncthreads();
            stamp(4);
        }
        // ---- store the uint8 tile ------------------------------------------------------------
        {
            const int tw = a.W - x0 < TW ? a.W - x0 : TW;   // valid columns / rows of this tile
            const int th = a.H - y0 < TH ? a.H - y0 : TH;
            const bool al16 = ((a.W & 15) == 0) && (((uintptr_t)fout & 15u) == 0) && ((TW * 3) % 16 == 0);
            const bool al4 = ((a.W & 3) == 0) && (((uintptr_t)fout & 3u) == 0) && ((TW * 3) % 4 == 0);
            if (al16 && tw == TW) {
                constexpr int VPR = TW * 3 / 16;
                for (int i = tid; i < th * VPR; i += NT) {
                    const int y = i / VPR, v = i - y * VPR;
                    const uint4 q = reinterpret_cast<const uint4*>(OUT + (size_t)y * TW * 3)[v];
                    reinterpret_cast<uint4*>(fout + ((size_t)(y0 + y) * a.W + x0) * 3)[v] = q;
                }
            } else if (al4) {
                const int dpr = tw * 3 / 4;  // W % 4 == 0 and TW % 4 == 0 => tw % 4 == 0
                for (int i = tid; i < th * dpr; i += NT) {
                    const int y = i / dpr, d = i - y * dpr;
                    const uint32_t q = reinterpret_cast<const uint32_t*>(OUT + (size_t)y * TW * 3)[d];
                    reinterpret_cast<uint32_t*>(fout + ((size_t)(y0 + y) * a.W + x0) * 3)[d] = q;
                }
            } else {
                const int bpr = tw * 3;
                for (int i = tid; i < th * bpr; i += NT) {
                    const int y = i / bpr, b = i - y * bpr;
                    fout[((size_t)(y0 + y) * a.W + x0) * 3 + b] = OUT[(size_t)y * TW * 3 + b];
                }
            }
        }
        __syncthreads();  // OUT/RAW, A, Bm are rewritten by the next tile
        stamp(5);
    }
    if constexpr (STAMP) {
        if (tid == 0 && a.stamps)
            for (int i = 0; i < 8; ++i) atomicAdd(a.stamps + i, st_acc[i]);
    }
}

template <typename T, int COLOR, int R, int TW, int TH, int NT, int NFIX>
int launch_tiled(avx_ctx* ctx, DichromatArgs& a, const avx_dichromat_desc* d, const QuantCoarse& qc, hipStream_t s) {
    using C = TileCfg<T, R, TW, TH>;
    a.TW = TW;
    a.TH = TH;
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    Taps<T> taps;
    for (int i = 0; i < AVX_MAX_KSIZE; ++i) taps.k[i] = (T)0;
    for (int i = 0; i < d->ksize; ++i) taps.k[i] = (T)d->taps_host[i];
    const size_t lds = C::lds_bytes;
    const long total = (long)a.tiles_x * a.tiles_y * a.n_frames;
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 2048 / NT) per_cu = 2048 / NT;
    if (per_cu < 1) per_cu = 1;
    long want = (long)ctx->num_cus * per_cu;
    const int grid = (int)(total < want ? total : want);
    auto kmain = dichromat_tiled_kernel<T, COLOR, false, R, TW, TH, NT, NFIX>;
    auto kdark = dichromat_tiled_kernel<T, COLOR, true, R, TW, TH, NT, NFIX>;
    if (getenv("AVX_STAMPS") && (R == 6 || R == 14 || R == 4) && TW == 64 && NT == 512) {
        // diagnostic: phase shares via in-kernel stamps, printed to stderr (never used for timing)
        if constexpr ((R == 6 || R == 14 || R == 4) && TW == 64 && NT == 512) {
            auto kst = dichromat_tiled_kernel<T, COLOR, false, R, TW, TH, NT, NFIX, true>;
            AVX_HIP(ctx, hipFuncSetAttribute((const void*)kst, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            avx_ws* wst = avx_workspace(ctx, s);
            AVX_HIP(ctx, (wst && avx_ensure_scratch(ctx, wst, 64) == AVX_OK) ? hipSuccess : hipErrorOutOfMemory);
            a.stamps = (unsigned long long*)wst->d_scratch;
            AVX_HIP(ctx, hipMemsetAsync(a.stamps, 0, 64, s));
            AVX_HIP(ctx, hipMemsetAsync(a.flags, 0, sizeof(uint32_t) * a.n_frames, s));
            hipLaunchKernelGGL(kst, dim3(grid), dim3(NT), lds, s, a, taps, qc);
            unsigned long long h[8];
            AVX_HIP(ctx, hipMemcpyAsync(h, a.stamps, 64, hipMemcpyDeviceToHost, s));
            AVX_HIP(ctx, hipStreamSynchronize(s));
            double tot = 0;
            for (int i = 0; i < 6; ++i) tot += (double)h[i];
            const double per = tot > 0 ? 100.0 / tot : 0;
            fprintf(stderr, "[avx stamps R=%d %dx%d] per-tile cycles (wave 0): total %.0f | raw->LDS %.1f%% issue+barrier %.1f%% decode %.1f%% row %.1f%% col+quant %.1f%% store %.1f%%\n",
                    R, TW, TH, tot / ((double)total / grid) / grid, h[0] * per, h[1] * per, h[2] * per, h[3] * per, h[4] * per, h[5] * per);
            a.stamps = nullptr;
        }
    }
    AVX_HIP(ctx, hipFuncSetAttribute((const void*)kmain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    AVX_HIP(ctx, hipFuncSetAttribute((const void*)kdark, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    AVX_HIP(ctx, hipMemsetAsync(a.flags, 0, sizeof(uint32_t) * a.n_frames, s));
    hipLaunchKernelGGL(kmain, dim3(grid), dim3(NT), lds, s, a, taps, qc);
    AVX_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(kdark, dim3(grid), dim3(NT), lds, s, a, taps, qc);
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
    AVX_REQUIRE(ctx, d->post_mode >= AVX_POST_NONE && d->post_mode <= AVX_POST_STREAK,
                "avx_dichromat_u8: unknown post_mode %d", d->post_mode);
    if (n_frames == 0) return AVX_OK;
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    avx_ws* ws = avx_workspace(ctx, s);
    if (!ws) return AVX_ERR_NOMEM;
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
        { const int rcu = avx_upload_row_table(ctx, ws, d->row_gain_host, sizeof(float) * (size_t)H, s); if (rcu) return rcu; }
        a.row_gain = ws->d_row_gain;
        a.row_gain_clamp = d->row_gain_clamp;
    }
    if ((size_t)n_frames > ws->flags_cap) {
        if (ws->d_flags) { AVX_HIP(ctx, hipStreamSynchronize(s)); AVX_HIP(ctx, hipFree(ws->d_flags)); }
        ws->d_flags = nullptr;
        ws->flags_cap = 0;
        size_t cap = (size_t)n_frames < 64 ? 64 : (size_t)n_frames;
        AVX_HIP(ctx, hipMalloc((void**)&ws->d_flags, sizeof(uint32_t) * cap));
        ws->flags_cap = cap;
    }
    a.flags = ws->d_flags;
    { const char* e = getenv("AVX_ABLATE"); a.ablate = e ? atoi(e) : 0; }
    a.chroma_enable = d->chroma_enable;
    a.chroma_keep = d->chroma_keep;
    a.decode_lut = ctx->d_decode_lut;
    // variant: 0 = auto (marching strip kernel, else 2-D tiled, else reference), 1 = reference kernel only,
    //          2 = 2-D tiled kernel, 3 = marching strip kernel.  All variants share one arithmetic contract.
    int variant = d->variant;
    { const char* e = getenv("AVX_VARIANT"); if (e && variant == 0) variant = atoi(e); }  // tuning only
    a.in_f32 = d->in_f32 ? 1 : 0;
    const bool tunable = !d->in_f32 && d->post_mode == AVX_POST_GAUSS && !d->chroma_enable && (((uintptr_t)in_hwc & 3u) == 0);
    const bool try_march = tunable && (variant == 0 || variant == 3);
    const bool use_tiled = tunable && (variant == 0 || variant == 2);
    const bool cat = d->color_mode == AVX_COLOR_CAT_MERGE;
    if (!cat) {
        for (int i = 0; i < 9; ++i) a.M[i] = d->matrix[i];
        a.enc_thr = ctx->d_enc_thr_f32;
    } else {
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
    }
    // auto: the marching strip kernel wherever an instantiation exists (it beats the 2-D tiled kernel for
    // every species measured); explicit variants are honoured as asked.
    if (d->post_mode == AVX_POST_STREAK) {
        AVX_REQUIRE(ctx, !cat && !d->in_f32, "avx_dichromat_u8: AVX_POST_STREAK needs AVX_COLOR_MATRIX and uint8 input");
        return avx_launch_dichromat_streak(ctx, ws, a, d, s);
    }
    if (try_march) {
        const int rc = avx_launch_dichromat_march(ctx, a, d, cat, s);
        if (rc != AVX_ERR_UNSUPPORTED) return rc;
    }
    constexpr int NF = kCoarseNFix;
    if (!cat) {
        if (use_tiled) {
            AVX_REQUIRE(ctx, ctx->coarse_n_fix[0] <= NF, "quantiser needs %d refinements, kernel built for %d", ctx->coarse_n_fix[0], NF);
            QuantCoarse qc{ctx->d_coarse_f32, ctx->coarse_lo_key[0], ctx->coarse_n_keys[0], ctx->coarse_n_fix[0]};
            switch (a.r) {
#define AVX_TILED_F32(RR) case RR: return launch_tiled<float, AVX_COLOR_MATRIX, RR, 64, 64, 512, NF>(ctx, a, d, qc, s);
                AVX_TILED_F32(1) AVX_TILED_F32(3) AVX_TILED_F32(4) AVX_TILED_F32(5) AVX_TILED_F32(6)
                AVX_TILED_F32(7) AVX_TILED_F32(8) AVX_TILED_F32(9) AVX_TILED_F32(14)
#undef AVX_TILED_F32
                default: break;  // other radii: generic-radius reference kernel
            }
        }
        return launch_simple<float, AVX_COLOR_MATRIX>(ctx, a, d, s);
    }
    if (use_tiled && a.r == 4) {
        AVX_REQUIRE(ctx, ctx->coarse_n_fix[1] <= NF, "quantiser needs %d refinements, kernel built for %d", ctx->coarse_n_fix[1], NF);
        QuantCoarse qc{ctx->d_coarse_f64, ctx->coarse_lo_key[1], ctx->coarse_n_keys[1], ctx->coarse_n_fix[1]};
        return launch_tiled<double, AVX_COLOR_CAT_MERGE, 4, 64, 48, 512, NF>(ctx, a, d, qc, s);
    }
    return launch_simple<double, AVX_COLOR_CAT_MERGE>(ctx, a, d, s);
}
