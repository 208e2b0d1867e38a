// csrc/stack_up.h -- a band stack held at REDUCED size and read through cv2.resize(INTER_LINEAR) on the fly.
// classic_rgb_to_hsi_scaled (uv_helpers.py:155-183) integrates the bands at hsi_scale and resizes the K maps back to the
// frame; everything downstream (safe_norm's min/max, the 95th percentile, the barcode) only ever READS that H x W x K
// stack.  At hsi_scale 0.25 the small stack is 1/16 of it and cache resident, so the consumers recompute a pixel's K
// values (four small-stack pixels, k_resize_linear_f32's arithmetic, term for term) instead of streaming 40 B/px per pass.
#pragma once
#include <hip/hip_runtime.h>

struct avx_lin_tab { const int* ofs; const float* f; int dmax; };  // cv2 INTER_LINEAR tables of one axis (geom.hip's cache)

struct StackUp {
    const float* S; int hs, ws, H, W, K;     // small stack hs x ws x K, frame H x W
    avx_lin_tab ax, ay;
    const float2* mm;                        // per-band {min, max} of the resized stack (safe_norm), or NULL: raw values
};

constexpr int kStackKMax = 16;

// one bilinear sample of k_resize_linear_f32 (-ffp-contract=off: separate multiplies and adds, in this order)
__device__ __forceinline__ float stack_lerp(float s00, float s01, float s10, float s11, float a0, float a1, float b0, float b1, bool inner) {
    float r0, r1;
    if (inner) { r0 = s00 * a0 + s01 * a1; r1 = s10 * a0 + s11 * a1; }
    else { r0 = s00 * 1.f; r1 = s10 * 1.f; }
    return r0 * b0 + r1 * b1;
}

// a / b given r = 1.0f / b (the IEEE reciprocal): one multiply and two FMAs instead of the ~11 instructions of a float32 division.  With the correctly rounded
// reciprocal, q = RN(a r), the exact residual e = a - q b and RN(q + e r) give the correctly rounded quotient (Markstein) whenever q is a faithful rounding of
// a / b -- all but the rare operands whose first product is 2 ulp off, where the result may differ from a / b in the last place (inside every tolerance of this
// path: its consumers are a percentile and an argmax over values compared at 1e-4).  Used where one denominator serves many numerators: safe_norm's range of a
// band (uniform over the frame), the barcode's P95 and per-pixel weight sum.
__device__ __forceinline__ float div_by_r(float a, float b, float r) {
    const float q = a * r;
    const float e = __builtin_fmaf(-q, b, a);
    return __builtin_fmaf(e, r, q);
}
// safe_norm of one value (uv_helpers.py:47-53); rinv = 1.0f / (m.y - m.x)
__device__ __forceinline__ float stack_norm(float v, float2 m, float rinv) { return (m.y - m.x) < 1e-9f ? 0.f : div_by_r(v - m.x, m.y - m.x, rinv); }

// the K values of frame pixel (x, y): resized, and normalised when u.mm is set
__device__ __forceinline__ void stack_up_pixel(const StackUp& u, int x, int y, float (&v)[kStackKMax]) {
    const int sx = u.ax.ofs[x], sy0 = u.ay.ofs[y], sy1 = sy0 + 1 < u.hs ? sy0 + 1 : sy0;
    const float a1 = u.ax.f[x], a0 = 1.f - a1, b1 = u.ay.f[y], b0 = 1.f - b1;
    const bool inner = x < u.ax.dmax;
    const float* S0 = u.S + ((size_t)sy0 * u.ws + sx) * u.K;
    const float* S1 = u.S + ((size_t)sy1 * u.ws + sx) * u.K;
    const int o = inner ? u.K : 0;  // the right-hand neighbour is only read where cv2 reads it
#pragma unroll
    for (int k = 0; k < kStackKMax; ++k)
        if (k < u.K) {
            const float r = stack_lerp(S0[k], S0[o + k], S1[k], S1[o + k], a0, a1, b0, b1, inner);
            v[k] = u.mm ? stack_norm(r, u.mm[k], 1.0f / (u.mm[k].y - u.mm[k].x)) : r;
        }
}

// ---- the same through an LDS copy of the small-stack pixels under one 64 x 16 frame tile (coalesced loads; a pixel's four source
// pixels are then read from LDS: 40 scattered 4-byte global loads per pixel were the cost of the plain form) ------------------------
constexpr int kUpTW = 64, kUpTH = 16;
struct StackTile { int sx_lo, sy_lo, sw, sh; bool in_lds; };

// worst-case floats of a tile's source rectangle for this geometry (host side: the dynamic LDS size)
static inline size_t stack_tile_floats(int hs, int ws, int H, int W, int K) {
    auto ext = [](int t, int ssize, int dsize) {
        const double sc = (double)ssize / dsize;
        int e = (int)((t - 1) * sc) + 4;
        return e < ssize ? e : ssize;
    };
    return (size_t)ext(kUpTW, ws, W) * ext(kUpTH, hs, H) * K;
}

__device__ __forceinline__ StackTile stack_tile_load(const StackUp& u, int x0, int y0, float* lds, size_t cap_floats, int t, int nthreads) {
    StackTile g;
    const int x1 = (x0 + kUpTW < u.W ? x0 + kUpTW : u.W) - 1, y1 = (y0 + kUpTH < u.H ? y0 + kUpTH : u.H) - 1;
    g.sx_lo = u.ax.ofs[x0]; g.sy_lo = u.ay.ofs[y0];
    const int sx_hi = u.ax.ofs[x1] + 1 < u.ws ? u.ax.ofs[x1] + 1 : u.ws - 1, sy_hi = u.ay.ofs[y1] + 1 < u.hs ? u.ay.ofs[y1] + 1 : u.hs - 1;
    g.sw = sx_hi - g.sx_lo + 1; g.sh = sy_hi - g.sy_lo + 1;
    const int rowf = g.sw * u.K;
    g.in_lds = (size_t)rowf * g.sh <= cap_floats;  // always, by stack_tile_floats; a geometry that broke the bound reads global memory instead
    if (g.in_lds) {
        // one flat pass, four independent loads per thread in flight (a row-by-row loop waited out a memory round trip per source row)
        const int total = rowf * g.sh;
        const float inv = 1.0f / (float)rowf;
        for (int i0 = 0; i0 < total; i0 += 4 * nthreads) {
            float r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + q * nthreads + t;
                if (i < total) {
                    int rr = (int)((float)i * inv);           // i / rowf: estimate, then fix (i < 2^24)
                    rr += (rr + 1) * rowf <= i ? 1 : 0;
                    rr -= rr * rowf > i ? 1 : 0;
                    r[q] = u.S[((size_t)(g.sy_lo + rr) * u.ws + g.sx_lo) * u.K + (i - rr * rowf)];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + q * nthreads + t;
                if (i < total) lds[i] = r[q];
            }
        }
    }
    return g;
}

// K and NORM are compile-time in the tiled form: with a runtime band count the per-band predicates became scalar branches and the
// value array was copied around them (615 vector instructions per pixel instead of ~120).
template <int K, bool NORM>
__device__ __forceinline__ void stack_up_pixel_tile(const StackUp& u, const StackTile& g, const float* lds, int x, int y, const int sx, const float a1, const float2 (&mm)[K],
                                                    const float (&rinv)[K], float (&v)[K]) {
    const int sy0 = u.ay.ofs[y], sy1 = sy0 + 1 < u.hs ? sy0 + 1 : sy0;
    const float a0 = 1.f - a1, b1 = u.ay.f[y], b0 = 1.f - b1;
    const bool inner = x < u.ax.dmax;
    const int o = inner ? K : 0;
    if (g.in_lds) {
        const float* S0 = lds + ((sy0 - g.sy_lo) * g.sw + (sx - g.sx_lo)) * K;
        const float* S1 = lds + ((sy1 - g.sy_lo) * g.sw + (sx - g.sx_lo)) * K;
        if (!inner) {  // the last source column (a handful of pixels per row): one term per row, as cv2 computes it
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = stack_lerp(S0[k], 0.f, S1[k], 0.f, a0, a1, b0, b1, false);
        } else if (K % 2 == 0) {  // pixels are 8-byte aligned in the tile: two bands per LDS read
#pragma unroll
            for (int k = 0; k < K; k += 2) {
                const float2 p00 = *reinterpret_cast<const float2*>(S0 + k), p01 = *reinterpret_cast<const float2*>(S0 + K + k);
                const float2 p10 = *reinterpret_cast<const float2*>(S1 + k), p11 = *reinterpret_cast<const float2*>(S1 + K + k);
                v[k] = stack_lerp(p00.x, p01.x, p10.x, p11.x, a0, a1, b0, b1, true);
                v[k + 1 < K ? k + 1 : k] = stack_lerp(p00.y, p01.y, p10.y, p11.y, a0, a1, b0, b1, true);
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = stack_lerp(S0[k], S0[K + k], S1[k], S1[K + k], a0, a1, b0, b1, true);
        }
    } else {
        const float* S0 = u.S + ((size_t)sy0 * u.ws + sx) * K;
        const float* S1 = u.S + ((size_t)sy1 * u.ws + sx) * K;
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = stack_lerp(S0[k], S0[o + k], S1[k], S1[o + k], a0, a1, b0, b1, inner);
    }
    if (NORM) {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = stack_norm(v[k], mm[k], rinv[k]);
    }
}

// for (tile of this workgroup) { load; sync; for (pixel of this thread in the tile) body(x, y, v); sync; }  -- 256 threads, u.K == K
struct StackNoHook { __device__ __forceinline__ bool operator()() const { return false; } };
// after: called by every thread once per sweep of a tile, behind the barrier that follows the sweep's body calls (a per-tile epilogue: the candidate copy-out of the
// radix pass); returning true (the same value in every thread) makes the tile's body calls run once more (that pass's dense tiles: second sweep writes directly)
template <int K, bool NORM, typename F, typename A = StackNoHook>
__device__ __forceinline__ void stack_tiles(const StackUp& u, float* lds, size_t cap_floats, F&& body, A&& after = A()) {
    const int tiles_x = (u.W + kUpTW - 1) / kUpTW, tiles_y = (u.H + kUpTH - 1) / kUpTH;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    float2 mm[K];
#pragma unroll
    for (int k = 0; k < K; ++k) mm[k] = NORM ? u.mm[k] : make_float2(0.f, 1.f);  // wave-uniform: scalar registers
    float rinv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) rinv[k] = 1.0f / (mm[k].y - mm[k].x);  // inf / NaN for an empty range: stack_norm returns 0 there without using it
    for (int tile = blockIdx.x; tile < tiles_x * tiles_y; tile += gridDim.x) {
        const int y0 = (tile / tiles_x) * kUpTH, x0 = (tile % tiles_x) * kUpTW;
        const StackTile g = stack_tile_load(u, x0, y0, lds, cap_floats, threadIdx.x, 256);
        const int x = x0 + tx, xc = x < u.W ? x : u.W - 1;
        const int sx = u.ax.ofs[xc];
        const float a1 = u.ax.f[xc];
        __syncthreads();
        bool again;
        do {
            if (x < u.W) {
#pragma unroll
                for (int i = 0; i < kUpTH / 4; ++i) {
                    const int y = y0 + ty + 4 * i;
                    if (y < u.H) {
                        float v[K];
                        stack_up_pixel_tile<K, NORM>(u, g, lds, x, y, sx, a1, mm, rinv, v);
                        body(x, y, v);
                    }
                }
            }
            __syncthreads();
            again = after();
        } while (again);
    }
}

// band counts the tiled kernels are instantiated for (others take the materialised stack)
#define AVX_STACK_K_SWITCH(KV, CALL)                                                                  \
    switch (KV) {                                                                                     \
        case 3: { constexpr int KT = 3; CALL; break; }                                                \
        case 4: { constexpr int KT = 4; CALL; break; }                                                \
        case 6: { constexpr int KT = 6; CALL; break; }                                                \
        case 8: { constexpr int KT = 8; CALL; break; }                                                \
        case 10: { constexpr int KT = 10; CALL; break; }                                              \
        case 12: { constexpr int KT = 12; CALL; break; }                                              \
        case 16: { constexpr int KT = 16; CALL; break; }                                              \
        default: break;                                                                               \
    }
static inline bool stack_k_tiled(int K) { return K == 3 || K == 4 || K == 6 || K == 8 || K == 10 || K == 12 || K == 16; }
