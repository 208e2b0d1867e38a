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
    const float* pin;   // PLANES instantiation: 3 x H x W float32 planes in / out (linear light, after the colour stage)
    float* pout;
};

// v2 schedule (the arithmetic is unchanged: same FMA chains, same order).  One workgroup per image row:
//   * the row lives in LDS as three PLANES (channel-major) with a 16-sample halo each side that holds the reflect-101
//     mirror, so the along-row passes index without any border logic;
//   * along-row passes: a thread owns 4 adjacent samples of one channel and reads its (4 + 2*RB)-sample window with
//     16-byte LDS loads (conflict-free), RB in {4, 8, 12, 16} picked per row from the row's radius; taps beyond the
//     radius are zero, and fma(x, 0, s) == s exactly, so the padded chain is bit-identical to the reference's;
//   * bucketed quantiser; bytes are staged in LDS and leave with 4-byte stores.
constexpr int kPad = 16;  // halo samples per side (>= the largest radius, multiple of 4)

// The planes are addressed in float4 units from the (16-byte aligned) start of the dynamic LDS block: with float pointers and
// a run-time pitch the compiler could not prove the 16-byte alignment of a window and split every float4 load into
// ds_read2_b32 / ds_read2_b64 -- at this 16-byte lane stride those are 4-way / 2-way bank conflicts (PMC: 85 % of the
// kernel's conflict cycles, which were 54 % of its LDS cycles); ds_read_b128 at that stride is conflict-free.
template <int RB>
__device__ __forceinline__ void streak_along_row(const float4* __restrict__ src4, float4* __restrict__ dst4, int W, int PW4, const float* __restrict__ td /*taps by distance, zero beyond r*/,
                                                int tid) {
    static_assert(RB % 4 == 0 && kPad % 4 == 0, "windows start on float4 boundaries");
    const int ngroups = (W + 3) / 4;
    for (int item = tid; item < 3 * ngroups; item += kST) {
        const int c = item / ngroups, xg = item - c * ngroups;
        const float4* base = src4 + (c * PW4 + (kPad - RB) / 4 + xg);
        float w[4 + 2 * RB];
#pragma unroll
        for (int q = 0; q < (4 + 2 * RB) / 4; ++q) {
            float4 v = base[q];
            // The packed FMAs below pair neighbouring samples, and every other pairing starts on an odd register: left to itself
            // the compiler fetches a second, one-sample-shifted copy of the window from LDS with ds_read2_b32 / ds_read2_b64,
            // which at this 16-byte lane stride are 4-way / 2-way bank conflicts (PMC with phases ablated: the two along-row
            // passes held 85 % of the kernel's conflict cycles, themselves 54 % of its LDS cycles).  Passing the loaded values
            // through an empty asm makes them opaque: one conflict-free ds_read_b128 per float4, pairs formed in registers.
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
            w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        }
        float o[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            float sacc = w[RB + x] * td[0];
#pragma unroll
            for (int j = 1; j <= RB; ++j) sacc = fma_t(w[RB + x + j] + w[RB + x - j], td[j], sacc);
            o[x] = sacc;
        }
        dst4[c * PW4 + kPad / 4 + xg] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// pass 1, "row" direction of the reference's W x 3 image = ACROSS the three channels (reflect-101 on a width of 3).  The radius
// is uniform per image row, so it is a template argument: every source-channel index is then a compile-time constant and the
// K1 taps sit in registers -- 3 * K1 FMAs per pixel, where the run-time form spent two LDS reads, a dynamic select and loop
// control on every tap (ablation: a third of the kernel's time).  Same sequential FMA chain, tap by tap.
__host__ __device__ constexpr int refl3(int p) {
    while (p < 0 || p >= 3) p = p < 0 ? -p : 4 - p;
    return p;
}
// `load(x, px)` produces the pixel's three linear-light channels (decode + colour stage, or a read of the float planes): the
// pass is pointwise in x, so it consumes them from registers -- no intermediate plane, no barrier in between.
template <int R1, typename Load>
__device__ __forceinline__ void streak_across(Load load, float* __restrict__ dst, int W, int PW, const float* __restrict__ t1, int tid) {
    constexpr int K1 = 2 * R1 + 1;
    float t[K1];
#pragma unroll
    for (int j = 0; j < K1; ++j) t[j] = t1[j];
    for (int x = tid; x < W; x += kST) {
        float px[3];
        load(x, px);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float sacc = px[refl3(c - R1)] * t[0];
#pragma unroll
            for (int j = 1; j < K1; ++j) sacc = fma_t(px[refl3(c - R1 + j)], t[j], sacc);
            dst[(size_t)c * PW + kPad + x] = sacc;
        }
    }
}

// mirror the interior of each plane into its halo (BORDER_REFLECT_101 on W)
__device__ __forceinline__ void streak_fill_halo(float* buf, int W, int PW, int r, int tid) {
    for (int item = tid; item < 3 * 2 * r; item += kST) {
        const int c = item / (2 * r), e = item - c * 2 * r;
        const int x = e < r ? -1 - e : W + (e - r);
        buf[(size_t)c * PW + kPad + x] = buf[(size_t)c * PW + kPad + reflect101(x, W)];
    }
}

// PLANES = true is the float-frame form (avx_streak_planes_f32): the same row passes between float32 planes, with the
// transfer functions and the chroma step left to the caller's plane program.
// PREF = true (uint8 frames whose batch size and base are multiples of 4 bytes): a row's bytes are fetched as aligned dwords,
// all of a thread's loads in flight together and ONE ROW AHEAD (they travel while the previous row is being filtered), staged
// in the LDS bytes that hold the output at the end of the row.  The plain form reads three bytes per pixel straight from
// global memory inside the decode loop: 8 dependent trips of ~1.5 us per row, a third of the row's time.
// NRAW = dwords per thread and row for that prefetch: 6 covers rows up to 6 * 256 * 4 - 8 bytes (W <= 2045), 12 up to W <= 4093.
// Three waves per SIMD (<= 168 VGPRs) so that three 51 KB workgroups really are resident per CU.
template <bool DARK, bool PLANES = false, bool PREF = false, int NRAW = 12>
__global__ __launch_bounds__(kST, 3) void dichromat_streak_kernel(DichromatArgs a, StreakArgs st, QuantCoarse qc) {
    constexpr int kRawRegs = NRAW;
    extern __shared__ __align__(16) float smem_f[];
    const int W = a.W, n3 = a.W * 3;
    const int PW = ((W + 3) / 4 * 4) + 2 * kPad;     // plane pitch (multiple of 4)
    float* buf0 = smem_f;                            // 3 planes
    float* buf1 = smem_f + (size_t)3 * PW;           // 3 planes
    const int PW4 = PW / 4;
    float4* const buf0_4 = reinterpret_cast<float4*>(smem_f);  // the same planes in float4 units (alignment visible to the compiler)
    float4* const buf1_4 = buf0_4 + 3 * PW4;
    // The row's bytes (staged input at the top of a row, quantised output at its end) live in buf0's storage: buf0 is written by
    // the first along-row pass only after the staged input has been decoded, and is dead again once the second pass has read
    // it.  Without a byte buffer of its own a workgroup needs 51 KB instead of 57 KB of LDS: three fit a CU instead of two,
    // and the kernel is bound by how many latency-limited rows are in flight.
    uint8_t* outb = reinterpret_cast<uint8_t*>(buf0);  // W*3 bytes (+ <= 8 of misalignment) <= 3 planes of floats
    __shared__ float lut[256];
    __shared__ float thr[256];
    __shared__ uint8_t coarse[kCoarseTableBytes];
    __shared__ float t1[16], td1[20], td2[20];
    const int tid = threadIdx.x;
    if constexpr (!PLANES) {
        for (int i = tid; i < 256; i += kST) { lut[i] = a.decode_lut[i]; thr[i] = reinterpret_cast<const float*>(a.enc_thr)[i]; }
        for (int i = tid; i < kCoarseTableBytes; i += kST) coarse[i] = i < (int)qc.n_keys ? qc.table[i] : (uint8_t)0;
    }
    const size_t frame_bytes = (size_t)a.H * a.W * 3;
    const int total_rows = a.n_frames * a.H;
    uint32_t rawv[kRawRegs];
    auto issue_row = [&](int g2) {  // aligned dwords of row g2 -> rawv; lanes past the row's last dword re-read it (always valid)
        const int f2 = g2 / a.H, y2 = g2 - f2 * a.H;
        const uint8_t* s2 = a.in + frame_bytes * f2 + (size_t)y2 * n3;
        const uint32_t sh = (uint32_t)((uintptr_t)s2 & 3u);
        const uint32_t* base = reinterpret_cast<const uint32_t*>(s2 - sh);
        const int last = ((int)sh + n3 - 1) >> 2;
#pragma unroll
        for (int k = 0; k < kRawRegs; ++k) { const int d = tid + k * kST; rawv[k] = base[d < last ? d : last]; }
    };
    // (the fix-up pass for all-<=1 frames touches only those frames' rows: it does not prefetch the others)
    auto wanted = [&](int g2) { return !DARK || a.flags[g2 / a.H] == 0u; };
    if constexpr (PREF) { if ((int)blockIdx.x < total_rows && wanted(blockIdx.x)) issue_row(blockIdx.x); }
    for (int gr = blockIdx.x; gr < total_rows; gr += gridDim.x) {
        const int f = gr / a.H, y = gr - f * a.H;
        __syncthreads();  // tables and buffers of the previous row are no longer read
        bool skip = false;
        if constexpr (!PLANES) skip = DARK && a.flags[f] != 0u;
        if constexpr (PREF) {
            if (!skip) {
                const int ndw = (n3 + 3 + 3) >> 2;  // covers any misalignment
#pragma unroll
                for (int k = 0; k < kRawRegs; ++k) { const int d = tid + k * kST; if (d < ndw) reinterpret_cast<uint32_t*>(outb)[d] = rawv[k]; }
            }
            if (gr + (int)gridDim.x < total_rows && wanted(gr + gridDim.x)) issue_row(gr + gridDim.x);
        }
        if (skip) continue;
        const float* rt = st.rows + (size_t)y * st.stride;
        const int k1 = (int)rt[0], k2 = (int)rt[1];
        const int r1 = k1 / 2, r2 = k2 / 2;
        if (tid < 16) t1[tid] = tid < k1 ? rt[2 + tid] : 0.f;
        if (tid < 20) { td1[tid] = tid <= r1 ? rt[2 + r1 + tid] : 0.f; td2[tid] = tid <= r2 ? rt[15 + r2 + tid] : 0.f; }
        const uint8_t* src = a.in + frame_bytes * f + (size_t)y * n3;
        uint8_t* dst = a.out + frame_bytes * f + (size_t)y * n3;
        // ---- decode + colour stage, then pass 1 "row" direction = across the 3 channels (width 3, reflect-101) -> buf1 planes ----
        uint32_t seen = 0;
        const uint8_t* rowb = PREF ? outb + ((uintptr_t)src & 3u) : src;
        auto load_px = [&](int x, float (&px)[3]) {
            if constexpr (PLANES) {
                const size_t plane = (size_t)a.H * W;
#pragma unroll
                for (int c = 0; c < 3; ++c) px[c] = st.pin[c * plane + (size_t)y * W + x];
            } else {
                const uint32_t b0 = rowb[3 * x], b1 = rowb[3 * x + 1], b2 = rowb[3 * x + 2];
                seen |= (b0 | b1 | b2) >> 1;
                float c0, c1, c2;
                if (DARK) { c0 = b0 ? 1.0f : 0.0f; c1 = b1 ? 1.0f : 0.0f; c2 = b2 ? 1.0f : 0.0f; }
                else { c0 = lut[b0]; c1 = lut[b1]; c2 = lut[b2]; }
                px[0] = fma_t(c2, a.M[2], fma_t(c1, a.M[1], c0 * a.M[0]));
                px[1] = fma_t(c2, a.M[5], fma_t(c1, a.M[4], c0 * a.M[3]));
                px[2] = fma_t(c2, a.M[8], fma_t(c1, a.M[7], c0 * a.M[6]));
            }
        };
        __syncthreads();  // this row's taps (t1 / td1 / td2) and, with PREF, its staged bytes are in place
        if (!(a.ablate & 1))
        switch (r1) {  // uniform per row
            case 0: streak_across<0>(load_px, buf1, W, PW, t1, tid); break;
            case 1: streak_across<1>(load_px, buf1, W, PW, t1, tid); break;
            case 2: streak_across<2>(load_px, buf1, W, PW, t1, tid); break;
            case 3: streak_across<3>(load_px, buf1, W, PW, t1, tid); break;
            case 4: streak_across<4>(load_px, buf1, W, PW, t1, tid); break;
            case 5: streak_across<5>(load_px, buf1, W, PW, t1, tid); break;
            default: streak_across<6>(load_px, buf1, W, PW, t1, tid); break;  // 13 taps: the table's maximum
        }
        if constexpr (!PLANES) {
            if (!DARK && seen) a.flags[f] = 1u;
        }
        __syncthreads();
        // the halo is filled out to the window radius RB (>= r): the zero taps beyond r must meet finite values
        const int rb1 = r1 <= 4 ? 4 : 8, rb2 = r2 <= 4 ? 4 : (r2 <= 8 ? 8 : (r2 <= 12 ? 12 : 16));
        streak_fill_halo(buf1, W, PW, rb1, tid);
        __syncthreads();
        // ---- pass 1, "column" direction = along the image row (sigma_x) -> buf0 ----------------------------
        if (a.ablate & 2) {}
        else if (r1 <= 4) streak_along_row<4>(buf1_4, buf0_4, W, PW4, td1, tid);
        else streak_along_row<8>(buf1_4, buf0_4, W, PW4, td1, tid);
        __syncthreads();
        streak_fill_halo(buf0, W, PW, rb2, tid);
        __syncthreads();
        // ---- pass 2: 1-tap row direction is x*1.0 (exact); along the image row again (sigma_y) -> buf1 ----
        if (a.ablate & 4) {}
        else if (r2 <= 4) streak_along_row<4>(buf0_4, buf1_4, W, PW4, td2, tid);
        else if (r2 <= 8) streak_along_row<8>(buf0_4, buf1_4, W, PW4, td2, tid);
        else if (r2 <= 12) streak_along_row<12>(buf0_4, buf1_4, W, PW4, td2, tid);
        else streak_along_row<16>(buf0_4, buf1_4, W, PW4, td2, tid);
        __syncthreads();
        if constexpr (PLANES) {
            const size_t plane = (size_t)a.H * W;
            for (int x = tid; x < W; x += kST) {
#pragma unroll
                for (int c = 0; c < 3; ++c) st.pout[c * plane + (size_t)y * W + x] = buf1[(size_t)c * PW + kPad + x];
            }
            continue;
        }
        // ---- chroma compression (rabbit, panda), quantise -> staged bytes -----------------------------------
        for (int x = (a.ablate & 8) ? W : tid; x < W; x += kST) {
            float v[3] = {buf1[kPad + x], buf1[PW + kPad + x], buf1[2 * PW + kPad + x]};
            if (a.chroma_enable) {  // apply_chroma_compression, animal_utils.py:180-181
                const float gray = ((v[0] + v[1]) + v[2]) / 3.0f;
                v[0] = gray + (v[0] - gray) * a.chroma_keep;
                v[1] = gray + (v[1] - gray) * a.chroma_keep;
                v[2] = gray + (v[2] - gray) * a.chroma_keep;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) outb[3 * x + c] = (uint8_t)quantize_coarse<float, kCoarseNFix>(v[c], thr, coarse, qc.lo_key);
        }
        __syncthreads();
        if ((((uintptr_t)dst | (uintptr_t)n3) & 3u) == 0) {
            for (int i = tid; i < n3 / 4; i += kST) reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(outb)[i];
        } else {
            for (int i = tid; i < n3; i += kST) dst[i] = outb[i];
        }
    }
}

}  // namespace

int avx_launch_dichromat_streak(avx_ctx* ctx, avx_ws* ws, DichromatArgs& a, const avx_dichromat_desc* d, hipStream_t s) {
    AVX_REQUIRE(ctx, d->streak_rows_host != nullptr && d->streak_stride >= 48, "avx_dichromat_u8: streak tables missing (stride >= 48)");
    const int PW = ((a.W + 3) / 4 * 4) + 2 * 16;
    const size_t lds = sizeof(float) * (size_t)PW * 6;  // (the byte staging shares buf0's storage)
    AVX_REQUIRE(ctx, ctx->coarse_n_fix[0] <= kCoarseNFix, "quantiser needs %d refinements, kernel built for %d", ctx->coarse_n_fix[0], kCoarseNFix);
    QuantCoarse qc{ctx->d_coarse_f32, ctx->coarse_lo_key[0], ctx->coarse_n_keys[0], ctx->coarse_n_fix[0]};
    AVX_REQUIRE(ctx, lds <= 150 * 1024, "avx_dichromat_u8: frame width %d too large for the streak kernel (row must fit LDS)", a.W);
    const size_t tbytes = sizeof(float) * (size_t)a.H * d->streak_stride;
    { const int rcu = avx_upload_row_table(ctx, ws, d->streak_rows_host, tbytes, s); if (rcu) return rcu; }
    StreakArgs st{ws->d_row_gain, d->streak_stride, nullptr, nullptr};
    if (!getenv("AVX_DIAG")) a.ablate = 0;  // phase ablation (wrong outputs, counters only) needs AVX_DIAG=1 next to AVX_ABLATE
    const bool aligned = ((((size_t)a.H * a.W * 3 * (size_t)a.n_frames) | (size_t)(uintptr_t)a.in) & 3u) == 0;
    const bool pref6 = aligned && a.W * 3 + 8 <= 6 * kST * 4, pref12 = aligned && a.W * 3 + 8 <= 12 * kST * 4;
    auto kmain = pref6 ? dichromat_streak_kernel<false, false, true, 6> : (pref12 ? dichromat_streak_kernel<false, false, true, 12> : dichromat_streak_kernel<false>);
    auto kdark = pref6 ? dichromat_streak_kernel<true, false, true, 6> : (pref12 ? dichromat_streak_kernel<true, false, true, 12> : dichromat_streak_kernel<true>);
    AVX_HIP(ctx, hipFuncSetAttribute((const void*)kmain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    AVX_HIP(ctx, hipFuncSetAttribute((const void*)kdark, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long rows = (long)a.n_frames * a.H;
    // Persistent workgroups, exactly one resident set: the grid is CUs x (workgroups that fit a CU).  With a fixed 4 per CU and
    // only 3 fitting (LDS), a quarter of the workgroups started when the first ones had finished -- their 34 rows each ran as a
    // second round at a third of the machine's width (sheep 1080p: 83 or 102 GP/s from one run to the next, depending on how
    // the dispatcher interleaved them; 45 rows each in one round is the same work without the tail).
    int per_cu = 0;
    AVX_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kmain, kST, lds));
    if (per_cu < 1) per_cu = 1;
    const long cap = (long)ctx->num_cus * per_cu;
    const int grid = (int)(rows < cap ? rows : cap);
    if (getenv("AVX_TUNE_LOG")) fprintf(stderr, "[avx streak] lds=%zu occupancy=%d workgroups/CU grid=%d\n", lds, per_cu, grid);
    AVX_HIP(ctx, hipMemsetAsync(a.flags, 0, sizeof(uint32_t) * a.n_frames, s));
    hipLaunchKernelGGL(kmain, dim3(grid), dim3(kST), lds, s, a, st, qc);
    AVX_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(kdark, dim3(grid), dim3(kST), lds, s, a, st, qc);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

// The streak blur alone, between float32 planes (float frames of the streak species; same row kernels and FMA order).
extern "C" int avx_streak_planes_f32(avx_ctx* ctx, const float* src_planes, float* dst_planes, int H, int W, const float* rows_host,
                                     int stride, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, src_planes && dst_planes && rows_host && H > 0 && W > 0 && stride >= 48, "avx_streak_planes_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    avx_ws* ws = avx_workspace(ctx, s);
    AVX_REQUIRE(ctx, ws != nullptr, "avx_streak_planes_f32: no workspace for this stream");
    const int PW = ((W + 3) / 4 * 4) + 2 * 16;
    const size_t lds = sizeof(float) * (size_t)PW * 6;
    AVX_REQUIRE(ctx, lds <= 150 * 1024, "avx_streak_planes_f32: frame width %d too large (a row must fit LDS)", W);
    const size_t tbytes = sizeof(float) * (size_t)H * stride;
    { const int rcu = avx_upload_row_table(ctx, ws, rows_host, tbytes, s); if (rcu) return rcu; }
    DichromatArgs a{};
    a.H = H; a.W = W; a.n_frames = 1;
    StreakArgs st{ws->d_row_gain, stride, src_planes, dst_planes};
    QuantCoarse qc{};
    auto k = dichromat_streak_kernel<false, true>;
    AVX_HIP(ctx, hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;  // one resident set of persistent workgroups (see avx_launch_dichromat_streak)
    AVX_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k, kST, lds));
    const long cap = (long)ctx->num_cus * (per_cu < 1 ? 1 : per_cu);
    hipLaunchKernelGGL(k, dim3((unsigned)(H < cap ? H : cap)), dim3(kST), lds, s, a, st, qc);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}
