// csrc/dichromat_march.hip -- variant 2 of the fused dichromat path: the MARCHING STRIP kernel.
//
// Same arithmetic contract as dichromat.hip (bit-exact with oracle/avxref.cpp); different schedule,
// designed around what the 2-D tile variant measured as its limits on MI355X (in-kernel stamps + PMC,
// DESIGN.md): 10 barriers per 4096 px, 3 halo-sized planes in LDS (=> 2 waves/SIMD) and every phase
// latency-bound.
//
//   * A 384-thread workgroup (128 column groups x 3 channels) owns a strip up to 512 px wide
//     (256 for the float64 cat tail) and MARCHES down a chunk of rows, SY rows per iteration.
//   * LDS holds only SY raw rows, SY decoded rows (3 channel planes) and SY output rows (~40 KiB):
//     3 workgroups per CU.  No vertical halo is ever recomputed.
//   * Each thread owns XPT adjacent columns of ONE channel.  Row pass: one 16-byte-vector window
//     read per new row (lanes read consecutive vectors: conflict-free).  Column pass: the 2R+SY
//     row window of its columns lives in REGISTERS and is shifted by SY per iteration - the column
//     pass reads no LDS at all.
//   * 3 barriers per SY x strip-width pixels; raw rows for iteration t+1 are in flight (registers)
//     while iteration t computes.
#include <cstdlib>

#include "dichromat_common.h"

using namespace avxk;

namespace {

// -DAVX_MARCH_DMA=1: the producer wave stages the raw rows by LDS-direct loads (global_load_lds_dword into a third raw buffer, counted vmcnt) instead of through
// registers + ds_write (VERDICT r02's lever 6a).  Built and measured in round 3, bit-exact in all 183 dichromat tests and SLOWER on every species (same-box A/B,
// profiles/r03/ab_march_lds_dma.txt: cat 161.0 -> 152.9, dog 128.8 -> 121.7, wolf 195.8 -> 184.6, squirrel 261.9 -> 257.5 GP/s): 4-byte LDS-direct loads cost the
// producer more than the register staging they replace.  Off by default.
#ifndef AVX_MARCH_DMA
#define AVX_MARCH_DMA 0
#endif
// Workgroup = NG column groups x 3 channels (NG a multiple of 64 keeps the channel wave-uniform).

struct MarchGeom {
    int nstrips, sw;   // strips per frame row, nominal strip width (multiple of XPT)
    int nchunks, ch;   // row chunks per frame, rows per chunk (multiple of SY)
    int xcd_remap;     // 1: blocks b and b+8 are neighbours (same XCD under round-robin dispatch)
};

template <typename T, int N> struct VecN;
template <> struct VecN<float, 4> { using type = float4; };
template <> struct VecN<float, 2> { using type = float2; };
template <> struct VecN<double, 2> { using type = double2; };
template <> struct VecN<double, 1> { using type = double; };

// SPEC (wave-specialised form): one extra wave per workgroup is the PRODUCER -- it alone stages the raw rows and decodes them
// for the next iteration -- while the 3 * NG / 64 compute waves only run the row/column passes and the quantiser.  In the
// plain form wave 0 did all of the decode on top of its share of compute and the other waves waited for it at the barrier
// (in-kernel stamps: raw staging 26 % + decode 27 % + compute 35 % of wave 0's iteration).
template <typename T, int R, int SY, int XPT_, int NG, bool SPEC = false>
struct MarchCfg {
    static constexpr int kComputeThreads = 3 * NG;
    static constexpr int kMarchThreads = 3 * NG + (SPEC ? 64 : 0);
    static constexpr int kGroups = NG;
    static constexpr int XPT = XPT_;                          // columns per thread = one LDS vector read
    static constexpr int SW = XPT * kGroups;
    static constexpr int AWS = SW + 2 * R;
    static constexpr int NWV = 1 + (2 * R + XPT - 1) / XPT;   // vector reads per row window
    static constexpr int NG4 = (AWS + 3) / 4;                 // decode groups (4 px) per row
    // SPEC: the producer decodes a row pair in ONE pass of its 64 lanes, so a strip (with halo) is at most 256 / SY px wide
    // (float64 only: its decode is heavy enough that a second, nearly empty pass would cost more than the narrower strips)
    static constexpr bool kCapStrips = SPEC && sizeof(T) == 8;
    static constexpr int NG4S = kCapStrips ? 64 / (SY / 2) : NG4;
    static constexpr int SW_CAP = kCapStrips ? (NG4S * 4 - 2 * R < SW ? NG4S * 4 - 2 * R : SW) : SW;
    static constexpr int PA0 = SW - XPT + NWV * XPT;
    // float32 SPEC: the decode's lanes alternate between the two row pairs (planes 3 PA entries apart) and each writes 32
    // bytes; with PA = 2 mod 4 the four lanes of one row pair and the four of the other in a ds_write_b128 group land on
    // interleaved 16-byte slots of the 128-byte write row instead of the same ones (2-way conflict otherwise).
    static constexpr int PA = ((PA0 > NG4 * 4 ? PA0 : NG4 * 4) + 3) / 4 * 4 + ((SPEC && sizeof(T) == 4) ? 2 : 0);
    // kSplit (the float64 cat): a decoded plane row is stored as its even pixels followed by its odd pixels (PAH entries each).
    // A thread of the row pass reads pixels 2 xg + i: with 16-byte entries in pixel order the lane stride is 32 bytes and every
    // ds_read_b128 group hits each 16-byte slot twice; split by parity the stride is 16 bytes and the reads are conflict-free.
    // The decode writes one pixel per lane per store (pixel = lane + 64 j), so the 8 lanes of a ds_write_b128 group put 4
    // entries in each half: conflict-free as well when the halves are 64 bytes apart modulo the 128-byte write row
    // (PAH = 4 mod 8).  Before: 56 % of the kernel's LDS cycles were bank conflicts (PMC), half from these stores (4-way:
    // 64-byte lane stride), most of the rest from the row-pass reads.
    static constexpr bool kSplit = SPEC && sizeof(T) == 8;
    static constexpr int PAH = PA / 2;
    static_assert(!kSplit || (PA % 2 == 0 && PAH % 8 == 4 && 2 * PAH >= SW + 2 * R && XPT_ % 2 == 0), "parity-split plane layout");
    static constexpr int RAW_LEAD = 64;
    static constexpr int DPR = (((AWS * 3 + 3 + 16 + 3) & ~3) / 4 + 63) / 64 * 64;  // dwords per raw row, whole 64-lane segments
    static constexpr int RAWP = RAW_LEAD + DPR * 4;
    static constexpr int OUTP = SW * 3;
    static constexpr int WIN = 2 * R + SY;                    // rows in a column window
    static constexpr size_t off_thr = 0;
    static constexpr size_t off_lut = off_thr + 256 * sizeof(T);
    static constexpr size_t off_coarse = off_lut + 256 * sizeof(float);
    static constexpr size_t off_ktab = off_coarse + kCoarseTableBytes;
    static constexpr size_t off_A = off_ktab + 64 * sizeof(T);
    // Every row buffer exists twice: one barrier per iteration (see the main loop).
    static constexpr size_t A_bytes = (size_t)SY * 3 * PA * sizeof(T) + 64;
    static constexpr size_t raw_bytes = (size_t)SY * RAWP;
    static constexpr size_t out_bytes = (size_t)SY * OUTP;
    // SPEC: the raw rows have THREE buffers (iteration mod 3): the producer's LDS-direct loads of iteration t + 3 are issued two intervals before
    // their rows are decoded, as the register-staged loads were (loaded in one interval, written in the next)
    static constexpr int kRawBufs = (AVX_MARCH_DMA && SPEC) ? 3 : 2;
    static constexpr size_t off_raw = off_A + 2 * A_bytes;
    static constexpr size_t off_out = off_raw + kRawBufs * raw_bytes;
    static constexpr size_t lds_bytes = ((off_out + 2 * out_bytes + 15) / 16) * 16;
    static_assert(3 * R <= RAW_LEAD, "raw lead-in too small for this radius");
    static_assert(lds_bytes <= 160 * 1024, "does not fit LDS");
};

__device__ __forceinline__ uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) {
    return __builtin_amdgcn_alignbyte(hi, lo, sh);  // ({hi,lo} >> 8*sh) & 0xffffffff
}

template <typename T> struct Pair;
template <> struct Pair<float> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct Pair<double> { typedef double type __attribute__((ext_vector_type(2))); };

// Packed arithmetic: P = {value for row/column a, value for row/column b}.  For float these are
// v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 - the only way to the 157 TFLOP/s f32 vector rate on
// gfx950 (a plain wave64 VALU instruction holds its SIMD for 4 cycles); for double they lower to two
// scalar instructions.  Each half is the same IEEE operation as the scalar contract.
template <typename P> __device__ __forceinline__ P pfma(P a, P b, P c) { return __builtin_elementwise_fma(a, b, c); }

// STAMP = true is a DIAGNOSTIC instantiation (AVX_STAMPS=1): tid 0 accumulates s_memtime deltas per section
// into a.stamps (never into an output); its run time is not representative.
template <typename T, int COLOR, bool DARK, int R, int SY, int XPT_, int NG, int MINW, int NFIX, bool STAMP = false, bool SPEC = false>
__global__ __launch_bounds__(3 * NG + (SPEC ? 64 : 0), MINW) void dichromat_march_kernel(DichromatArgs a, Taps<T> taps, QuantCoarse qc, MarchGeom g) {
    using C = MarchCfg<T, R, SY, XPT_, NG, SPEC>;
    constexpr int kMarchThreads = C::kMarchThreads;
    constexpr int kGroups = C::kGroups;
    using P = typename Pair<T>::type;
    constexpr int XPT = C::XPT;
    constexpr int NWAVES = kMarchThreads / 64;
    static_assert(SY % 2 == 0 && XPT % 2 == 0, "rows and columns are processed in pairs");
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* thr = reinterpret_cast<T*>(smem_raw + C::off_thr);
    float* lut = reinterpret_cast<float*>(smem_raw + C::off_lut);
    uint8_t* coarse = smem_raw + C::off_coarse;
    T* ktab = reinterpret_cast<T*>(smem_raw + C::off_ktab);
    // double-buffered by iteration parity: A [SY/2][3][PA] of {row 2k, row 2k+1}; RAW [SY][RAWP]; OUT [SY][OUTP]
    auto A_of = [&](int t) { return reinterpret_cast<P*>(smem_raw + C::off_A + (size_t)(t & 1) * C::A_bytes); };
    auto RAW_of = [&](int t) { return smem_raw + C::off_raw + (size_t)(C::kRawBufs == 3 ? (unsigned)t % 3u : (unsigned)t & 1u) * C::raw_bytes; };
    auto OUT_of = [&](int t) { return smem_raw + C::off_out + (size_t)(t & 1) * C::out_bytes; };
    // AVX_ABLATE (phase skipping, tuning only) exists in the diagnostic instantiation alone: as run-time tests these uniform
    // branches split the loop into small blocks and kept the compiler from overlapping the quantiser's LDS lookups across rows
    const int ablate = STAMP ? a.ablate : 0;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: row bookkeeping on the SALU

    // ---- which strip / chunk / frame ----------------------------------------------------------
    int b = blockIdx.x;
    if (g.xcd_remap) b = (b & 7) * ((int)gridDim.x >> 3) + (b >> 3);  // gridDim.x % 8 == 0 (host checked)
    const int strip = b % g.nstrips;
    const int chunk = (b / g.nstrips) % g.nchunks;
    const int f = b / (g.nstrips * g.nchunks);
    if (DARK && a.flags[f] != 0u) return;  // frame was not "all <= 1": the main pass is already right
    const int xs = strip * g.sw, ys = chunk * g.ch;
    const int vw = a.W - xs < g.sw ? a.W - xs : g.sw;    // valid output columns of this strip
    const int ch = a.H - ys < g.ch ? a.H - ys : g.ch;    // valid output rows of this chunk
    if (vw <= 0 || ch <= 0) return;
    const size_t frame_bytes = (size_t)a.H * a.W * 3;
    const uint8_t* fin = a.in + frame_bytes * f;
    uint8_t* fout = a.out + frame_bytes * f;
    const int gx0 = xs - R > 0 ? xs - R : 0;
    const int gx1 = xs + vw + R < a.W ? xs + vw + R : a.W;
    const int row_bytes = (gx1 - gx0) * 3;
    const int lead = gx0 - (xs - R);                       // halo pixels left of the image (0 off-border)
    const bool border = (xs - R < 0) || (xs + vw + R > a.W);
    const int aws = vw + 2 * R;                            // haloed width actually used

    for (int i = tid; i < 256; i += kMarchThreads) {
        thr[i] = reinterpret_cast<const T*>(a.enc_thr)[i];
        lut[i] = a.decode_lut[i];
    }
    for (int i = tid; i < kCoarseTableBytes; i += kMarchThreads) coarse[i] = i < (int)qc.n_keys ? qc.table[i] : (uint8_t)0;
    if (tid <= R) ktab[tid] = taps.k[R + tid];  // symmetric taps, by distance from the centre
    // The colour constants live in LDS (read per decode item), not in 30 SGPRs for the whole kernel: the scalar register file
    // was spilling through v_readlane (90 spilled SGPRs, 22 % of the main loop's VALU instructions were v_readlane_b32).
    float* cmf = reinterpret_cast<float*>(smem_raw + C::off_ktab + 16 * sizeof(T));         // [12]: M[9], alpha, 1 - alpha
    double* cbk = reinterpret_cast<double*>(smem_raw + C::off_ktab + 16 * sizeof(T) + 64);  // [9] (cat only)
    if (tid < 9) cmf[tid] = a.M[tid];
    if (tid == 9) cmf[9] = a.alpha;
    if (tid == 10) cmf[10] = a.one_minus_alpha;
    if constexpr (COLOR != AVX_COLOR_MATRIX) { if (tid < 9) cbk[tid] = a.Bk[tid]; }
    __syncthreads();
    // Taps, splat into both halves, straight from the kernel arguments: uniform values, so they live in SGPRs and enter each
    // packed FMA as its one scalar operand -- 2 (R + 1) VGPRs fewer than the vector copies (read back from LDS) this kernel used
    // while its scalar file was overcommitted (dog, 29 taps: 189 -> 165 VGPRs = three waves per SIMD instead of two, 79 -> 98 GP/s).
    P kk[R + 1];
#pragma unroll
    for (int d = 0; d <= R; ++d) { const T k = sizeof(T) == 8 ? ktab[d] : taps.k[R + d]; kk[d] = P{k, k}; }  // (cat's float64 taps: 4 SGPRs
                                                                                          // each, the scalar file spills again: VGPR copies)

    const bool producer = SPEC && wave >= C::kComputeThreads / 64;  // from the readfirstlane'd wave index: a SCALAR condition, so the
                                                                    // role branches are uniform and the row bookkeeping stays on the SALU
    const bool decoder = !SPEC || producer;                   // who decodes
    // The producer is one wave feeding three; it shares its SIMD with compute waves of this and other workgroups, and when it
    // falls behind everyone waits at the barrier: it gets the higher issue priority (AVX_ABLATE bit 128 in the diagnostic
    // build leaves it at the default).  Measured per configuration: cat 142 -> 149 GP/s, squirrel 244 -> 251; but dog 129 ->
    // 121, wolf 198 -> 161, lion 205 -> 197 (their compute waves are the critical path), so only where it paid.
    if (SPEC && (sizeof(T) == 8 || XPT_ == 4) && producer && !(ablate & 128)) __builtin_amdgcn_s_setprio(3);
    const bool stager = !SPEC || producer;                    // who stages the raw rows
    const int c = producer ? 0 : tid / kGroups;  // channel of this thread (uniform per wave: 128 = 2 waves)
    const int xg = tid - c * kGroups;            // column group: columns xg*XPT .. xg*XPT+XPT-1 of the strip
    const bool col_active = !producer && xg * XPT < vw;

    // ---- raw row loads: (row, 64-dword segment) units dealt to waves; row math is scalar ---------
    constexpr int NSEG = (C::DPR + 63) / 64;
    constexpr int NSTAGE = SPEC ? 1 : NWAVES;           // waves that share the raw-row units
    const int swave = SPEC ? 0 : wave;
    constexpr int NLD = (SY * NSEG + NSTAGE - 1) / NSTAGE;
    // The host only launches this kernel on batches whose byte size and base address are multiples of 4 (every standard
    // video size): aligned dwords that contain a valid byte then never cross the end of the batch, and no tail guard is
    // needed.  (The guarded path used to live here too; its scalar bookkeeping alone spilled 14 SGPRs through v_readlane.)
    // Row addressing in 32 bits: offsets are relative to the frame base rounded DOWN to a dword (the host checks that a frame
    // is < 4 GiB), so a row costs three scalar instructions instead of a 64-bit multiply-add chain, and a load is one vector
    // add on top of a scalar base.  A wave issues one instruction every four cycles at best: this bookkeeping, not the
    // 8 loads it serves, was half of the producer wave's iteration.
    const uint32_t fin_align = (uint32_t)((uintptr_t)fin & 3u);
    const uint8_t* const fin_base = fin - fin_align;
    auto row_off = [&](int t, int s, uint32_t& shift) -> uint32_t {
        const int y = ys - R + t * SY + s;
        const int ay = y < 0 ? -y : y;
        const int my = 2 * (a.H - 1) - ay;
        const int gy = ay < my ? ay : my;  // BORDER_REFLECT_101 in one step (the host sends frames shorter than 2 (R + SY) rows
                                           // elsewhere): three scalar instructions, where the general loop was ~25 per row
        const uint32_t off = fin_align + ((uint32_t)gy * (uint32_t)a.W + (uint32_t)gx0) * 3u;
        shift = off & 3u;
        return off - shift;
    };
    // Unconditional, straight-line issue: lanes past the row's last dword re-read that last dword (always a valid address:
    // batch size and base are multiples of 4) instead of being masked off -- a guarded load is a branch, and a branch between
    // two loads keeps them from being in flight together.  Nothing consumes those duplicates but the detector, which masks them.
    auto issue_raw_loads = [&](int t, uint32_t (&rv)[NLD]) {
#pragma unroll
        for (int n = 0; n < NLD; ++n) {
            const int u = swave + n * NSTAGE;            // uniform
            const int s = u / NSEG, seg = u - s * NSEG; // uniform
            uint32_t shift;
            const uint32_t aoff = row_off(t, s < SY ? s : 0, shift);
            const int last = ((int)shift + row_bytes - 1) >> 2;
            const int d = seg * 64 + lane;
            const int dc = d < last ? d : last;
            rv[n] = *reinterpret_cast<const uint32_t*>(fin_base + (size_t)(aoff + (uint32_t)dc * 4u));
        }
    };
    // The same rows by LDS-direct loads (global_load_lds_dword: lane l's dword lands at M0 + 4 l): no staging registers, no write_raw.  Used by the producer
    // wave once the "byte > 1" detector has nothing left to find (it needs the bytes in registers): the loads go straight into RAW_of(t).  hipcc does not count
    // them: the producer waits for them itself (dma_wait) -- it issues no other vector-memory operation in the steady state (the compute waves store).
    constexpr bool kDma = AVX_MARCH_DMA && SPEC && NLD <= 32;
    auto dma_raw = [&](int t) {
        const unsigned raw_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)RAW_of(t);
#pragma unroll
        for (int n = 0; n < NLD; ++n) {
            const int u = swave + n * NSTAGE;            // uniform
            const int s = u / NSEG, seg = u - s * NSEG; // uniform
            if (s < SY) {
                uint32_t shift;
                const uint32_t aoff = row_off(t, s, shift);
                const int last = ((int)shift + row_bytes - 1) >> 2;
                const int d = seg * 64 + lane;
                const int dc = d < last ? d : last;
                const uint8_t* g = fin_base + (size_t)(aoff + (uint32_t)dc * 4u);
                const unsigned dst = raw_lds + (unsigned)(s * C::RAWP + C::RAW_LEAD + seg * 256);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"  // "m0 is reserved": nothing of the compiler's lives in M0 across this statement
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(__builtin_amdgcn_readfirstlane(dst)), "v"(g) : "memory", "m0");
#pragma clang diagnostic pop
            }
        }
    };
    // every LDS-direct load but the NLD most recent ones (all == true: every one) has landed
    auto dma_wait = [&](bool all) {
        if (all) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NLD) : "memory");
    };
    bool flag_done = false;  // wave-uniform: this wave has already reported a byte > 1 for the frame -- nothing left to detect
    auto write_raw = [&](int t, const uint32_t (&rv)[NLD]) {
        uint8_t* RAW = RAW_of(t);
        uint32_t seen = 0;
#pragma unroll
        for (int n = 0; n < NLD; ++n) {
            const int u = swave + n * NSTAGE;
            const int s = u / NSEG, seg = u - s * NSEG;
            if (s < SY) {
                const int d = seg * 64 + lane;
                uint32_t shift;
                (void)row_off(t, s, shift);                       // scalar
                {
                    *reinterpret_cast<uint32_t*>(RAW + (size_t)s * C::RAWP + C::RAW_LEAD + d * 4) = rv[n];
                    if (!DARK && !flag_done) {
                        // "byte > 1" detector, masked to the row's own bytes: only the first and the last dword of a row
                        // hold foreign bytes (<= 3 before / after it); lanes past the last dword hold copies of it
                        const int last = ((int)shift + row_bytes - 1) >> 2;                        // scalar
                        const uint32_t mlo = 0xfefefefeu << (8 * shift);                           // scalar
                        const uint32_t mhi = 0xfefefefeu >> (8 * (3 - (((int)shift + row_bytes - 1) & 3)));
                        uint32_t m = d >= last ? mhi : 0xfefefefeu;
                        m = d == 0 ? (m & mlo) : m;
                        seen |= rv[n] & m;
                    }
                }
            }
        }
        if (!DARK && !flag_done) {
            if (seen) a.flags[f] = 1u;  // benign race: every writer stores the same value
            flag_done = __builtin_amdgcn_readfirstlane((uint32_t)(__ballot(seen != 0u) != 0ull)) != 0u;
        }
    };
    // ---- output rows of iteration t: OUT (LDS) -> HBM -------------------------------------------
    // Store granularity is the same for every row of the strip: 16 bytes when row pitch, strip origin
    // and width allow it (all standard video sizes), else 4 bytes, else single bytes.
    const int nbytes = vw * 3;
    const uintptr_t o0 = (uintptr_t)(fout + (size_t)xs * 3);
    const uintptr_t pitch = (uintptr_t)a.W * 3;
    const uintptr_t oall = o0 | pitch | (uintptr_t)nbytes;
    const int gran = ((oall & 15u) == 0) ? 16 : (((oall & 7u) == 0) ? 8 : (((oall & 3u) == 0) ? 4 : 1));
    // Each thread's first (row, vector) item is fixed for the whole strip: the division by the run-time row length is done once,
    // not every iteration (one pass covers the tile whenever SY * per_row <= threads, which holds at 8- and 16-byte granularity).
    const int per_row = nbytes / gran;
    const int st_i0 = tid / per_row, st_e0 = tid - st_i0 * per_row;
    const uint32_t row_pitch = (uint32_t)a.W * 3u, strip_off = (uint32_t)xs * 3u;
    auto store_out = [&](int t) {
        const uint8_t* OUT = OUT_of(t);
        if (producer) return;  // SPEC: the compute waves store (they wait for the producer anyway)
        int i = st_i0, e = st_e0;
#pragma unroll 1
        for (int v = tid; v < SY * per_row; v += C::kComputeThreads) {
            if (v != tid) { i = v / per_row; e = v - i * per_row; }
            const int rel = t * SY + i - 2 * R;
            if (rel < 0 || rel >= ch) continue;
            uint8_t* grow = fout + (size_t)((uint32_t)(ys + rel) * row_pitch + strip_off);  // 32-bit offset inside the frame (< 4 GiB, host-checked)
            const uint8_t* orow = OUT + (size_t)i * C::OUTP;
            if (gran == 16) reinterpret_cast<uint4*>(grow)[e] = reinterpret_cast<const uint4*>(orow)[e];
            else if (gran == 8) reinterpret_cast<uint2*>(grow)[e] = reinterpret_cast<const uint2*>(orow)[e];
            else if (gran == 4) reinterpret_cast<uint32_t*>(grow)[e] = reinterpret_cast<const uint32_t*>(orow)[e];
            else grow[e] = orow[e];
        }
    };

    // Column windows in registers: cw[xp][j] = {row j @ column 2xp, row j @ column 2xp+1}, newest row last.
    P cw[XPT / 2][C::WIN];
#pragma unroll
    for (int xp = 0; xp < XPT / 2; ++xp)
#pragma unroll
        for (int j = 0; j < C::WIN; ++j) cw[xp][j] = P{(T)0, (T)0};

    const int n_iter = (ch + 2 * R + SY - 1) / SY;
    uint32_t rv[NLD];

    // decode iteration t's raw rows (RAW_of(t)) into A_of(t).  One item = 4 pixels x 2 rows: raw bytes ->
    // decode table -> colour chain on {row a, row b} pairs -> the row-paired layout the row pass reads.
    auto decode = [&](int t) {
        const uint8_t* RAW = RAW_of(t);
        P* A = A_of(t);
        if (!decoder) return;  // (SPEC only: the plain form has a barrier below that every thread must reach)
        constexpr int kStep = SPEC ? 64 : kMarchThreads;
        if constexpr (C::kSplit) {
            // One pixel x two rows per lane and step (pixel = lane + 64 j): every plane store of the wave is lane-contiguous.
            using F2 = typename Pair<float>::type;
            using D2 = typename Pair<double>::type;
            float M[12];
            {
                const float4 m0 = reinterpret_cast<const float4*>(cmf)[0], m1 = reinterpret_cast<const float4*>(cmf)[1], m2 = reinterpret_cast<const float4*>(cmf)[2];
                M[0] = m0.x; M[1] = m0.y; M[2] = m0.z; M[3] = m0.w; M[4] = m1.x; M[5] = m1.y; M[6] = m1.z; M[7] = m1.w; M[8] = m2.x; M[9] = m2.y; M[10] = m2.z; M[11] = m2.w;
            }
#pragma unroll 1
            for (int sp = (ablate & 1) ? SY : 0; sp < SY / 2; ++sp) {
                uint32_t shift0, shift1;
                (void)row_off(t, 2 * sp, shift0);
                (void)row_off(t, 2 * sp + 1, shift1);
                const int o0 = C::RAW_LEAD + (int)shift0 - 3 * lead, o1 = C::RAW_LEAD + (int)shift1 - 3 * lead;  // byte offset of pixel 0
                const uint8_t* r0 = RAW + (size_t)(2 * sp) * C::RAWP;
                const uint8_t* r1 = r0 + C::RAWP;
#pragma unroll 2
                for (int px = lane; px < aws; px += 64) {
                    // the three codes of a pixel as byte reads (ds_read_u8 with immediate offsets 0/1/2 on one address): no
                    // dword alignment, funnel shift or field extraction -- this wave is issue-bound, LDS instructions are cheap
                    const uint8_t* b0 = r0 + o0 + 3 * px;
                    const uint8_t* b1 = r1 + o1 + 3 * px;
                    const uint32_t a0 = b0[0], a1 = b0[1], a2 = b0[2], e0 = b1[0], e1 = b1[1], e2 = b1[2];
                    F2 c0, c1, c2;
                    if (DARK) {
                        c0 = F2{a0 ? 1.0f : 0.0f, e0 ? 1.0f : 0.0f};
                        c1 = F2{a1 ? 1.0f : 0.0f, e1 ? 1.0f : 0.0f};
                        c2 = F2{a2 ? 1.0f : 0.0f, e2 ? 1.0f : 0.0f};
                    } else {
                        c0 = F2{lut[a0], lut[e0]};
                        c1 = F2{lut[a1], lut[e1]};
                        c2 = F2{lut[a2], lut[e2]};
                    }
                    const F2 l = pfma(c2, F2{M[2], M[2]}, pfma(c1, F2{M[1], M[1]}, c0 * F2{M[0], M[0]}));
                    const F2 m = pfma(c2, F2{M[5], M[5]}, pfma(c1, F2{M[4], M[4]}, c0 * F2{M[3], M[3]}));
                    const F2 sv = pfma(c2, F2{M[8], M[8]}, pfma(c1, F2{M[7], M[7]}, c0 * F2{M[6], M[6]}));
                    P* dst = A + (size_t)(sp * 3) * C::PA + (px & 1) * C::PAH + (px >> 1);
                    if constexpr (COLOR == AVX_COLOR_MATRIX) {
                        dst[0] = P{(T)l.x, (T)l.y};
                        dst[C::PA] = P{(T)m.x, (T)m.y};
                        dst[2 * C::PA] = P{(T)sv.x, (T)sv.y};
                    } else {
                        const F2 lm = F2{M[9], M[9]} * l + F2{M[10], M[10]} * m;  // mul, mul, add (cat.py:99)
                        const D2 dlm = D2{(double)lm.x, (double)lm.y}, ds = D2{(double)sv.x, (double)sv.y};
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc) {
                            const double k0 = cbk[3 * cc], k1 = cbk[3 * cc + 1], k2 = cbk[3 * cc + 2];
                            const D2 r = pfma(ds, D2{k2, k2}, pfma(dlm, D2{k1, k1}, dlm * D2{k0, k0}));
                            dst[cc * C::PA] = P{(T)r.x, (T)r.y};
                        }
                    }
                }
            }
            if (border) {  // halo mirrors: the same wave wrote the sources (LDS serves a wave's accesses in order)
                const int nl = lead, nr = (xs + vw + R) - gx1;
                for (int item = lane; item < (SY / 2) * 3 * (nl + nr); item += 64) {
                    const int e = item % (nl + nr), pl = item / (nl + nr);
                    const int lx = e < nl ? e : (gx1 - (xs - R)) + (e - nl);
                    const int src = reflect101(xs - R + lx, a.W) - (xs - R);
                    A[(size_t)pl * C::PA + (lx & 1) * C::PAH + (lx >> 1)] = A[(size_t)pl * C::PA + (src & 1) * C::PAH + (src >> 1)];
                }
            }
            return;
        }
        // SPEC: items run over the strip's ACTUAL haloed width, row pairs interleaved (item = q4 * SY/2 + sp) -- how many 64-lane
        // passes the producer needs is then the host's choice of strip width alone (launch_march), not a compile-time constant
        constexpr bool kMix = SPEC && sizeof(T) == 4;  // (the float64 cat keeps row-pair-major items over its fixed 120-px strips:
                                                       //  interleaved, its 16-byte plane writes conflict -- 143 -> 114 GP/s)
        constexpr int kNG4 = kMix ? 1 : (SPEC ? C::NG4S : C::NG4);
        const int item_end = kMix ? (SY / 2) * ((aws + 3) >> 2) : (SY / 2) * kNG4;
#pragma unroll 1
        for (int item = (ablate & 1) ? (1 << 30) : (SPEC ? lane : tid); item < item_end; item += kStep) {
            const int sp = kMix ? item % (SY / 2) : item / kNG4, q4 = kMix ? item / (SY / 2) : item - sp * kNG4;
            if (!kMix && q4 * 4 >= aws) continue;
            uint32_t code[2][4][3];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int s = 2 * sp + h;
                uint32_t shift;  // the row's dword misalignment: three scalar instructions to recompute -- as a byte parked in LDS by the
                (void)row_off(t, s, shift);  // staging step it was one more dependent LDS round trip at the head of the decode chain
                const int boff = C::RAW_LEAD + (int)shift - 3 * lead + 12 * q4;  // LDS byte offset of pixel lx = 4*q4
                const uint32_t* dw = reinterpret_cast<const uint32_t*>(RAW + (size_t)s * C::RAWP + (boff & ~3));
                const uint32_t sh = (uint32_t)boff & 3u;
                const uint32_t d0 = dw[0], d1 = dw[1], d2 = dw[2], d3 = dw[3];
                const uint32_t u0 = alignbyte(d1, d0, sh), u1 = alignbyte(d2, d1, sh), u2 = alignbyte(d3, d2, sh);
                code[h][0][0] = u0 & 255u; code[h][0][1] = (u0 >> 8) & 255u; code[h][0][2] = (u0 >> 16) & 255u;
                code[h][1][0] = u0 >> 24;  code[h][1][1] = u1 & 255u;        code[h][1][2] = (u1 >> 8) & 255u;
                code[h][2][0] = (u1 >> 16) & 255u; code[h][2][1] = u1 >> 24; code[h][2][2] = u2 & 255u;
                code[h][3][0] = (u2 >> 8) & 255u; code[h][3][1] = (u2 >> 16) & 255u; code[h][3][2] = u2 >> 24;
            }
            P o[3][4];
            using F2 = typename Pair<float>::type;
            float M[12];
            {
                const float4 m0 = reinterpret_cast<const float4*>(cmf)[0], m1 = reinterpret_cast<const float4*>(cmf)[1], m2 = reinterpret_cast<const float4*>(cmf)[2];
                M[0] = m0.x; M[1] = m0.y; M[2] = m0.z; M[3] = m0.w; M[4] = m1.x; M[5] = m1.y; M[6] = m1.z; M[7] = m1.w; M[8] = m2.x; M[9] = m2.y; M[10] = m2.z; M[11] = m2.w;
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                F2 c0, c1, c2;
                if (DARK) {  // get_normalized_image skips /255 when max <= 1: codes 0/1 are 0.0/1.0
                    c0 = F2{code[0][p][0] ? 1.0f : 0.0f, code[1][p][0] ? 1.0f : 0.0f};
                    c1 = F2{code[0][p][1] ? 1.0f : 0.0f, code[1][p][1] ? 1.0f : 0.0f};
                    c2 = F2{code[0][p][2] ? 1.0f : 0.0f, code[1][p][2] ? 1.0f : 0.0f};
                } else {
                    c0 = F2{lut[code[0][p][0]], lut[code[1][p][0]]};
                    c1 = F2{lut[code[0][p][1]], lut[code[1][p][1]]};
                    c2 = F2{lut[code[0][p][2]], lut[code[1][p][2]]};
                }
                // out_i = fma(c2, M[i][2], fma(c1, M[i][1], c0*M[i][0]))   (dog.py:47 as an FMA chain)
                const F2 l = pfma(c2, F2{M[2], M[2]}, pfma(c1, F2{M[1], M[1]}, c0 * F2{M[0], M[0]}));
                const F2 m = pfma(c2, F2{M[5], M[5]}, pfma(c1, F2{M[4], M[4]}, c0 * F2{M[3], M[3]}));
                const F2 sv = pfma(c2, F2{M[8], M[8]}, pfma(c1, F2{M[7], M[7]}, c0 * F2{M[6], M[6]}));
                if constexpr (COLOR == AVX_COLOR_MATRIX) {
                    o[0][p] = P{(T)l.x, (T)l.y};
                    o[1][p] = P{(T)m.x, (T)m.y};
                    o[2][p] = P{(T)sv.x, (T)sv.y};
                } else {
                    const F2 lm = F2{M[9], M[9]} * l + F2{M[10], M[10]} * m;  // mul, mul, add (cat.py:99)
                    using D2 = typename Pair<double>::type;
                    const D2 dlm = D2{(double)lm.x, (double)lm.y}, ds = D2{(double)sv.x, (double)sv.y};
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        const double b0 = cbk[3 * cc], b1 = cbk[3 * cc + 1], b2 = cbk[3 * cc + 2];
                        const D2 r = pfma(ds, D2{b2, b2}, pfma(dlm, D2{b1, b1}, dlm * D2{b0, b0}));
                        o[cc][p] = P{(T)r.x, (T)r.y};
                    }
                }
            }
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                P* dst = A + ((size_t)(sp * 3 + cc)) * C::PA + q4 * 4;
#pragma unroll
                for (int p = 0; p < 4; ++p) dst[p] = o[cc][p];
            }
        }
        if (border) {  // BORDER_REFLECT_101 in x: halo entries outside the image copy their mirror pixel
            if constexpr (!SPEC) __syncthreads();  // SPEC: the same wave wrote A (LDS serves a wave's accesses in order)
            const int nl = lead, nr = (xs + vw + R) - gx1;  // halo pixels left / right of the image
            for (int item = SPEC ? lane : tid; item < (SY / 2) * 3 * (nl + nr); item += kStep) {
                const int e = item % (nl + nr), pl = item / (nl + nr);
                const int lx = e < nl ? e : (gx1 - (xs - R)) + (e - nl);
                const int src = reflect101(xs - R + lx, a.W) - (xs - R);
                A[(size_t)pl * C::PA + lx] = A[(size_t)pl * C::PA + src];
            }
        }
    };
    // row pass for the SY new rows of iteration t (two rows per packed op), then the column pass and the
    // quantiser from the register windows (two columns per packed op) -> OUT_of(t).
    auto compute = [&](int t) {
        const P* A = A_of(t);
        uint8_t* OUT = OUT_of(t);
        if (col_active) {
#pragma unroll
            for (int xp = 0; xp < XPT / 2; ++xp)
#pragma unroll
                for (int j = 0; j < 2 * R; ++j) cw[xp][j] = cw[xp][j + SY];
#pragma unroll
            for (int sp = 0; sp < ((ablate & 2) ? 0 : SY / 2); ++sp) {
                const P* src = A + ((size_t)(sp * 3 + c)) * C::PA + xg * XPT;
                P w[XPT + 2 * R];  // {row 2sp, row 2sp+1} at columns xg*XPT + i
#pragma unroll
                for (int i = 0; i < XPT + 2 * R; ++i) {
                    if constexpr (C::kSplit) w[i] = (A + ((size_t)(sp * 3 + c)) * C::PA + xg * (XPT / 2))[(i & 1) * C::PAH + (i >> 1)];
                    else w[i] = src[i];
                }
                P acc[XPT];
#pragma unroll
                for (int x = 0; x < XPT; ++x) {
                    P sacc = w[x] * kk[R];  // tap 0 is at distance R from the centre
#pragma unroll
                    for (int j = 1; j <= 2 * R; ++j) sacc = pfma(w[x + j], kk[j < R ? R - j : j - R], sacc);
                    acc[x] = sacc;
                }
#pragma unroll
                for (int xp = 0; xp < XPT / 2; ++xp) {  // {rows} x columns -> rows x {columns}
                    cw[xp][2 * R + 2 * sp] = P{acc[2 * xp].x, acc[2 * xp + 1].x};
                    cw[xp][2 * R + 2 * sp + 1] = P{acc[2 * xp].y, acc[2 * xp + 1].y};
                }
                __builtin_amdgcn_sched_barrier(0);  // one row-pair window live at a time (register budget)
            }
            // Column pass for all SY rows first, then the quantiser in BATCHES: every bucket lookup of the iteration is issued
            // before the first refinement read, and every byte is written at the end.  Written row by row, each row's two
            // dependent LDS lookups (bucket table, then threshold) completed before the next row started -- the OUT byte
            // stores in between keep the compiler from hoisting the next row's reads -- and the quantiser was 47 % of the compute
            // waves' iteration (in-kernel stamps with the lookups ablated).
            constexpr int NS = SY * XPT;  // samples of this thread in the iteration
            T sv[NS];
#pragma unroll
            for (int i = 0; i < SY; ++i) {
#pragma unroll
                for (int xp = 0; xp < XPT / 2; ++xp) {
                    P sacc = cw[xp][R + i] * kk[0];
#pragma unroll
                    for (int j = 1; j <= R; ++j) sacc = pfma(cw[xp][R + i + j] + cw[xp][R + i - j], kk[j], sacc);
                    sv[i * XPT + 2 * xp] = sacc.x;
                    sv[i * XPT + 2 * xp + 1] = sacc.y;
                }
            }
            if (!(ablate & 4)) {
                uint32_t q[NS];
                if (ablate & 8) {
#pragma unroll
                    for (int n = 0; n < NS; ++n) q[n] = (uint32_t)(sv[n] * (T)255);
                } else {
#pragma unroll
                    for (int n = 0; n < NS; ++n) {  // clip + bucket lookups, all in flight together
                        sv[n] = sv[n] < (T)0 ? (T)0 : (sv[n] > (T)1 ? (T)1 : sv[n]);
                        const uint32_t key = key_of(sv[n]);
                        q[n] = (uint32_t)coarse[(key > qc.lo_key ? key : qc.lo_key) - qc.lo_key];
                    }
#pragma unroll
                    for (int f2 = 0; f2 < NFIX; ++f2) {  // refinement: thr[255] is a huge pad, never passes
                        T tv[NS];
#pragma unroll
                        for (int n = 0; n < NS; ++n) tv[n] = thr[q[n]];
#pragma unroll
                        for (int n = 0; n < NS; ++n) q[n] += (tv[n] <= sv[n]) ? 1u : 0u;
                    }
                }
#pragma unroll
                for (int i = 0; i < SY; ++i)
#pragma unroll
                    for (int xp = 0; xp < XPT / 2; ++xp) {
                        uint8_t* dst = OUT + (size_t)i * C::OUTP + (xg * XPT + 2 * xp) * 3 + c;
                        dst[0] = (uint8_t)q[i * XPT + 2 * xp];
                        dst[3] = (uint8_t)q[i * XPT + 2 * xp + 1];
                    }
            }
        }
    };

    // Software pipeline, ONE barrier per iteration.  In interval t the workgroup stores the outputs of t-1,
    // stages the raw rows of t+2 (already in registers) and fetches those of t+3, decodes t+1 and computes t:
    // all on different buffers (parity double-buffering), so nothing inside an interval needs ordering.
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = 0;
    auto stamp = [&](int slot) {
        if constexpr (STAMP) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (slot >= 0) st_acc[slot] += now - st_t;
            st_t = now;
        }
    };
    if (stager) {
        issue_raw_loads(0, rv);
        write_raw(0, rv);
        if (n_iter > 1) issue_raw_loads(1, rv);
    }
    __syncthreads();
    decode(0);
    if (stager && n_iter > 1) {
        write_raw(1, rv);
        if (n_iter > 2) issue_raw_loads(2, rv);
    }
    __syncthreads();
    bool have_regs = stager && n_iter > 2;  // the prologue's issue_raw_loads(2)
    bool dma_used = false;
#pragma unroll 1
    for (int t = 0; t < n_iter; ++t) {
        stamp(-1);
        // raw staging first: its s_waitcnt vmcnt covers only the loads issued one interval ago (vmcnt counts
        // stores too, in order: behind store_out it would also wait for those stores to retire)
        bool dma_issued = false;
        if (stager) {
            if (have_regs) {  // the rows of t + 2, loaded in the previous interval
                write_raw(t + 2, rv);
                have_regs = false;
            }
            stamp(1);
            if (t + 3 < n_iter) {
                if (kDma && producer && (DARK || flag_done)) { dma_raw(t + 3); dma_issued = true; dma_used = true; }
                else { issue_raw_loads(t + 3, rv); have_regs = true; }
            }
        }
        stamp(2);
        if (t + 1 < n_iter) decode(t + 1);
        stamp(3);
        compute(t);
        stamp(4);
        if (t > 0 && !(ablate & 32)) store_out(t - 1);
        stamp(0);
        if (kDma && dma_used) dma_wait(!dma_issued);  // the rows decode(t + 2) will read (issued in the previous interval) are in LDS
        __syncthreads();
        stamp(5);
    }
    store_out(n_iter - 1);
    if constexpr (STAMP) {
        if (tid == ((SPEC && !(a.ablate & 64)) ? C::kComputeThreads : 0) && a.stamps) {  // AVX_ABLATE=64: the first compute wave instead of the producer
            for (int i = 0; i < 6; ++i) atomicAdd(a.stamps + i, st_acc[i]);
            atomicAdd(a.stamps + 6, (unsigned long long)n_iter);
        }
    }
}

constexpr int kNarrowStrips = 1 << 16;  // flag bit in a remembered launch geometry: use the narrowed strips

template <typename T, int COLOR, int R, int SY, int XPT, int NG, int MINW, int NFIX, bool SPEC = false>
int launch_march(avx_ctx* ctx, DichromatArgs& a, const avx_dichromat_desc* d, const QuantCoarse& qc, hipStream_t s) {
    using C = MarchCfg<T, R, SY, XPT, NG, SPEC>;
    constexpr int kMarchThreads = C::kMarchThreads;
    Taps<T> taps;
    for (int i = 0; i < AVX_MAX_KSIZE; ++i) taps.k[i] = (T)0;
    for (int i = 0; i < d->ksize; ++i) taps.k[i] = (T)d->taps_host[i];
    auto kmain = dichromat_march_kernel<T, COLOR, false, R, SY, XPT, NG, MINW, NFIX, false, SPEC>;
    auto kdark = dichromat_march_kernel<T, COLOR, true, R, SY, XPT, NG, MINW, NFIX, false, SPEC>;
    const size_t lds = C::lds_bytes;
    AVX_HIP(ctx, hipFuncSetAttribute((const void*)kmain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    AVX_HIP(ctx, hipFuncSetAttribute((const void*)kdark, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    AVX_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kmain, kMarchThreads, lds));
    if (per_cu < 1) per_cu = 1;
    MarchGeom g{};
    // Strip width.  The producer wave (SPEC) decodes a strip's row pairs in passes of 64 items (4 px x 2 rows each); a pass
    // costs the same whether 64 lanes or 6 are active, so a strip whose last pass would be nearly empty MAY be better narrowed
    // to fit one pass less (cat: 120 px, fixed in MarchCfg; float32: a multiple of 16 px so rows still store as 16-byte
    // vectors -- wolf 1080p: 112-px strips, +4 %; lion, squirrel: no difference; dog: 96-px strips leave a quarter of the
    // compute lanes idle, -17 %).  Like the row split below it is measured on first use.  AVX_MARCH_SWCAP pins it (0 = full).
    int sw_narrow = 0;  // the narrowed alternative, 0 = none
    if constexpr (SPEC && sizeof(T) == 4) {
        constexpr int items = (SY / 2) * C::NG4, passes = (items + 63) / 64;
        if (passes > 1) sw_narrow = (((passes - 1) * 64 / (SY / 2)) * 4 - 2 * R) / 16 * 16;
        if (sw_narrow < 32 || sw_narrow >= C::SW) sw_narrow = 0;
    }
    long cols = 0;
    auto set_strips = [&](int sw_cap) {
        g.nstrips = (a.W + sw_cap - 1) / sw_cap;
        g.sw = ((a.W + g.nstrips - 1) / g.nstrips + C::XPT - 1) / C::XPT * C::XPT;
        // prefer 16-px multiples (16-byte output vectors) when they fit under the cap, even if the last strip comes out narrower
        const int sw16 = (g.sw + 15) / 16 * 16;
        if (sw16 <= sw_cap) { g.sw = sw16; g.nstrips = (a.W + g.sw - 1) / g.sw; }
        cols = (long)a.n_frames * g.nstrips;
    };
    int sw_base = C::SW_CAP;
    if constexpr (SPEC && sizeof(T) == 4) {
        const char* ce = getenv("AVX_MARCH_SWCAP");
        if (ce && *ce) { const int v = atoi(ce); sw_base = v >= 32 && v < C::SW ? v / C::XPT * C::XPT : C::SW; sw_narrow = 0; }
    }
    if (a.W <= sw_narrow) sw_narrow = 0;
    set_strips(sw_base);
    // Row chunks per (frame, strip).  The best split depends on how the workgroup count tiles the resident slots, the XCD
    // round-robin and the priming cost (2R rows per chunk): measured, not modelled -- the first call for a (kernel
    // configuration, batch, frame size) times a handful of candidates on the caller's own frames (the output does not
    // depend on the split) and the winner is remembered in the context.  AVX_MARCH_CHUNKS pins it.
    const long resident = (long)ctx->num_cus * per_cu;
    const long max_chunks = a.H / (8 * SY) > 0 ? a.H / (8 * SY) : 1;  // never shorter than 8*SY rows
    auto set_chunks = [&](long nc) {
        if (nc > max_chunks) nc = max_chunks;
        if (nc < 1) nc = 1;
        g.ch = (int)(((a.H + nc - 1) / nc + SY - 1) / SY * SY);
        g.nchunks = (a.H + g.ch - 1) / g.ch;
        const long tot = cols * g.nchunks;
        g.xcd_remap = (tot % 8 == 0) ? 1 : 0;
        return tot;
    };
    long nchunks = cols >= resident ? 1 : (resident + cols - 1) / cols;  // fallback: one resident wave of equal workgroups
    const char* e = getenv("AVX_MARCH_CHUNKS");
    const uint64_t tkey = ((uint64_t)(sizeof(T) == 8) << 63) | ((uint64_t)R << 56) | ((uint64_t)NG << 46) |
                          ((uint64_t)(a.n_frames & 0xfff) << 32) | ((uint64_t)(a.H & 0xffff) << 16) | (uint64_t)(a.W & 0xffff);  // == chunk_key()
    if (e && *e) {
        nchunks = atol(e);
    } else {
        int found = 0;
        for (int i = 0; i < ctx->n_march_tuned; ++i)
            if (ctx->march_tuned[i].key == tkey) { found = ctx->march_tuned[i].nchunks; break; }
        if (!found && (size_t)a.n_frames * a.H * a.W >= (size_t)256 * 1024) {  // tiny launches: not worth timing
            const long cand[] = {nchunks, 1, 2, 3, 4, 6, 8, 12, 16, 24};
            float best_ms = 3.4e38f;
            hipEvent_t e0, e1;
            AVX_HIP(ctx, hipEventCreate(&e0));
            AVX_HIP(ctx, hipEventCreate(&e1));
            {   // An idle GPU runs its first tens of milliseconds at lower clocks: the candidates timed first would lose to the
                // ones timed last.  Run the fallback geometry until ~40 ms have passed before timing anything.
                const long tot = set_chunks(nchunks);
                float warm_ms = 0.f;
                for (int i = 0; i < 200 && warm_ms < 40.f && tot < (1L << 30); ++i) {
                    AVX_HIP(ctx, hipEventRecord(e0, s));
                    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(kmain, dim3((unsigned)tot), dim3(kMarchThreads), lds, s, a, taps, qc, g);
                    AVX_HIP(ctx, hipEventRecord(e1, s));
                    AVX_HIP(ctx, hipEventSynchronize(e1));
                    float ms = 0.f;
                    AVX_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
                    warm_ms += ms;
                }
            }
            for (int narrow = 0; narrow <= (sw_narrow ? 1 : 0); ++narrow) {
            if (narrow) set_strips(sw_narrow);
            long tried[10];
            int ntried = 0;
            for (long nc : cand) {
                if (nc > max_chunks) nc = max_chunks;
                bool dup = false;
                for (int i = 0; i < ntried; ++i) dup = dup || tried[i] == nc;
                if (dup) continue;
                tried[ntried++] = nc;
                const long tot = set_chunks(nc);
                if (tot >= (1L << 30)) continue;
                float ms_min = 3.4e38f;
                for (int rep = 0; rep < 2; ++rep) {  // second run: warm instruction cache / tables
                    AVX_HIP(ctx, hipMemsetAsync(a.flags, 0, sizeof(uint32_t) * a.n_frames, s));
                    AVX_HIP(ctx, hipEventRecord(e0, s));
                    hipLaunchKernelGGL(kmain, dim3((unsigned)tot), dim3(kMarchThreads), lds, s, a, taps, qc, g);
                    AVX_HIP(ctx, hipEventRecord(e1, s));
                    AVX_HIP(ctx, hipEventSynchronize(e1));
                    float ms = 0.f;
                    AVX_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
                    ms_min = ms < ms_min ? ms : ms_min;
                }
                if (ms_min < best_ms) { best_ms = ms_min; found = (int)nc | (narrow ? kNarrowStrips : 0); }
            }
            }
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            if (found && ctx->n_march_tuned < 64) ctx->march_tuned[ctx->n_march_tuned++] = {tkey, found};
            if (getenv("AVX_TUNE_LOG")) fprintf(stderr, "[avx tune] chunks: f64=%d R=%d NG=%d frames=%d H=%d W=%d -> %d%s (%.3f ms)\n", (int)(sizeof(T) == 8), R, NG, a.n_frames, a.H, a.W, found & (kNarrowStrips - 1), (found & kNarrowStrips) ? " narrow strips" : "", best_ms);
        }
        if (found) nchunks = found & (kNarrowStrips - 1);
        set_strips((found & kNarrowStrips) && sw_narrow ? sw_narrow : sw_base);
    }
    const long total = set_chunks(nchunks);
    AVX_REQUIRE(ctx, total < (1L << 30), "avx_dichromat_u8: too many workgroups");
    if (getenv("AVX_STAMPS")) {
        if constexpr (R == 6 || R == 14 || (R == 4 && sizeof(T) == 8)) {
            auto kst = dichromat_march_kernel<T, COLOR, false, R, SY, XPT, NG, MINW, NFIX, true, SPEC>;
            AVX_HIP(ctx, hipFuncSetAttribute((const void*)kst, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            avx_ws* wst = avx_workspace(ctx, s);
            AVX_HIP(ctx, (wst && avx_ensure_scratch(ctx, wst, 64) == AVX_OK) ? hipSuccess : hipErrorOutOfMemory);
            a.stamps = (unsigned long long*)wst->d_scratch;
            AVX_HIP(ctx, hipMemsetAsync(a.stamps, 0, 64, s));
            AVX_HIP(ctx, hipMemsetAsync(a.flags, 0, sizeof(uint32_t) * a.n_frames, s));
            hipLaunchKernelGGL(kst, dim3((unsigned)total), dim3(kMarchThreads), lds, s, a, taps, qc, g);
            unsigned long long h[8];
            AVX_HIP(ctx, hipMemcpyAsync(h, a.stamps, 64, hipMemcpyDeviceToHost, s));
            AVX_HIP(ctx, hipStreamSynchronize(s));
            double tot = 0;
            for (int i = 0; i < 6; ++i) tot += (double)h[i];
            const double per = tot > 0 ? 100.0 / tot : 0, it = (double)h[6] > 0 ? (double)h[6] : 1;
            fprintf(stderr, "[avx march stamps R=%d XPT=%d blocks=%ld per_cu=%d ch=%d] cycles/iteration (wave 0) %.0f | store %.1f%% raw->LDS %.1f%% issue loads %.1f%% decode %.1f%% row+col+quant %.1f%% barrier %.1f%%\n",
                    R, XPT, total, per_cu, g.ch, tot / it, h[0] * per, h[1] * per, h[2] * per, h[3] * per, h[4] * per, h[5] * per);
            a.stamps = nullptr;
        }
    }
    AVX_HIP(ctx, hipMemsetAsync(a.flags, 0, sizeof(uint32_t) * a.n_frames, s));
    hipLaunchKernelGGL(kmain, dim3((unsigned)total), dim3(kMarchThreads), lds, s, a, taps, qc, g);
    AVX_HIP(ctx, hipGetLastError());
    // Fix-up for frames whose every byte is <= 1 (get_normalized_image does not divide those by 255);
    // every workgroup of any other frame exits on its first instruction.
    hipLaunchKernelGGL(kdark, dim3((unsigned)total), dim3(kMarchThreads), lds, s, a, taps, qc, g);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}

}  // namespace

static uint64_t chunk_key(bool f64, int R, int NG, int frames, int H, int W) {
    return ((uint64_t)f64 << 63) | ((uint64_t)R << 56) | ((uint64_t)NG << 46) | ((uint64_t)(frames & 0xfff) << 32) | ((uint64_t)(H & 0xffff) << 16) | (uint64_t)(W & 0xffff);
}
static uint64_t width_key(bool f64, int R, int frames, int H, int W) {
    return (1ull << 62) | ((uint64_t)f64 << 61) | ((uint64_t)R << 52) | ((uint64_t)(frames & 0xfff) << 32) | ((uint64_t)(H & 0xffff) << 16) | (uint64_t)(W & 0xffff);
}

// Launch geometries measured on MI355X for the BASELINE.json workloads (tools/gpu_bench.sh with AVX_TUNE_LOG=1): these
// shapes start tuned; any other (kernel, batch, frame size) is measured on its first call.
void avx_march_seed_tuned(avx_ctx* ctx) {
    static const struct { int f64, R, frames, H, W, NG, chunks; } kSeed[] = {
        {0, 14, 32, 1080, 1920, 64, 4},   // dog 1080p
        {0, 14, 8, 2160, 3840, 64, 6},    // dog 4K
        {0, 6, 32, 1080, 1920, 64, 8 | kNarrowStrips},  // wolf 1080p (112-px strips: 191 -> 199 GP/s)
        {0, 5, 32, 1080, 1920, 64, 8},    // lion / tiger 1080p
        {0, 3, 32, 1080, 1920, 64, 3},    // squirrel 1080p (3: 264.6 GP/s, 6: 254.1)
        {1, 4, 32, 1080, 1920, 64, 3},    // cat 1080p (default bench): sweep 3 / 4 / 6 / 8 / 12 / 16 -> 161.6 / 153.7 / 156.2 / 152.2 / 148.8 / 142.2 GP/s
        {1, 4, 8, 2160, 3840, 64, 3},     // cat 4K: 163.4 / 131.9 / 161.1 / 154.3 / 155.8 / 151.7
    };
    if (getenv("AVX_MARCH_NOSEED")) return;  // measure everything on first use (re-deriving the table below)
    for (const auto& e : kSeed) {
        if (ctx->n_march_tuned + 2 > 64) break;
        ctx->march_tuned[ctx->n_march_tuned++] = {width_key(e.f64, e.R, e.frames, e.H, e.W), e.NG};
        ctx->march_tuned[ctx->n_march_tuned++] = {chunk_key(e.f64, e.R, e.NG, e.frames, e.H, e.W), e.chunks};
    }
}

static int march_dispatch(avx_ctx* ctx, DichromatArgs& a, const avx_dichromat_desc* d, bool f64_cat, bool ng64, hipStream_t s) {
    constexpr int NF = kCoarseNFix;
    const int w = f64_cat ? 1 : 0;
    AVX_REQUIRE(ctx, ctx->coarse_n_fix[w] <= NF, "quantiser needs %d refinements, kernel built for %d", ctx->coarse_n_fix[w], NF);
    QuantCoarse qc{f64_cat ? ctx->d_coarse_f64 : ctx->d_coarse_f32, ctx->coarse_lo_key[w], ctx->coarse_n_keys[w], ctx->coarse_n_fix[w]};
    if (f64_cat) {
        if (a.r == 4) return ng64 ? launch_march<double, AVX_COLOR_CAT_MERGE, 4, 4, 2, 64, 3, NF, true>(ctx, a, d, qc, s)
                                  : launch_march<double, AVX_COLOR_CAT_MERGE, 4, 4, 2, 128, 3, NF>(ctx, a, d, qc, s);
        return AVX_ERR_UNSUPPORTED;
    }
    switch (a.r) {
        // <R, SY, XPT, min waves/SIMD>, from A/B runs on MI355X: 2 columns per thread keeps the
        // row-window reads conflict-free (16-byte lane stride) and the register windows under 128 VGPRs.
        // SP: the 192-thread form runs wave-specialised (+ one producer wave that stages and decodes; full-width strips, the
        // producer takes two or three passes over a row pair).  Measured on for every float32 radius: wolf 165 -> 175, lion
        // 176 -> 203, squirrel 202 -> 236, dog 110 -> 120 GP/s.  The 384-thread form stays plain (one producer cannot feed
        // six compute waves: 0.56 vs 0.33 ms for squirrel) and remains the tuner's alternative.
#define AVX_MARCH_F32(RR, XX, MW, SP)                                                                        \
    case RR:                                                                                                 \
        return ng64 ? launch_march<float, AVX_COLOR_MATRIX, RR, 4, XX, 64, MW, NF, SP>(ctx, a, d, qc, s)      \
                    : launch_march<float, AVX_COLOR_MATRIX, RR, 4, XX, 128, MW, NF>(ctx, a, d, qc, s);
        AVX_MARCH_F32(1, 4, 3, true) AVX_MARCH_F32(3, 4, 3, true) AVX_MARCH_F32(4, 2, 3, true) AVX_MARCH_F32(5, 2, 3, true)
        AVX_MARCH_F32(6, 2, 3, true) AVX_MARCH_F32(7, 2, 3, true) AVX_MARCH_F32(8, 2, 3, true) AVX_MARCH_F32(9, 2, 3, true)
        AVX_MARCH_F32(14, 2, 2, true)
#undef AVX_MARCH_F32
        default: return AVX_ERR_UNSUPPORTED;
    }
}

int avx_launch_dichromat_march(avx_ctx* ctx, DichromatArgs& a, const avx_dichromat_desc* d, bool f64_cat, hipStream_t s) {
    // The kernel reads rows as aligned dwords without a tail guard: batches whose byte size or base address is not a multiple
    // of 4 (no standard video size) take the 2-D tiled / reference kernels instead.
    if (((((size_t)a.H * a.W * 3 * (size_t)a.n_frames) | (size_t)(uintptr_t)a.in) & 3u) != 0) return AVX_ERR_UNSUPPORTED;
    if ((size_t)a.H * a.W * 3 >= ((size_t)1 << 32) - 16) return AVX_ERR_UNSUPPORTED;  // 32-bit row offsets inside a frame
    if (a.H < 2 * (a.r + 4)) return AVX_ERR_UNSUPPORTED;                              // single-step row reflection (SY = 4)
    // Workgroup width: 192 threads (NG = 64) or 384 (NG = 128).  Which is faster depends on the radius, the element type
    // and the batch geometry (cat: 64; dog: 128; wolf at 1080p: 64 by 13 %), so like the row split it is measured on the
    // first call per (radius, type, batch, frame size) and remembered.  AVX_MARCH_NG=64|128 pins it.
    { const char* e = getenv("AVX_MARCH_NG"); if (e && *e) return march_dispatch(ctx, a, d, f64_cat, atoi(e) == 64, s); }
    const uint64_t key = width_key(f64_cat, a.r, a.n_frames, a.H, a.W);
    for (int i = 0; i < ctx->n_march_tuned; ++i)
        if (ctx->march_tuned[i].key == key) return march_dispatch(ctx, a, d, f64_cat, ctx->march_tuned[i].nchunks == 64, s);
    if ((size_t)a.n_frames * a.H * a.W < (size_t)256 * 1024) return march_dispatch(ctx, a, d, f64_cat, f64_cat, s);  // tiny: defaults
    hipEvent_t e0, e1;
    AVX_HIP(ctx, hipEventCreate(&e0));
    AVX_HIP(ctx, hipEventCreate(&e1));
    float ms[2] = {3.4e38f, 3.4e38f};
    int rc = AVX_OK;
    for (int v = 0; v < 2 && rc == AVX_OK; ++v) {
        const bool ng64 = v == 0;
        rc = march_dispatch(ctx, a, d, f64_cat, ng64, s);  // first call: tunes its own row split
        if (rc) break;
        AVX_HIP(ctx, hipEventRecord(e0, s));
        rc = march_dispatch(ctx, a, d, f64_cat, ng64, s);
        AVX_HIP(ctx, hipEventRecord(e1, s));
        AVX_HIP(ctx, hipEventSynchronize(e1));
        AVX_HIP(ctx, hipEventElapsedTime(&ms[v], e0, e1));
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    const bool pick64 = ms[0] <= ms[1];
    if (ctx->n_march_tuned < 64) ctx->march_tuned[ctx->n_march_tuned++] = {key, pick64 ? 64 : 128};
    if (getenv("AVX_TUNE_LOG")) fprintf(stderr, "[avx tune] width: f64=%d R=%d frames=%d H=%d W=%d -> NG=%d (%.3f vs %.3f ms)\n", (int)f64_cat, a.r, a.n_frames, a.H, a.W, pick64 ? 64 : 128, ms[0], ms[1]);
    return march_dispatch(ctx, a, d, f64_cat, pick64, s);
}
