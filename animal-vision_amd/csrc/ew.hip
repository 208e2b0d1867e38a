// csrc/ew.hip -- fused elementwise stage of the UV stylisation stack.
//
// The 14 remaining UV species (animals/reindeer.py, goldfish.py, damselfish.py, ... ) are NumPy expression chains
// over H x W float32 planes: +,-,*,/, clip, where, sqrt/exp/sin/cos/atan2/pow, broadcasts of per-row / per-column
// vectors and of frame-wide scalars (min/max/mean/percentiles), with a Gaussian blur, a resize or a Sobel in
// between.  Executed literally, every NumPy operator is one HBM round trip (8-12 B/px each, 100-200 of them per
// frame).  Here a whole chain between two non-elementwise stages is ONE launch: the host lowers the chain to a short
// register program (<= 384 instructions, 32 float32 registers per pixel), the kernel interprets it per pixel (4 or 8
// pixels per thread: one instruction dispatch serves 256 or 512 pixels per wave) with
// wave-uniform control flow (the instruction stream lives in the kernel arguments, read through the scalar cache), the
// per-pixel registers live in LDS ([reg][lane]: conflict-free), and the only HBM traffic is the planes the chain
// really reads and the planes it really leaves behind.  Frame-wide reductions ride along as accumulator registers
// (deterministic two-level reduction; results land in a device-side scalar table that later programs read, so no
// value ever visits the host).
//
// Arithmetic contract: every instruction is one IEEE float32 operation, in the order NumPy evaluates the
// reference's expression (-ffp-contract=off: nothing fuses); transcendental functions are the device's (1-2 ulp
// from NumPy's): within 1e-4 relative of the reference, the UV path's bar.  uint8 encode = the exact threshold table.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "dichromat_common.h"
#include "ew_common.h"

using namespace avxk;

using namespace avxew;

namespace {

// The per-pixel register files live in VGPRs: four float[NREG] arrays (one per pixel of the thread), indexed by the
// wave-uniform register numbers of the instruction through s_set_gpr_idx (no LDS, no scratch).  NREG = 16 leaves
// room for 4 waves/SIMD, NREG = 32 for 2; the host picks the smallest that fits the program.
#define EW_FOR for (int k = 0; k < PX; ++k)
#define EW_RD(dst, idx) { dst[0] = R0[idx]; dst[1] = R1[idx]; dst[2] = R2[idx]; dst[3] = R3[idx]; if constexpr (PX == 8) { dst[4] = R4[idx]; dst[5] = R5[idx]; dst[6] = R6[idx]; dst[7] = R7[idx]; } }
#define EW_WR(idx, src) { R0[idx] = src[0]; R1[idx] = src[1]; R2[idx] = src[2]; R3[idx] = src[3]; if constexpr (PX == 8) { R4[idx] = src[4]; R5[idx] = src[5]; R6[idx] = src[6]; R7[idx] = src[7]; } }
#define EW_UN(expr) { _Pragma("unroll") EW_FOR { const float x = xs[k]; r[k] = (expr); } } break
#define EW_BIN(expr) { _Pragma("unroll") EW_FOR { const float x = xs[k], y = ys[k]; r[k] = (expr); } } break

template <int NREG, int PX>
__global__ __launch_bounds__(kET) void k_ew(const EwArgs a) {
    __shared__ float thr[256];
    __shared__ uint8_t coarse[kCoarseTableBytes];
    __shared__ float red[kET / 64];
    const int tid = threadIdx.x;
    if (a.uses_encode) {
        for (int i = tid; i < 256; i += kET) thr[i] = a.thr[i];
        for (int i = tid; i < kCoarseTableBytes; i += kET) coarse[i] = a.coarse[i];
        __syncthreads();
    }
    float R0[NREG], R1[NREG], R2[NREG], R3[NREG], R4[NREG], R5[NREG], R6[NREG], R7[NREG];  // R4..R7 vanish when PX == 4
#pragma unroll
    for (int i = 0; i < NREG; ++i) { R0[i] = 0.f; R1[i] = 0.f; R2[i] = 0.f; R3[i] = 0.f; R4[i] = 0.f; R5[i] = 0.f; R6[i] = 0.f; R7[i] = 0.f; }
    for (int k = 0; k < a.n_acc; ++k) {
        const float v = acc_init(a.acc_kind[k]);
        const int reg = a.acc_reg[k] & (NREG - 1);
        float vv[PX];
#pragma unroll
        EW_FOR vv[k] = v;
        EW_WR(reg, vv);
    }
    const uint32_t n = (uint32_t)a.n, W = (uint32_t)a.W;  // the host checks H*W < 2^31
    for (uint32_t base = blockIdx.x * (kET * PX); base < n; base += gridDim.x * (kET * PX)) {
        uint32_t ii[PX], px[PX], py[PX];
        bool valid[PX];
#pragma unroll
        EW_FOR {
            const uint32_t i = base + tid + k * kET;
            valid[k] = i < n;
            ii[k] = valid[k] ? i : n - 1;  // invalid lanes compute on the last pixel, store / accumulate nothing
            px[k] = py[k] = 0;
        }
        if (a.uses_xy) {
#pragma unroll
            EW_FOR { py[k] = ii[k] / W; px[k] = ii[k] - py[k] * W; }
        }
        // The instruction word of pc + 1 is fetched (scalar load from the kernel-argument segment) while pc executes: fetched at
        // the top of its own iteration, every instruction began with an exposed scalar-cache round trip.
        avx_ew_insn in_next = a.insn[0];
        for (int pc = 0; pc < a.n_insn; ++pc) {
            const avx_ew_insn in = in_next;
            in_next = a.insn[pc + 1 < a.n_insn ? pc + 1 : pc];
            const int ra = in.a & (NREG - 1), rb = in.b & (NREG - 1), rd = in.dst & (NREG - 1);
            const float imm = __uint_as_float(in.imm);
            const bool ia = in.op & AVX_EW_IMM_A, ib = in.op & AVX_EW_IMM_B;  // an operand is the immediate constant
            float xs[PX], ys[PX];
            EW_RD(xs, ra);
            EW_RD(ys, rb);
#pragma unroll
            EW_FOR { xs[k] = ia ? imm : xs[k]; ys[k] = ib ? imm : ys[k]; }
            float r[PX];
            switch (in.op & AVX_EW_OPCODE_MASK) {
                case AVX_EW_CONST: { const float c = __uint_as_float(in.imm); _Pragma("unroll") EW_FOR r[k] = c; } break;
                case AVX_EW_SCALAR: { const float c = (float)a.scalars[in.imm]; _Pragma("unroll") EW_FOR r[k] = c; } break;
                case AVX_EW_LOAD: {
                    const avx_ew_plane p = a.planes[in.imm];
                    if (p.kind == AVX_EW_PLANE_F32) { _Pragma("unroll") EW_FOR r[k] = ((const float*)p.ptr)[(size_t)ii[k] * p.stride]; }
                    else if (p.kind == AVX_EW_PLANE_U8) { _Pragma("unroll") EW_FOR r[k] = (float)((const uint8_t*)p.ptr)[(size_t)ii[k] * p.stride]; }
                    else if (p.kind == AVX_EW_PLANE_U8_LUT) { _Pragma("unroll") EW_FOR r[k] = a.lut[((const uint8_t*)p.ptr)[(size_t)ii[k] * p.stride]]; }
                    else if (p.kind == AVX_EW_PLANE_COL) { _Pragma("unroll") EW_FOR r[k] = ((const float*)p.ptr)[px[k]]; }
                    else { _Pragma("unroll") EW_FOR r[k] = ((const float*)p.ptr)[py[k]]; }
                    break;
                }
                case AVX_EW_STORE: {
                    const avx_ew_plane p = a.planes[in.imm];
                    if (p.kind == AVX_EW_PLANE_F32) {
#pragma unroll
                        EW_FOR if (valid[k]) ((float*)p.ptr)[(size_t)ii[k] * p.stride] = xs[k];
                    } else {
#pragma unroll
                        EW_FOR if (valid[k]) ((uint8_t*)p.ptr)[(size_t)ii[k] * p.stride] = (uint8_t)quantize_coarse<float, kCoarseNFix>(xs[k], thr, coarse, a.lo_key);
                    }
                    continue;
                }
                case AVX_EW_ADD: EW_BIN(x + y);
                case AVX_EW_SUB: EW_BIN(x - y);
                case AVX_EW_MUL: EW_BIN(x * y);
                case AVX_EW_DIV: EW_BIN(x / y);
                case AVX_EW_MIN: EW_BIN(fminf(x, y));
                case AVX_EW_MAX: EW_BIN(fmaxf(x, y));
                case AVX_EW_POW: EW_BIN(powf(x, y));
                case AVX_EW_ATAN2: EW_BIN(atan2f(x, y));
                case AVX_EW_NEG: EW_UN(-x);
                case AVX_EW_ABS: EW_UN(fabsf(x));
                case AVX_EW_SQRT: EW_UN(__fsqrt_rn(x));
                case AVX_EW_EXP: EW_UN(expf(x));
                case AVX_EW_LOG: EW_UN(logf(x));
                case AVX_EW_SIN: EW_UN(sinf(x));
                case AVX_EW_COS: EW_UN(cosf(x));
                case AVX_EW_FLOOR: EW_UN(floorf(x));
                case AVX_EW_CEIL: EW_UN(ceilf(x));
                case AVX_EW_CLIP01: EW_UN(x < 0.f ? 0.f : (x > 1.f ? 1.f : x));
                case AVX_EW_TANH: EW_UN(tanhf(x));
                case AVX_EW_LT: EW_BIN(x < y ? 1.f : 0.f);
                case AVX_EW_LE: EW_BIN(x <= y ? 1.f : 0.f);
                case AVX_EW_GT: EW_BIN(x > y ? 1.f : 0.f);
                case AVX_EW_GE: EW_BIN(x >= y ? 1.f : 0.f);
                case AVX_EW_EQ: EW_BIN(x == y ? 1.f : 0.f);
                case AVX_EW_AND: EW_BIN((x != 0.f && y != 0.f) ? 1.f : 0.f);
                case AVX_EW_OR: EW_BIN((x != 0.f || y != 0.f) ? 1.f : 0.f);
                case AVX_EW_NOT: EW_UN(x != 0.f ? 0.f : 1.f);
                case AVX_EW_SELECT: {
                    const int rc = in.imm & (NREG - 1);
                    float zs[PX];
                    EW_RD(zs, rc);
#pragma unroll
                    EW_FOR r[k] = xs[k] != 0.f ? ys[k] : zs[k];
                    break;
                }
                case AVX_EW_ACCMIN: case AVX_EW_ACCMAX: case AVX_EW_ACCSUM: {
                    const int opc = in.op & AVX_EW_OPCODE_MASK;
                    const int kind = opc == AVX_EW_ACCMIN ? AVX_EW_ACC_MIN : (opc == AVX_EW_ACCMAX ? AVX_EW_ACC_MAX : AVX_EW_ACC_SUM);
                    float cur[PX];
                    EW_RD(cur, rd);
#pragma unroll
                    EW_FOR r[k] = valid[k] ? acc_merge(kind, cur[k], xs[k]) : cur[k];
                    break;
                }
                default: { _Pragma("unroll") EW_FOR r[k] = 0.f; } break;
            }
            EW_WR(rd, r);
        }
    }
    // block-level reduction of the accumulators -> partial[block][k]; the last workgroup to finish (ticket) folds the
    // per-block partials into the scalar table: min / max, or the sum in double over the float32 partial sums
    // (AVX_EW_ACC_MEAN = float32(sum) / n like ndarray.mean() of a float32 array).  Partials travel in agent-scope
    // atomic stores / loads: no cache write-back or invalidate is needed (see k_sel_pass in uv.hip).
    if (a.n_acc == 0) return;
    const int lane = tid & 63, wave = tid >> 6;
    __shared__ int is_last;
    for (int k = 0; k < a.n_acc; ++k) {
        const int kind = a.acc_kind[k], reg = a.acc_reg[k] & (NREG - 1);
        float fin[PX];
        EW_RD(fin, reg);
        float v = fin[0];
#pragma unroll
        for (int q = 1; q < PX; ++q) v = acc_merge(kind, v, fin[q]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = acc_merge(kind, v, __shfl_xor(v, o));
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        if (tid == 0) {
            float t = red[0];
            for (int w = 1; w < kET / 64; ++w) t = acc_merge(kind, t, red[w]);
            __hip_atomic_store(&a.partial[(size_t)blockIdx.x * a.n_acc + k], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        is_last = avxk::ticket_is_last(a.ticket, blockIdx.x, gridDim.x);
    }
    __syncthreads();
    if (!is_last) return;
    for (int k = wave; k < a.n_acc; k += kET / 64) {  // one wave per accumulator; 8 independent (uncached) loads in flight per lane
        const int kind = a.acc_out_kind[k];
        const bool mm = kind == AVX_EW_ACC_MIN || kind == AVX_EW_ACC_MAX;
        const float init = kind == AVX_EW_ACC_MIN ? INFINITY : (kind == AVX_EW_ACC_MAX ? -INFINITY : 0.f);
        float vm = init;
        double vs = 0.0;
        for (int i0 = lane; i0 < (int)gridDim.x; i0 += 64 * 8) {
            float vals[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + 64 * j;
                vals[j] = i < (int)gridDim.x ? __hip_atomic_load(&a.partial[(size_t)i * a.n_acc + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : init;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (mm) vm = acc_merge(kind, vm, vals[j]);
                else vs += (double)vals[j];
            }
        }
        if (mm) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) vm = acc_merge(kind, vm, __shfl_xor(vm, o));
            if (lane == 0) a.scalars_out[a.acc_slot[k]] = (double)vm;
        } else {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) vs += __shfl_xor(vs, o);
            if (lane == 0) a.scalars_out[a.acc_slot[k]] = kind == AVX_EW_ACC_MEAN ? (double)((float)vs / (float)a.n) : (double)(float)vs;
        }
    }

}

}  // namespace

static std::atomic<unsigned long long> g_spec_hits{0}, g_spec_misses{0};

extern "C" int avx_ew_spec_stats(unsigned long long* hits, unsigned long long* misses) {
    if (hits) *hits = g_spec_hits.load();
    if (misses) *misses = g_spec_misses.load();
    return avxew::kEwSpecCount;
}

extern "C" int avx_ew_run(avx_ctx* ctx, const avx_ew_program* p, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, p && p->struct_size == sizeof(avx_ew_program), "avx_ew_run: program is NULL or struct_size mismatch");
    AVX_REQUIRE(ctx, p->n_insn >= 1 && p->n_insn <= AVX_EW_MAX_INSN && p->insn_host, "avx_ew_run: instruction count out of range");
    AVX_REQUIRE(ctx, p->n_planes >= 0 && p->n_planes <= AVX_EW_MAX_PLANES && (p->n_planes == 0 || p->planes_host), "avx_ew_run: plane count out of range");
    AVX_REQUIRE(ctx, p->n_acc >= 0 && p->n_acc <= AVX_EW_MAX_ACC, "avx_ew_run: accumulator count out of range");
    AVX_REQUIRE(ctx, p->H > 0 && p->W > 0, "avx_ew_run: bad frame size");
    AVX_REQUIRE(ctx, p->n_acc == 0 || p->scalars_dev, "avx_ew_run: reductions need the scalar table");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    avx_ws* ws = avx_workspace(ctx, s);
    if (!ws) return AVX_ERR_NOMEM;
    EwArgs a;
    a.n_insn = p->n_insn; a.n_acc = p->n_acc; a.W = p->W; a.n = (size_t)p->H * p->W;
    a.uses_encode = 0; a.uses_xy = 0;
    AVX_REQUIRE(ctx, (size_t)p->H * p->W < ((size_t)1 << 31), "avx_ew_run: frame too large");
    int max_reg = 0;
    for (int i = 0; i < p->n_planes; ++i) {
        a.planes[i] = p->planes_host[i];
        AVX_REQUIRE(ctx, a.planes[i].ptr && a.planes[i].stride >= 1 && a.planes[i].kind >= 0 && a.planes[i].kind <= AVX_EW_PLANE_U8_ENC, "avx_ew_run: bad plane entry");
    }
    for (int i = 0; i < p->n_insn; ++i) {
        const avx_ew_insn in = p->insn_host[i];
        AVX_REQUIRE(ctx, in.dst < AVX_EW_MAX_REGS && in.a < AVX_EW_MAX_REGS && in.b < AVX_EW_MAX_REGS, "avx_ew_run: register index out of range");
        const int opc = in.op & AVX_EW_OPCODE_MASK;
        if ((in.op & (AVX_EW_IMM_A | AVX_EW_IMM_B)) != 0)
            AVX_REQUIRE(ctx, opc >= AVX_EW_ADD && opc <= AVX_EW_OR && opc != AVX_EW_NOT && (in.op & (AVX_EW_IMM_A | AVX_EW_IMM_B)) != (AVX_EW_IMM_A | AVX_EW_IMM_B),
                        "avx_ew_run: immediate operand on an opcode that takes none");
        if (opc == AVX_EW_LOAD || opc == AVX_EW_STORE) {
            AVX_REQUIRE(ctx, (int)in.imm < p->n_planes, "avx_ew_run: plane index out of range");
            const int kind = a.planes[in.imm].kind;
            if (kind == AVX_EW_PLANE_COL || kind == AVX_EW_PLANE_ROW) a.uses_xy = 1;
            if (opc == AVX_EW_STORE) {
                AVX_REQUIRE(ctx, kind == AVX_EW_PLANE_F32 || kind == AVX_EW_PLANE_U8_ENC, "avx_ew_run: store to a read-only plane kind");
                if (kind == AVX_EW_PLANE_U8_ENC) a.uses_encode = 1;
            } else {
                AVX_REQUIRE(ctx, kind != AVX_EW_PLANE_U8_ENC, "avx_ew_run: load from an encode-only plane");
            }
        }
        if (opc == AVX_EW_SCALAR) AVX_REQUIRE(ctx, p->scalars_dev && (int)in.imm < p->n_scalars, "avx_ew_run: scalar slot out of range");
        if (opc == AVX_EW_SELECT) AVX_REQUIRE(ctx, (in.imm & 0xff) < AVX_EW_MAX_REGS, "avx_ew_run: register index out of range");
        AVX_REQUIRE(ctx, opc >= AVX_EW_CONST && opc <= AVX_EW_ACCSUM, "avx_ew_run: unknown opcode");
        a.insn[i] = in;
        max_reg = max_reg > in.dst ? max_reg : in.dst;
        max_reg = max_reg > in.a ? max_reg : in.a;
        max_reg = max_reg > in.b ? max_reg : in.b;
        if (opc == AVX_EW_SELECT) max_reg = max_reg > (int)(in.imm & 0xff) ? max_reg : (int)(in.imm & 0xff);
    }
    for (int k = 0; k < p->n_acc; ++k) {
        const int reg = p->acc_host[3 * k], kind = p->acc_host[3 * k + 1], slot = p->acc_host[3 * k + 2];
        AVX_REQUIRE(ctx, reg >= 0 && reg < AVX_EW_MAX_REGS && kind >= AVX_EW_ACC_MIN && kind <= AVX_EW_ACC_MEAN && slot >= 0 && slot < p->n_scalars, "avx_ew_run: bad accumulator entry");
        a.acc_reg[k] = (uint8_t)reg; a.acc_kind[k] = (uint8_t)kind;
        max_reg = max_reg > reg ? max_reg : reg;
        a.acc_out_kind[k] = (uint8_t)kind; a.acc_slot[k] = slot;
    }
    if (const char* dump = getenv("AVX_EW_DUMP")) {  // tools/gen_ew_kernels.py: the structure of every program that runs (one line each)
        if (FILE* f = fopen(dump, "a")) {
            fprintf(f, "%d %d %d :", p->n_insn, p->n_planes, p->n_acc);
            for (int i = 0; i < p->n_insn; ++i) {
                const avx_ew_insn in = a.insn[i];
                const int opc = in.op & AVX_EW_OPCODE_MASK;
                const unsigned simm = (opc == AVX_EW_LOAD || opc == AVX_EW_STORE) ? in.imm : (opc == AVX_EW_SELECT ? (in.imm & 0xff) : 0u);
                fprintf(f, " %u,%u,%u,%u,%u", in.op, in.dst, in.a, in.b, simm);
            }
            fprintf(f, " :");
            for (int i = 0; i < p->n_planes; ++i) fprintf(f, " %d", a.planes[i].kind);
            fprintf(f, " :");
            for (int k = 0; k < p->n_acc; ++k) fprintf(f, " %d,%d", a.acc_reg[k], a.acc_kind[k]);
            fprintf(f, "\n");
            fclose(f);
        }
    }
    // programs with reductions: fewer, fatter workgroups (the last one folds gridDim partials per accumulator)
    const int px = (max_reg < 16 && a.n >= (size_t)1 << 20 && !getenv("AVX_EW_PX4")) ? 8 : 4;  // big frames, small programs: 8 pixels per thread
    const size_t want = (a.n + (size_t)kET * px - 1) / ((size_t)kET * px), cap = (size_t)ctx->num_cus * (p->n_acc ? 4 : 8);
    const int grid = (int)(want < cap ? want : cap);
    if (p->n_acc && ws->d_ew == nullptr) {  // per-block partials + the ticket (zero at rest)
        const size_t full = (size_t)ctx->num_cus * 8 * AVX_EW_MAX_ACC * sizeof(float) + 256;
        AVX_HIP(ctx, hipMalloc(&ws->d_ew, full));
        AVX_HIP(ctx, hipMemsetAsync(ws->d_ew, 0, full, s));
        ws->ew_cap = full;
    }
    a.scalars = p->scalars_dev; a.partial = (float*)ws->d_ew + 64; a.ticket = (uint32_t*)ws->d_ew; a.scalars_out = p->scalars_dev;
    a.lut = ctx->d_decode_lut; a.thr = ctx->d_enc_thr_f32; a.coarse = ctx->d_coarse_f32; a.lo_key = ctx->coarse_lo_key[0];
    // A program whose structure was recorded when ew_gen.hip was generated runs as its own straight-line kernel (same arithmetic,
    // registers and loads scheduled by the compiler instead of an instruction at a time); anything else is interpreted.
    // AVX_EW_NO_SPEC=1 pins the interpreter.
    if (!getenv("AVX_EW_NO_SPEC")) {
        uint64_t h1 = 0xcbf29ce484222325ull, h2 = 0x84222325cbf29ce4ull;
        auto mix = [&](unsigned byte) { h1 = (h1 ^ byte) * 0x100000001b3ull; h2 = (h2 ^ (byte + 0x9e)) * 0x100000001b3ull; h2 ^= h2 >> 29; };
        for (int i = 0; i < p->n_insn; ++i) {
            const avx_ew_insn in = a.insn[i];
            const int opc = in.op & AVX_EW_OPCODE_MASK;
            const unsigned simm = (opc == AVX_EW_LOAD || opc == AVX_EW_STORE) ? in.imm : (opc == AVX_EW_SELECT ? (in.imm & 0xff) : 0u);
            mix(in.op); mix(in.dst); mix(in.a); mix(in.b); mix(simm & 0xff);
        }
        mix(0xff);
        for (int i = 0; i < p->n_planes; ++i) mix((unsigned)a.planes[i].kind);
        mix(0xfe);
        for (int k = 0; k < p->n_acc; ++k) { mix(a.acc_reg[k]); mix(a.acc_kind[k]); }
        int lo = 0, hi = kEwSpecCount - 1;
        while (lo <= hi) {
            const int mid = (lo + hi) / 2;
            const EwSpecEntry& e = kEwSpec[mid];
            if (e.h1 == h1 && e.h2 == h2) {
                ++g_spec_hits;
                hipLaunchKernelGGL(px == 8 ? e.k8 : e.k4, dim3(grid), dim3(kET), 0, s, a);
                AVX_HIP(ctx, hipGetLastError());
                return AVX_OK;
            }
            if (e.h1 < h1 || (e.h1 == h1 && e.h2 < h2)) lo = mid + 1; else hi = mid - 1;
        }
        ++g_spec_misses;
    }
    if (max_reg < 16 && px == 8) hipLaunchKernelGGL((k_ew<16, 8>), dim3(grid), dim3(kET), 0, s, a);
    else if (max_reg < 16) hipLaunchKernelGGL((k_ew<16, 4>), dim3(grid), dim3(kET), 0, s, a);
    else hipLaunchKernelGGL((k_ew<32, 4>), dim3(grid), dim3(kET), 0, s, a);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}
