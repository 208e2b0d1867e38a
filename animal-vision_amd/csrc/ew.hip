// csrc/ew.hip -- fused elementwise stage of the UV stylisation stack.
//
// The 14 remaining UV species (animals/reindeer.py, goldfish.py, damselfish.py, ... ) are NumPy expression chains
// over H x W float32 planes: +,-,*,/, clip, where, sqrt/exp/sin/cos/atan2/pow, broadcasts of per-row / per-column
// vectors and of frame-wide scalars (min/max/mean/percentiles), with a Gaussian blur, a resize or a Sobel in
// between.  Executed literally, every NumPy operator is one HBM round trip (8-12 B/px each, 100-200 of them per
// frame).  Here a whole chain between two non-elementwise stages is ONE launch: the host lowers the chain to a short
// register program (<= 384 instructions, 32 float32 registers per pixel), the kernel interprets it per pixel with
// wave-uniform control flow (the instruction stream lives in the kernel arguments, read through the scalar cache), the
// per-pixel registers live in LDS ([reg][lane]: conflict-free), and the only HBM traffic is the planes the chain
// really reads and the planes it really leaves behind.  Frame-wide reductions ride along as accumulator registers
// (deterministic two-level reduction; results land in a device-side scalar table that later programs read, so no
// value ever visits the host).
//
// Arithmetic contract: every instruction is one IEEE float32 operation, in the order NumPy evaluates the
// reference's expression (-ffp-contract=off: nothing fuses); transcendental functions are the device's (1-2 ulp
// from NumPy's): within 1e-4 relative of the reference, the UV path's bar.  uint8 encode = the exact threshold table.
#include <cmath>

#include "dichromat_common.h"

using namespace avxk;

namespace {

constexpr int kET = 256;

struct EwArgs {
    avx_ew_insn insn[AVX_EW_MAX_INSN];
    avx_ew_plane planes[AVX_EW_MAX_PLANES];
    uint8_t acc_reg[AVX_EW_MAX_ACC], acc_kind[AVX_EW_MAX_ACC];
    int n_insn, n_acc;
    int W;
    size_t n;
    const double* scalars;
    float* partial;  // [block][n_acc]
    const float* lut; const float* thr; const uint8_t* coarse; uint32_t lo_key;
    int uses_encode;
};

__device__ __forceinline__ float acc_init(int kind) { return kind == AVX_EW_ACC_MIN ? INFINITY : (kind == AVX_EW_ACC_MAX ? -INFINITY : 0.f); }

__global__ __launch_bounds__(kET) void k_ew(const EwArgs a) {
    __shared__ float R[AVX_EW_MAX_REGS][kET];
    __shared__ float thr[256];
    __shared__ uint8_t coarse[1024];
    __shared__ float red[kET / 64];
    const int tid = threadIdx.x;
    if (a.uses_encode) {
        for (int i = tid; i < 256; i += kET) thr[i] = a.thr[i];
        for (int i = tid; i < 1024; i += kET) coarse[i] = a.coarse[i];
        __syncthreads();
    }
    for (int k = 0; k < a.n_acc; ++k) R[a.acc_reg[k]][tid] = acc_init(a.acc_kind[k]);
    for (size_t base = (size_t)blockIdx.x * kET; base < a.n; base += (size_t)gridDim.x * kET) {
        const size_t i = base + tid;
        const bool valid = i < a.n;
        const size_t ii = valid ? i : a.n - 1;  // invalid lanes compute on the last pixel, store / accumulate nothing
        for (int pc = 0; pc < a.n_insn; ++pc) {
            const avx_ew_insn in = a.insn[pc];
            const float x = R[in.a][tid], y = R[in.b][tid];
            float r;
            switch (in.op) {
                case AVX_EW_CONST: r = __uint_as_float(in.imm); break;
                case AVX_EW_SCALAR: r = (float)a.scalars[in.imm]; break;
                case AVX_EW_LOAD: {
                    const avx_ew_plane p = a.planes[in.imm];
                    if (p.kind == AVX_EW_PLANE_F32) r = ((const float*)p.ptr)[ii * (size_t)p.stride];
                    else if (p.kind == AVX_EW_PLANE_U8) r = (float)((const uint8_t*)p.ptr)[ii * (size_t)p.stride];
                    else if (p.kind == AVX_EW_PLANE_U8_LUT) r = a.lut[((const uint8_t*)p.ptr)[ii * (size_t)p.stride]];
                    else if (p.kind == AVX_EW_PLANE_COL) r = ((const float*)p.ptr)[ii % (size_t)a.W];
                    else r = ((const float*)p.ptr)[ii / (size_t)a.W];
                    break;
                }
                case AVX_EW_STORE: {
                    const avx_ew_plane p = a.planes[in.imm];
                    if (valid) {
                        if (p.kind == AVX_EW_PLANE_F32) ((float*)p.ptr)[i * (size_t)p.stride] = x;
                        else ((uint8_t*)p.ptr)[i * (size_t)p.stride] = (uint8_t)quantize_coarse<float, 2>(x, thr, coarse, a.lo_key);
                    }
                    continue;
                }
                case AVX_EW_ADD: r = x + y; break;
                case AVX_EW_SUB: r = x - y; break;
                case AVX_EW_MUL: r = x * y; break;
                case AVX_EW_DIV: r = x / y; break;
                case AVX_EW_MIN: r = fminf(x, y); break;
                case AVX_EW_MAX: r = fmaxf(x, y); break;
                case AVX_EW_POW: r = powf(x, y); break;
                case AVX_EW_ATAN2: r = atan2f(x, y); break;
                case AVX_EW_NEG: r = -x; break;
                case AVX_EW_ABS: r = fabsf(x); break;
                case AVX_EW_SQRT: r = __fsqrt_rn(x); break;
                case AVX_EW_EXP: r = expf(x); break;
                case AVX_EW_LOG: r = logf(x); break;
                case AVX_EW_SIN: r = sinf(x); break;
                case AVX_EW_COS: r = cosf(x); break;
                case AVX_EW_FLOOR: r = floorf(x); break;
                case AVX_EW_CEIL: r = ceilf(x); break;
                case AVX_EW_CLIP01: r = x < 0.f ? 0.f : (x > 1.f ? 1.f : x); break;
                case AVX_EW_TANH: r = tanhf(x); break;
                case AVX_EW_LT: r = x < y ? 1.f : 0.f; break;
                case AVX_EW_LE: r = x <= y ? 1.f : 0.f; break;
                case AVX_EW_GT: r = x > y ? 1.f : 0.f; break;
                case AVX_EW_GE: r = x >= y ? 1.f : 0.f; break;
                case AVX_EW_EQ: r = x == y ? 1.f : 0.f; break;
                case AVX_EW_AND: r = (x != 0.f && y != 0.f) ? 1.f : 0.f; break;
                case AVX_EW_OR: r = (x != 0.f || y != 0.f) ? 1.f : 0.f; break;
                case AVX_EW_NOT: r = x != 0.f ? 0.f : 1.f; break;
                case AVX_EW_SELECT: r = x != 0.f ? y : R[in.imm & 0xff][tid]; break;
                case AVX_EW_ACCMIN: if (valid) R[in.dst][tid] = fminf(R[in.dst][tid], x); continue;
                case AVX_EW_ACCMAX: if (valid) R[in.dst][tid] = fmaxf(R[in.dst][tid], x); continue;
                case AVX_EW_ACCSUM: if (valid) R[in.dst][tid] = R[in.dst][tid] + x; continue;
                default: r = 0.f; break;
            }
            R[in.dst][tid] = r;
        }
    }
    // block-level reduction of the accumulators -> partial[block][k]
    const int lane = tid & 63, wave = tid >> 6;
    for (int k = 0; k < a.n_acc; ++k) {
        float v = R[a.acc_reg[k]][tid];
        const int kind = a.acc_kind[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float w = __shfl_xor(v, o);
            v = kind == AVX_EW_ACC_MIN ? fminf(v, w) : (kind == AVX_EW_ACC_MAX ? fmaxf(v, w) : v + w);
        }
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        if (tid == 0) {
            float t = red[0];
            for (int w = 1; w < kET / 64; ++w) t = kind == AVX_EW_ACC_MIN ? fminf(t, red[w]) : (kind == AVX_EW_ACC_MAX ? fmaxf(t, red[w]) : t + red[w]);
            a.partial[(size_t)blockIdx.x * a.n_acc + k] = t;
        }
    }
}

struct FinalArgs { const float* partial; int nblocks, n_acc; uint8_t kind[AVX_EW_MAX_ACC]; int slot[AVX_EW_MAX_ACC]; double* scalars; double n; };
// one wave per accumulator: min / max, or the sum in double over the per-block float32 partial sums;
// AVX_EW_ACC_MEAN = float32(sum) / n like ndarray.mean() of a float32 array
__global__ void k_ew_final(const FinalArgs a) {
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (k >= a.n_acc) return;
    const int kind = a.kind[k];
    if (kind == AVX_EW_ACC_MIN || kind == AVX_EW_ACC_MAX) {
        float v = kind == AVX_EW_ACC_MIN ? INFINITY : -INFINITY;
        for (int i = lane; i < a.nblocks; i += 64) { const float w = a.partial[(size_t)i * a.n_acc + k]; v = kind == AVX_EW_ACC_MIN ? fminf(v, w) : fmaxf(v, w); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const float w = __shfl_xor(v, o); v = kind == AVX_EW_ACC_MIN ? fminf(v, w) : fmaxf(v, w); }
        if (lane == 0) a.scalars[a.slot[k]] = (double)v;
    } else {
        double v = 0.0;
        for (int i = lane; i < a.nblocks; i += 64) v += (double)a.partial[(size_t)i * a.n_acc + k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) a.scalars[a.slot[k]] = kind == AVX_EW_ACC_MEAN ? (double)((float)v / (float)a.n) : (double)(float)v;
    }
}

}  // namespace

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
    a.uses_encode = 0;
    for (int i = 0; i < p->n_planes; ++i) {
        a.planes[i] = p->planes_host[i];
        AVX_REQUIRE(ctx, a.planes[i].ptr && a.planes[i].stride >= 1 && a.planes[i].kind >= 0 && a.planes[i].kind <= AVX_EW_PLANE_U8_ENC, "avx_ew_run: bad plane entry");
    }
    for (int i = 0; i < p->n_insn; ++i) {
        const avx_ew_insn in = p->insn_host[i];
        AVX_REQUIRE(ctx, in.dst < AVX_EW_MAX_REGS && in.a < AVX_EW_MAX_REGS && in.b < AVX_EW_MAX_REGS, "avx_ew_run: register index out of range");
        if (in.op == AVX_EW_LOAD || in.op == AVX_EW_STORE) {
            AVX_REQUIRE(ctx, (int)in.imm < p->n_planes, "avx_ew_run: plane index out of range");
            const int kind = a.planes[in.imm].kind;
            if (in.op == AVX_EW_STORE) {
                AVX_REQUIRE(ctx, kind == AVX_EW_PLANE_F32 || kind == AVX_EW_PLANE_U8_ENC, "avx_ew_run: store to a read-only plane kind");
                if (kind == AVX_EW_PLANE_U8_ENC) a.uses_encode = 1;
            } else {
                AVX_REQUIRE(ctx, kind != AVX_EW_PLANE_U8_ENC, "avx_ew_run: load from an encode-only plane");
            }
        }
        if (in.op == AVX_EW_SCALAR) AVX_REQUIRE(ctx, p->scalars_dev && (int)in.imm < p->n_scalars, "avx_ew_run: scalar slot out of range");
        if (in.op == AVX_EW_SELECT) AVX_REQUIRE(ctx, (in.imm & 0xff) < AVX_EW_MAX_REGS, "avx_ew_run: register index out of range");
        AVX_REQUIRE(ctx, in.op <= AVX_EW_ACCSUM, "avx_ew_run: unknown opcode");
        a.insn[i] = in;
    }
    FinalArgs f{};
    for (int k = 0; k < p->n_acc; ++k) {
        const int reg = p->acc_host[3 * k], kind = p->acc_host[3 * k + 1], slot = p->acc_host[3 * k + 2];
        AVX_REQUIRE(ctx, reg >= 0 && reg < AVX_EW_MAX_REGS && kind >= AVX_EW_ACC_MIN && kind <= AVX_EW_ACC_MEAN && slot >= 0 && slot < p->n_scalars, "avx_ew_run: bad accumulator entry");
        a.acc_reg[k] = (uint8_t)reg; a.acc_kind[k] = (uint8_t)kind;
        f.kind[k] = (uint8_t)kind; f.slot[k] = slot;
    }
    const size_t want = (a.n + kET - 1) / kET, cap = (size_t)ctx->num_cus * 8;
    const int grid = (int)(want < cap ? want : cap);
    if (p->n_acc) {
        const size_t need = (size_t)grid * p->n_acc * sizeof(float);
        if (ws->ew_cap < need) {
            if (ws->d_ew) { AVX_HIP(ctx, hipStreamSynchronize(s)); AVX_HIP(ctx, hipFree(ws->d_ew)); }
            ws->d_ew = nullptr; ws->ew_cap = 0;
            const size_t full = (size_t)ctx->num_cus * 8 * AVX_EW_MAX_ACC * sizeof(float);
            AVX_HIP(ctx, hipMalloc(&ws->d_ew, full));
            ws->ew_cap = full;
        }
    }
    a.scalars = p->scalars_dev; a.partial = (float*)ws->d_ew;
    a.lut = ctx->d_decode_lut; a.thr = ctx->d_enc_thr_f32; a.coarse = ctx->d_coarse_f32; a.lo_key = ctx->coarse_lo_key[0];
    hipLaunchKernelGGL(k_ew, dim3(grid), dim3(kET), 0, s, a);
    if (p->n_acc) {
        f.partial = a.partial; f.nblocks = grid; f.n_acc = p->n_acc; f.scalars = p->scalars_dev; f.n = (double)a.n;
        hipLaunchKernelGGL(k_ew_final, dim3(1), dim3(64 * p->n_acc), 0, s, f);
    }
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}
