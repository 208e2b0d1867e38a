// csrc/ew_common.h -- what the plane-program interpreter (ew.hip) and the program-specific kernels generated from recorded
// programs (ew_gen.hip, tools/gen_ew_kernels.py) share: the launch arguments, the accumulator rules and the reduction epilogue.
#pragma once
#include <cmath>
#include <cstdint>

#include "dichromat_common.h"

namespace avxew {

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
    int uses_encode, uses_xy;
    uint32_t* ticket; double* scalars_out; uint8_t acc_out_kind[AVX_EW_MAX_ACC]; int acc_slot[AVX_EW_MAX_ACC];
};

__device__ __forceinline__ float acc_init(int kind) { return kind == AVX_EW_ACC_MIN ? INFINITY : (kind == AVX_EW_ACC_MAX ? -INFINITY : 0.f); }
__device__ __forceinline__ float acc_merge(int kind, float a, float b) { return kind == AVX_EW_ACC_MIN ? fminf(a, b) : (kind == AVX_EW_ACC_MAX ? fmaxf(a, b) : a + b); }

// The tail of a program with reductions, as in k_ew: per-thread accumulators (acc[k][0..PX)) -> partial[block][k] -> the last
// workgroup (ticket) folds the per-block partials into the scalar table.  Same merge order as the interpreter, so a program gives
// the same scalars through either kernel.
template <int NACC, int PX>
__device__ __forceinline__ void ew_reduce_tail(const EwArgs& a, const float (&acc)[NACC][PX]) {
    __shared__ float red[kET / 64];
    __shared__ int is_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        const int kind = a.acc_kind[k];
        float v = acc[k][0];
#pragma unroll
        for (int q = 1; q < PX; ++q) v = acc_merge(kind, v, acc[k][q]);
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
    for (int k = wave; k < a.n_acc; k += kET / 64) {
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

// program-specific kernels: keyed by two 64-bit FNV-1a hashes of the program's structure bytes (ew_structure_hash in ew.hip)
typedef void (*EwKernel)(const EwArgs);
struct EwSpecEntry { uint64_t h1, h2; EwKernel k4, k8; };
extern const EwSpecEntry kEwSpec[];
extern const int kEwSpecCount;

}  // namespace avxew
