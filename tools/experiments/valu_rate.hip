// VALU issue-rate probe for gfx950: cycles a wave64 instruction holds its SIMD, per opcode class.
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
            if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 3) asm volatile("v_med3_f32 %0, %0, %0, %0" : "+v"(a[i]));
            if (OP == 4) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(a[i]));
            if (OP == 5) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
            if (OP == 6) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            if (OP == 7) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
__global__ __launch_bounds__(256) void kp(float* out, int iters, float seed) {
    f2 a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = f2{seed + threadIdx.x * 1e-3f + i, seed};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
            if (OP == 1) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(a[i]));
            if (OP == 2) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a[i]));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K>
static void run(const char* name, K kern, float* d) {
    const int blocks = 256 * 4, iters = 4096;  // 4 workgroups per CU = 4 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 16, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves x iters x 8 instructions
    const double instr = 4.0 * iters * 8;
    printf("%-22s %8.3f ms  -> %.2f cycles per wave64 instruction per SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / instr);
}
int main() {
    float* d;
    hipMalloc(&d, 256 * 4 * 256 * 4);
    run("v_fma_f32", k<0>, d); run("v_exp_f32", k<1>, d); run("v_rcp_f32", k<2>, d); run("v_med3_f32", k<3>, d);
    run("v_cvt_pk_f16_f32", k<4>, d); run("v_rsq_f32", k<5>, d); run("v_sqrt_f32", k<6>, d); run("v_log_f32", k<7>, d);
    run("v_pk_fma_f32", kp<0>, d); run("v_pk_mul_f32", kp<1>, d); run("v_fma_f64", kp<2>, d);
    return 0;
}
