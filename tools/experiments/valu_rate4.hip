// VALU issue-rate probe, part 4: whole-kernel time (HIP events), workgroups pinned to `w` per CU by their LDS size,
// so a SIMD holds exactly w waves; 8 or 16 independent chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP, int NCH>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    extern __shared__ float lds[];
    float a[NCH];
    f2 p[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; p[i] = f2{a[i], seed}; }
    if (seed < 0) lds[threadIdx.x] = seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
            if (OP == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[(i + 1) % NCH]), "v"(a[(i + 2) % NCH]));
            if (OP == 2) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a[i]));
            if (OP == 3) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
            if (OP == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(p[(i + 1) % NCH]), "v"(p[(i + 2) % NCH]));
            if (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[i]));
            if (OP == 6) asm volatile("v_fma_mix_f32 %0, %0, %0, %0" : "+v"(a[i]));
            if (OP == 7) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 9) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(a[i]));
            if (OP == 10) asm volatile("v_med3_f32 %0, %0, %0, %0" : "+v"(a[i]));
            if (OP == 11) asm volatile("v_pk_fma_f16 %0, %0, %0, %0" : "+v"(a[i]));
            if (OP == 12) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[(i + 1) % NCH]), "v"(a[(i + 2) % NCH]));
            if (OP == 13) asm volatile("v_pk_mul_f16 %0, %0, %0" : "+v"(a[i]));
            if (OP == 14) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[(i + 1) % NCH]), "v"(a[(i + 2) % NCH]));
            if (OP == 15) asm volatile("v_pk_max_f16 %0, %0, %0" : "+v"(a[i]));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K>
static double run(K kern, float* d, int w, int nch) {
    const int blocks = 256 * w, iters = 8192;
    const int lds = (w == 8 ? 18 : w == 4 ? 36 : w == 2 ? 72 : 150) * 1024;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d, 16, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / ((double)w * iters * nch);  // ns per instruction per SIMD
}
#define ROW(name, OP)                                                                                                        \
    printf("%-26s", name);                                                                                                   \
    for (int w : {1, 2, 4, 8}) printf("  w%d: %5.2f / %5.2f", w, run(k<OP, 8>, d, w, 8), run(k<OP, 16>, d, w, 16));          \
    printf("\n");
int main() {
    float* d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    printf("ns per wave64 instruction per SIMD (8 chains / 16 chains per wave), w waves per SIMD\n");
    ROW("v_fma_f32 (same reg)", 0) ROW("v_fma_f32 (3 regs)", 1) ROW("v_mul_f32", 2) ROW("v_pk_fma_f32 (same reg)", 3) ROW("v_pk_fma_f32 (3 regs)", 4)
    ROW("v_pk_mul_f32", 5) ROW("v_fma_mix_f32", 6) ROW("v_exp_f32", 7) ROW("v_rcp_f32", 8) ROW("v_cvt_pk_f16_f32", 9) ROW("v_med3_f32", 10)
    ROW("v_pk_fma_f16 (same reg)", 11) ROW("v_pk_fma_f16 (3 regs)", 12) ROW("v_pk_mul_f16", 13) ROW("v_dot2_f32_f16", 14) ROW("v_pk_max_f16", 15)
    return 0;
}
