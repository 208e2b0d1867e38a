// VALU issue-rate probe, part 3: 8 vs 16 independent chains per wave, 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP, int NCH>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters, float seed) {
    float a[NCH];
    f2 p[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; p[i] = f2{a[i], seed}; }
    __syncthreads();
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
            if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
            if (OP == 3) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a[i]));
        }
    }
    const long long c1 = clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = c1 - c0;
}
template <typename K>
static void run(const char* name, K kern, float* d, long long* dc, int wgs_per_cu, int nch) {
    const int blocks = 256 * wgs_per_cu, iters = 2048;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, dc, 16, 1.0f);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, dc, iters, 1.0f);
    hipDeviceSynchronize();
    long long c[1];
    hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
    printf("%-14s chains %2d waves/SIMD %d: %.2f cycles / instr / SIMD\n", name, nch, wgs_per_cu, c[0] / ((double)wgs_per_cu * iters * nch));
}
int main() {
    float* d; long long* dc;
    hipMalloc(&d, 256 * 8 * 256 * 4); hipMalloc(&dc, 256 * 8 * 8);
    for (int w : {1, 2, 3, 4, 6, 8}) {
        run("v_fma_f32", k<0, 8>, d, dc, w, 8); run("v_fma_f32", k<0, 16>, d, dc, w, 16);
        run("v_mul_f32", k<3, 8>, d, dc, w, 8); run("v_mul_f32", k<3, 16>, d, dc, w, 16);
        run("v_pk_fma_f32", k<2, 8>, d, dc, w, 8); run("v_pk_fma_f32", k<2, 16>, d, dc, w, 16);
        run("v_exp_f32", k<1, 8>, d, dc, w, 8);
    }
    return 0;
}
