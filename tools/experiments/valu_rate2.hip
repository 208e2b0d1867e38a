// VALU issue-rate probe, part 2: cycles from the shader clock (s_memtime) around the loop, per waves-per-SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, long long* wall, int iters, float seed) {
    float a[8];
    f2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; p[i] = f2{a[i], seed}; }
    __syncthreads();
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
            if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
            if (OP == 3) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a[i]));
            if (OP == 4) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[i]));
            if (OP == 5) asm volatile("v_fma_mix_f32 %0, %0, %0, %0" : "+v"(a[i]));
            if (OP == 6) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
            if (OP == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7]));
            if (OP == 8) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a[i]));
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = c1 - c0; wall[blockIdx.x] = w1 - w0; }
}
template <typename K>
static void run(const char* name, K kern, float* d, long long* dc, long long* dw, int wgs_per_cu) {
    const int blocks = 256 * wgs_per_cu, iters = 4096;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, dc, dw, 16, 1.0f);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, dc, dw, iters, 1.0f);
    hipDeviceSynchronize();
    long long c[4], w[4];
    hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
    hipMemcpy(w, dw, sizeof(w), hipMemcpyDeviceToHost);
    const double instr = (double)wgs_per_cu * iters * 8;  // per SIMD (one wave of each workgroup per SIMD)
    printf("%-28s wg/CU %d: s_memtime %lld  wall(100MHz) %lld  -> %.2f memtime-ticks / instr / SIMD, %.2f ns / instr / SIMD\n", name, wgs_per_cu, c[0], w[0], c[0] / instr,
           w[0] * 10.0 / instr);
}
int main() {
    float* d; long long *dc, *dw;
    hipMalloc(&d, 256 * 8 * 256 * 4); hipMalloc(&dc, 256 * 8 * 8); hipMalloc(&dw, 256 * 8 * 8);
    for (int w : {1, 2, 4}) {
        run("v_fma_f32 (same reg)", k<0>, d, dc, dw, w); run("v_fma_f32 (3 regs)", k<6>, d, dc, dw, w); run("v_mul_f32", k<3>, d, dc, dw, w); run("v_add_f32", k<8>, d, dc, dw, w);
        run("v_exp_f32", k<1>, d, dc, dw, w);
        run("v_pk_fma_f32 (same reg)", k<2>, d, dc, dw, w); run("v_pk_fma_f32 (3 regs)", k<7>, d, dc, dw, w); run("v_pk_mul_f32", k<4>, d, dc, dw, w); run("v_fma_mix_f32", k<5>, d, dc, dw, w);
    }
    return 0;
}
