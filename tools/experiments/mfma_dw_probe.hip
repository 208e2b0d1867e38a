// Probe for moving the depthwise 3x3 convs of MST++ from the vector unit to the matrix pipe (round 3):
//   * issue rate of v_mfma_f32_16x16x32_f16, v_mfma_f32_32x32x16_f16 and v_mfma_f32_4x4x4_16b_f16, w waves per SIMD;
//   * what a stream of NV packed FMAs per MFMA costs beside them (does the vector unit issue under a running MFMA?);
//   * whether v_pk_fma_f32 honours the clamp bit (a one-instruction scale + shift + clamp to [0, 1]).
// Whole-kernel HIP-event timing, workgroups pinned to w waves per SIMD by their LDS size (as valu_rate4.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// KIND 0: 16x16x32, 1: 32x32x16, 2: 4x4x4 (16 blocks), 3: no MFMA (vector stream only).  NV packed FMAs follow every MFMA.
template <int KIND, int NV, int VOP = 0>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    extern __shared__ float lds[];
    constexpr int NACC = 4;
    f4 acc4[NACC];
    f16v acc16[2];
    f2 p[8];
    h8 a, b;
    h4 a4, b4;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed * 0.001f + threadIdx.x * 1e-4f); b[i] = (_Float16)(0.5f + i * 0.01f); p[i] = f2{seed + i, seed}; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { a4[i] = a[i]; b4[i] = b[i]; }
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc4[i] = f4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc16[i][j] = 0.f;
    if (seed < 0) lds[threadIdx.x] = seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NACC; ++m) {
            if (KIND == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4[m]) : "v"(a), "v"(b));
            if (KIND == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc16[m & 1]) : "v"(a), "v"(b));
            if (KIND == 2) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %0" : "+v"(acc4[m]) : "v"(a4), "v"(b4));
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (VOP == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[v % 8]) : "v"(p[(v + 1) % 8]), "v"(p[(v + 2) % 8]));
                if (VOP == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p[v % 8].x) : "v"(p[(v + 1) % 8].x), "v"(p[(v + 2) % 8].x));
                if (VOP == 2) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(p[v % 8].x) : "v"(p[(v + 1) % 8].x), "v"(p[(v + 2) % 8].x));
                if (VOP == 3) asm volatile("v_fma_mix_f32 %0, %1, %2, %0" : "+v"(p[v % 8].x) : "v"(p[(v + 1) % 8].x), "v"(p[(v + 2) % 8].x));
                if (VOP == 4) asm volatile("v_pk_mul_f32 %0, %1, %2" : "+v"(p[v % 8]) : "v"(p[(v + 1) % 8]), "v"(p[(v + 2) % 8]));
                if (VOP == 5) asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(p[v % 8].x) : "v"(p[(v + 1) % 8].x), "v"(p[(v + 2) % 8].x));
                if (VOP == 6) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "+v"(p[v % 8].x) : "v"(p[(v + 1) % 8].x), "v"(p[(v + 2) % 8].x));
                if (VOP == 7) asm volatile("v_pk_add_f32 %0, %1, %2" : "+v"(p[v % 8]) : "v"(p[(v + 1) % 8]), "v"(p[(v + 2) % 8]));
                if (VOP == 8) asm volatile("v_mul_f32 %0, %1, %2" : "+v"(p[v % 8].x) : "v"(p[(v + 1) % 8].x), "v"(p[(v + 2) % 8].x));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc16[i][j];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K>
static double run(K kern, float* d, int w) {
    const int blocks = 256 * w, iters = 4096;
    const int lds = (w == 8 ? 18 : w == 4 ? 36 : w == 2 ? 72 : 150) * 1024;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d, 64, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / ((double)w * iters * 4);  // ns per (MFMA + NV packed FMAs) group per SIMD
}

__global__ void k_clamp(const float* x, float* y, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    f2 v = f2{x[i], -x[i]}, s = f2{0.125f, 0.125f}, o = f2{0.5f, 0.5f}, r;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(v), "v"(s), "v"(o));
    y[2 * i] = r.x; y[2 * i + 1] = r.y;
}

#define ROW(name, KIND)                                                                                                                   \
    for (int w : {1, 2, 4}) {                                                                                                             \
        printf("%-22s w%d:", name, w);                                                                                                    \
        printf("  nv0 %6.2f  nv2 %6.2f  nv4 %6.2f  nv6 %6.2f  nv8 %6.2f  nv12 %6.2f  nv16 %6.2f\n", run(k<KIND, 0>, d, w), run(k<KIND, 2>, d, w), \
               run(k<KIND, 4>, d, w), run(k<KIND, 6>, d, w), run(k<KIND, 8>, d, w), run(k<KIND, 12>, d, w), run(k<KIND, 16>, d, w));   \
    }

int main() {
    float* d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    printf("ns per group (one MFMA followed by NV v_pk_fma_f32) per SIMD, w waves per SIMD\n");
    ROW("mfma 16x16x32 f16", 0)
    ROW("mfma 32x32x16 f16", 1)
    ROW("mfma 4x4x4 16b f16", 2)
    ROW("no mfma (vector only)", 3)
#define ROW2(name, KIND, VOP)                                                                                                                 \
    for (int w : {2, 4}) {                                                                                                                    \
        printf("%-34s w%d:", name, w);                                                                                                        \
        printf("  nv2 %6.2f  nv4 %6.2f  nv8 %6.2f  nv16 %6.2f  nv24 %6.2f\n", run(k<KIND, 2, VOP>, d, w), run(k<KIND, 4, VOP>, d, w),         \
               run(k<KIND, 8, VOP>, d, w), run(k<KIND, 16, VOP>, d, w), run(k<KIND, 24, VOP>, d, w));                                         \
    }
    printf("other vector instructions beside v_mfma_f32_16x16x32_f16 (KIND 0) and alone (KIND 3)\n");
    ROW2("16x16x32 + v_fma_f32", 0, 1) ROW2("alone  v_fma_f32", 3, 1)
    ROW2("16x16x32 + v_pk_fma_f16", 0, 2) ROW2("alone  v_pk_fma_f16", 3, 2)
    ROW2("16x16x32 + v_fma_mix_f32", 0, 3) ROW2("alone  v_fma_mix_f32", 3, 3)
    ROW2("16x16x32 + v_pk_mul_f32", 0, 4) ROW2("alone  v_pk_mul_f32", 3, 4)
    ROW2("16x16x32 + v_med3_f32", 0, 5) ROW2("alone  v_med3_f32", 3, 5)
    ROW2("16x16x32 + v_cvt_pk_f16_f32", 0, 6) ROW2("alone  v_cvt_pk_f16_f32", 3, 6)
    ROW2("16x16x32 + v_pk_add_f32", 0, 7) ROW2("alone  v_pk_add_f32", 3, 7)
    ROW2("16x16x32 + v_mul_f32", 0, 8) ROW2("alone  v_mul_f32", 3, 8)
    ROW2("32x32x16 + v_fma_f32", 1, 1) ROW2("32x32x16 + v_pk_fma_f16", 1, 2)
    float hx[8] = {-10.f, -4.f, -1.f, 0.f, 1.f, 3.9f, 4.f, 10.f}, hy[16], *dx, *dy;
    hipMalloc(&dx, sizeof(hx)); hipMalloc(&dy, sizeof(hy));
    hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_clamp, dim3(1), dim3(64), 0, 0, dx, dy, 8);
    hipMemcpy(hy, dy, sizeof(hy), hipMemcpyDeviceToHost);
    printf("v_pk_fma_f32 clamp: x/8 + 0.5 ->");
    for (int i = 0; i < 8; ++i) printf("  x=%g: %g | %g", hx[i], hy[2 * i], hy[2 * i + 1]);
    printf("\n");
    return 0;
}
