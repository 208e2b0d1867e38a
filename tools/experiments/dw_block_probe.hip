// Block-level probe (round 3): what one "16 channels x 16 pixels" unit of the FeedForward kernel's second phase costs per SIMD
//   V0  as built in round 2: 9 taps x (2 ds_read_b128 + 8 v_fma_mix_f32) per 8 channels x 32 px ... here normalised to the same
//       256 outputs: 18 ds_read_b128 + 36 v_fma_mix_f32, then the packed-float32 GELU of 4 values per lane (2 pairs)
//   V1  depthwise conv as 5 chained v_mfma_f32_16x16x32_f16 (accumulator in VGPRs), 10 ds_read_b128, GELU in plain float32
//   V2  V1 with the accumulator in AGPRs (+ 4 v_accvgpr_read)
//   V3  V1 with the packed-float32 GELU
//   V4  V1 without the GELU (MFMA + LDS only), V5 GELU plain only, V6 GELU packed only
// 4 waves per SIMD (256-thread workgroups, 36 KB of LDS each), whole-kernel HIP-event time / blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f2 gelu_pk(f2 x) {
    const f2 xc = f2{__builtin_amdgcn_fmed3f(x.x, -4.f, 4.f), __builtin_amdgcn_fmed3f(x.y, -4.f, 4.f)};
    const f2 t = xc * xc;
    f2 r = __builtin_elementwise_fma(t, f2{-1e-9f, -1e-9f}, f2{1e-7f, 1e-7f});
    r = __builtin_elementwise_fma(r, t, f2{-1e-5f, -1e-5f});
    r = __builtin_elementwise_fma(r, t, f2{1e-4f, 1e-4f});
    r = __builtin_elementwise_fma(r, t, f2{-1e-3f, -1e-3f});
    r = __builtin_elementwise_fma(r, t, f2{1e-2f, 1e-2f});
    r = __builtin_elementwise_fma(r, t, f2{-6e-2f, -6e-2f});
    r = __builtin_elementwise_fma(r, t, f2{0.39f, 0.39f});
    return x * __builtin_elementwise_fma(xc, r, f2{0.5f, 0.5f});
}
__device__ __forceinline__ float gelu_1(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -4.f, 4.f), t = xc * xc;
    float r = __builtin_fmaf(t, -1e-9f, 1e-7f);
    r = __builtin_fmaf(r, t, -1e-5f);
    r = __builtin_fmaf(r, t, 1e-4f);
    r = __builtin_fmaf(r, t, -1e-3f);
    r = __builtin_fmaf(r, t, 1e-2f);
    r = __builtin_fmaf(r, t, -6e-2f);
    r = __builtin_fmaf(r, t, 0.39f);
    return x * __builtin_fmaf(xc, r, 0.5f);
}
__device__ __forceinline__ void fma_mix_lo(float& acc, unsigned a, unsigned b) { asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(a), "v"(b)); }
__device__ __forceinline__ void fma_mix_hi(float& acc, unsigned a, unsigned b) { asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(a), "v"(b)); }

template <int V>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    extern __shared__ __align__(16) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 32 * 1024 / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x2c002c00u + (i & 0xff);
    __syncthreads();
    const unsigned char* base = lds + (lane & 15) * 144 + ((lane >> 4) & 1) * 2816 + (lane >> 5) * 16;  // q pairs (0,1), (2,3) share a lane group: same banks only when their chunks differ by whole rows
    float sum = 0.f;
    h8 areg[5];
#pragma unroll
    for (int mm = 0; mm < 5; ++mm) areg[mm] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(lds + 20 * 1024 + mm * 1024 + lane * 16));
    for (int it = 0; it < iters; ++it) {
        const unsigned char* b = base + (it & 7) * 2816;
        f4 acc = f4{0, 0, 0, 0};
        if (V == 0) {
            float a8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a8[j] = 0.f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {  // 8 outputs per lane: two of these "half blocks" make 256 x 2; the loop below runs it once per 2 units -> normalise in the report
                const uint4 hv = *reinterpret_cast<const uint4*>(b + (tap / 3) * 2816 + (tap % 3) * 144);
                const uint4 wv = *reinterpret_cast<const uint4*>(lds + 30 * 1024 + tap * 64 + (lane >> 5) * 16);
                fma_mix_lo(a8[0], hv.x, wv.x); fma_mix_hi(a8[1], hv.x, wv.x); fma_mix_lo(a8[2], hv.y, wv.y); fma_mix_hi(a8[3], hv.y, wv.y);
                fma_mix_lo(a8[4], hv.z, wv.z); fma_mix_hi(a8[5], hv.z, wv.z); fma_mix_lo(a8[6], hv.w, wv.w); fma_mix_hi(a8[7], hv.w, wv.w);
            }
#pragma unroll
            for (int j = 0; j < 8; j += 2) { const f2 g = gelu_pk(f2{a8[j], a8[j + 1]}); sum += g.x + g.y; }
        } else {
            if (V != 5 && V != 6) {
                if (V == 2) {
                    f4 accA = f4{0, 0, 0, 0};
#pragma unroll
                    for (int mm = 0; mm < 5; ++mm) {
                        const h8 bv = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(b + mm * 32));
                        const h8 av = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(lds + 20 * 1024 + mm * 1024 + lane * 16));
                        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(accA) : "v"(av), "v"(bv));
                    }
                    asm volatile("s_nop 7\n s_nop 7\n v_accvgpr_read_b32 %0, %4\n v_accvgpr_read_b32 %1, %5\n v_accvgpr_read_b32 %2, %6\n v_accvgpr_read_b32 %3, %7"
                                 : "=v"(acc[0]), "=v"(acc[1]), "=v"(acc[2]), "=v"(acc[3]) : "a"(accA[0]), "a"(accA[1]), "a"(accA[2]), "a"(accA[3]));
                } else if (V >= 7) {
#pragma unroll
                    for (int mm = 0; mm < 5; ++mm) {
                        const h8 bv = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(b + mm * 32));
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(areg[mm], bv, acc, 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int mm = 0; mm < 5; ++mm) {
                        const h8 bv = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(b + mm * 32));
                        const h8 av = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(lds + 20 * 1024 + mm * 1024 + lane * 16));
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc, 0, 0, 0);
                    }
                }
            } else {
                acc = f4{(float)it, sum, 1.f, 2.f};
            }
            if (V == 1 || V == 2 || V == 5 || V == 7) sum += gelu_1(acc[0]) + gelu_1(acc[1]) + gelu_1(acc[2]) + gelu_1(acc[3]);
            if (V == 3 || V == 6 || V == 8) { const f2 g0 = gelu_pk(f2{acc[0], acc[1]}), g1 = gelu_pk(f2{acc[2], acc[3]}); sum += g0.x + g0.y + g1.x + g1.y; }
            if (V == 4 || V == 9) sum += acc[0] + acc[1] + acc[2] + acc[3];
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}
template <typename K>
static double run(K kern, float* d, double units_per_iter) {
    const int w = 4, blocks = 256 * w, iters = 20000, lds = 36 * 1024;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d, 200);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / ((double)w * iters * units_per_iter);  // ns per 256-output unit per SIMD
}
int main() {
    float* d;
    (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    printf("ns per unit (16 channels x 16 pixels: depthwise 3x3 + GELU) per SIMD, 4 waves per SIMD\n");
    printf("V0 fma_mix + packed GELU (round 2) : %6.1f\n", run(k<0>, d, 2.0));
    printf("V1 5 MFMA (VGPR acc) + plain GELU   : %6.1f\n", run(k<1>, d, 1.0));
    printf("V2 5 MFMA (AGPR acc) + plain GELU   : %6.1f\n", run(k<2>, d, 1.0));
    printf("V3 5 MFMA (VGPR acc) + packed GELU  : %6.1f\n", run(k<3>, d, 1.0));
    printf("V4 5 MFMA + LDS reads only          : %6.1f\n", run(k<4>, d, 1.0));
    printf("V5 plain GELU only (4 values)       : %6.1f\n", run(k<5>, d, 1.0));
    printf("V6 packed GELU only (2 pairs)       : %6.1f\n", run(k<6>, d, 1.0));
    printf("V7 5 MFMA (A in registers) + plain GELU : %6.1f\n", run(k<7>, d, 1.0));
    printf("V8 5 MFMA (A in registers) + packed GELU: %6.1f\n", run(k<8>, d, 1.0));
    printf("V9 5 MFMA (A in registers), 5 LDS reads : %6.1f\n", run(k<9>, d, 1.0));
    return 0;
}
