// Experiment: is the k-accumulation of the f32 / f64 MFMA on gfx950 a left-to-right chain of fused
// multiply-adds, i.e. bit-identical to  s = fma(a[k], b[k], s)  for k = 0, 1, ...  starting from C?
// If so, the row pass of the separable Gaussian (an FMA chain, left to right) can run on the matrix cores as a
// banded-Toeplitz product without breaking the bit-exact contract.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_order mfma_order.hip && ./mfma_order
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// D(32x32) = sum over KT steps of A(32x2) * B(2x32), v_mfma_f32_32x32x2f32.  A: lane l -> row l%32, k = l/32.
// B: lane l -> col l%32, k = l/32.  D: lane l -> col l%32, rows 8*(i/4) + 4*(l/32)... (taken from the ISA layout:
// element i of the 16-vector is row (i % 4) + 8 * (i / 4) + 4 * (l / 32)).
__global__ void k_f32(const float* A /*[32][K]*/, const float* B /*[K][32]*/, float* D /*[32][32]*/, int K) {
    const int l = threadIdx.x;
    f32x16 acc = {0};
    for (int k0 = 0; k0 < K; k0 += 2) {
        const float a = A[(l % 32) * K + k0 + l / 32];
        const float b = B[(k0 + l / 32) * 32 + l % 32];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) {
        const int row = (i % 4) + 8 * (i / 4) + 4 * (l / 32);
        D[row * 32 + l % 32] = acc[i];
    }
}

// v_mfma_f64_16x16x4f64: A: lane l -> row l%16, k = l/16; B: lane l -> col l%16, k = l/16; D: 4 values per lane:
// col l%16, row 4*(l/16) + i.
__global__ void k_f64(const double* A /*[16][K]*/, const double* B /*[K][16]*/, double* D /*[16][16]*/, int K) {
    const int l = threadIdx.x;
    f64x4 acc = {0};
    for (int k0 = 0; k0 < K; k0 += 4) {
        const double a = A[(l % 16) * K + k0 + l / 16];
        const double b = B[(k0 + l / 16) * 16 + l % 16];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) D[(4 * (l / 16) + i) * 16 + l % 16] = acc[i];
}

template <typename T> static T rnd() { return (T)((double)rand() / RAND_MAX * 2.0 - 1.0) * (T)(1 + rand() % 1000) / (T)(1 + rand() % 1000); }

int main() {
    const int K = 32;
    int bad32 = 0, bad64 = 0, badsum32 = 0;
    for (int trial = 0; trial < 50; ++trial) {
        std::vector<float> A(32 * K), B(K * 32), D(32 * 32);
        for (auto& v : A) v = rnd<float>();
        for (auto& v : B) v = rnd<float>();
        float *dA, *dB, *dD;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_f32, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                float s = 0.f, t = 0.f;
                for (int k = 0; k < K; ++k) { s = fmaf(A[i * K + k], B[k * 32 + j], s); t = t + A[i * K + k] * B[k * 32 + j]; }
                if (s != D[i * 32 + j]) ++bad32;
                if (t != D[i * 32 + j]) ++badsum32;
            }
        hipFree(dA); hipFree(dB); hipFree(dD);
        std::vector<double> A6(16 * K), B6(K * 16), D6(16 * 16);
        for (auto& v : A6) v = rnd<double>();
        for (auto& v : B6) v = rnd<double>();
        double *eA, *eB, *eD;
        hipMalloc(&eA, A6.size() * 8); hipMalloc(&eB, B6.size() * 8); hipMalloc(&eD, D6.size() * 8);
        hipMemcpy(eA, A6.data(), A6.size() * 8, hipMemcpyHostToDevice); hipMemcpy(eB, B6.data(), B6.size() * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_f64, dim3(1), dim3(64), 0, 0, eA, eB, eD, K);
        hipMemcpy(D6.data(), eD, D6.size() * 8, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double s = 0.0;
                for (int k = 0; k < K; ++k) s = fma(A6[i * K + k], B6[k * 16 + j], s);
                if (s != D6[i * 16 + j]) ++bad64;
            }
        hipFree(eA); hipFree(eB); hipFree(eD);
    }
    printf("f32 32x32x2: %d of %d elements differ from the sequential fmaf chain (%d from the unfused mul+add chain)\n", bad32, 50 * 1024, badsum32);
    printf("f64 16x16x4: %d of %d elements differ from the sequential fma chain\n", bad64, 50 * 256);
    return 0;
}
