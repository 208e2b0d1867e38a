// Probe (gfx950): global_load_lds_dwordx4 places lane l's 16 bytes at M0 base + 16 l, also above 64 KB of LDS, also from inline asm
// with M0 written by hand.  hipcc --offload-arch=gfx950 -O3 -o lds_direct lds_direct.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void dma16(const void* g, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_base), "v"(g) : "memory", "m0");
}
__global__ void k(const uint4* __restrict__ src, uint4* __restrict__ dst, int n, int mode) {
    __shared__ __align__(16) unsigned char buf[150 * 1024];
    const int tid = threadIdx.x;
    unsigned char* b = buf + (mode ? 140 * 1024 : 0);
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)b);
    for (int k = 0; k < 4; ++k) {
        const uint4* g = src + ((tid * 7 + k * 64) % n);
        if (mode == 2) dma16(g, base + k * 1024);
        else __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g, (void __attribute__((address_space(3)))*)(b + k * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int k = 0; k < 4; ++k) dst[tid + k * 64] = *reinterpret_cast<uint4*>(b + k * 1024 + tid * 16);
}
int main() {
    const int n = 1000;
    std::vector<uint4> h(n);
    for (int i = 0; i < n; ++i) h[i] = uint4{(unsigned)i, (unsigned)i + 1000u, (unsigned)i + 2000u, (unsigned)i + 3000u};
    uint4 *s, *d;
    if (hipMalloc(&s, n * 16) != hipSuccess || hipMalloc(&d, 256 * 16) != hipSuccess) return 2;
    (void)hipMemcpy(s, h.data(), n * 16, hipMemcpyHostToDevice);
    int rc = 0;
    for (int mode = 0; mode < 3; ++mode) {
        (void)hipMemset(d, 0xff, 256 * 16);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, d, n, mode);
        std::vector<uint4> o(256);
        (void)hipMemcpy(o.data(), d, 256 * 16, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int k = 0; k < 4; ++k) for (int t = 0; t < 64; ++t) { const unsigned e = (t * 7 + k * 64) % n; const uint4 v = o[t + k * 64]; if (v.x != e || v.y != e + 1000 || v.z != e + 2000 || v.w != e + 3000) { if (bad < 4) printf("mode %d k %d lane %d: got %u %u %u %u want %u\n", mode, k, t, v.x, v.y, v.z, v.w, e); ++bad; } }
        printf("lds_direct mode %d (%s): %d mismatches\n", mode, mode == 0 ? "builtin, low LDS" : mode == 1 ? "builtin, LDS offset 140 KB" : "inline asm, LDS offset 140 KB", bad);
        rc |= bad != 0;
    }
    return rc;
}
