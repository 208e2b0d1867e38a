// Probe: does global_load_lds_dwordx4 place lane l's 16 bytes at M0 base + 16 l (gfx950)?  hipcc --offload-arch=gfx950 -O3 -o lds_direct lds_direct.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const uint4* __restrict__ src, uint4* __restrict__ dst, int n) {
    __shared__ __align__(16) unsigned char buf[64 * 16 * 4];
    const int tid = threadIdx.x;
    for (int k = 0; k < 4; ++k)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + ((tid * 7 + k * 64) % n)), (void __attribute__((address_space(3)))*)(buf + k * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int k = 0; k < 4; ++k) dst[tid + k * 64] = *reinterpret_cast<uint4*>(buf + k * 1024 + tid * 16);
}
int main() {
    const int n = 1000;
    std::vector<uint4> h(n);
    for (int i = 0; i < n; ++i) h[i] = uint4{(unsigned)i, (unsigned)i + 1000u, (unsigned)i + 2000u, (unsigned)i + 3000u};
    uint4 *s, *d;
    hipMalloc(&s, n * 16); hipMalloc(&d, 256 * 16);
    hipMemcpy(s, h.data(), n * 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, d, n);
    std::vector<uint4> o(256);
    hipMemcpy(o.data(), d, 256 * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int k = 0; k < 4; ++k) for (int t = 0; t < 64; ++t) { const unsigned e = (t * 7 + k * 64) % n; const uint4 v = o[t + k * 64]; if (v.x != e || v.y != e + 1000 || v.z != e + 2000 || v.w != e + 3000) { if (bad < 8) printf("k %d lane %d: got %u %u %u %u want %u\n", k, t, v.x, v.y, v.z, v.w, e); ++bad; } }
    printf("lds_direct: %d mismatches\n", bad);
    return bad != 0;
}
