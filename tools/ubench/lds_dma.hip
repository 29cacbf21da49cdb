// Probe of the gfx950 direct global->LDS load: which LDS bytes does lane i write?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_dma.hip -o /tmp/lds_dma && /tmp/lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define GL __attribute__((address_space(1)))
#define LD __attribute__((address_space(3)))
template <int SZ>
__global__ void probe(const float* src, float* dst, int misalign) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sm = (float*)smem;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) sm[i] = -1.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) {
        const GL float* g = (const GL float*)src + misalign + lane * (SZ / 4);
        LD char* l = (LD char*)smem;
        if constexpr (SZ == 4) {
            __builtin_amdgcn_global_load_lds(g, l, 4, 0, 0);
            __builtin_amdgcn_global_load_lds(g + 64, l + 256, 4, 0, 0);
        } else {
            __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
            __builtin_amdgcn_global_load_lds(g + 256, l + 1024, 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) dst[i] = sm[i];
}
int main() {
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    float *d, *o;
    (void)hipMalloc(&d, 4096 * 4);
    (void)hipMalloc(&o, 1024 * 4);
    (void)hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    std::vector<float> r(1024);
    for (int mis = 0; mis < 2; ++mis) {
        for (int sz : {4, 16}) {
            if (sz == 4) hipLaunchKernelGGL(probe<4>, dim3(1), dim3(256), 4096, 0, d, o, mis);
            else hipLaunchKernelGGL(probe<16>, dim3(1), dim3(256), 4096, 0, d, o, mis);
            hipError_t e = hipDeviceSynchronize();
            (void)hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost);
            int bad = 0, n = 2 * 64 * sz / 4;
            for (int i = 0; i < n; ++i) bad += (r[i] != (float)(i + mis));
            printf("size %2d misalign %d: err=%d linear-copy mismatches %d of %d; first: %g %g %g %g %g %g\n", sz, mis,
                   (int)e, bad, n, r[0], r[1], r[2], r[3], r[4], r[5]);
        }
    }
    return 0;
}
