// Does the fp32 matrix pipe run beside the VALU?  One 1024-thread workgroup per CU
// (4 waves per SIMD, as pass 1): waves 0-7 run VALU chains, waves 8-15 run MFMAs
// (either role can be switched off).  Each wave times its own loop with the shader
// clock; printed: cycles per instruction and wave, averaged per role.
//   mfma kinds: 4x4x1 (16 blocks, 2 passes), 16x16x4 (8 passes), 32x32x2 (16 passes)
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate tools/ubench/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, long* cyc, int iters, int valu_on, int mfma_on,
                                          float a, float b) {
    const int wave = threadIdx.x >> 6;
    long t0 = 0, t1 = 0;
    float s = 0.f;
    if (wave < 8) {
        if (valu_on) {
            float x[16];
            for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
            t0 = clock64();
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], a, b);
            t1 = clock64();
            for (int i = 0; i < 16; ++i) s += x[i];
        }
    } else if (mfma_on) {
        if (KIND == 0) {
            v4f acc[12];
            for (int i = 0; i < 12; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
            t0 = clock64();
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int i = 0; i < 12; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
            t1 = clock64();
            for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][3];
        } else if (KIND == 1) {
            v4f acc[12];
            for (int i = 0; i < 12; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
            t0 = clock64();
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int i = 0; i < 12; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            t1 = clock64();
            for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][3];
        } else {
            v16f acc[4];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
            t0 = clock64();
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            t1 = clock64();
            for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int KIND>
void run(const char* name, int valu_on, int mfma_on) {
    float* d; long* c;
    const int blocks = 256, iters = 4000;
    hipMalloc(&d, blocks * 1024 * 4);
    hipMalloc(&c, blocks * 16 * 8);
    k<KIND><<<blocks, 1024>>>(d, c, 50, valu_on, mfma_on, 1.0001f, 0.5f);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<KIND><<<blocks, 1024>>>(d, c, iters, valu_on, mfma_on, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long> h(blocks * 16);
    hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
    double v = 0, m = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < 16; ++w) (w < 8 ? v : m) += (double)h[b * 16 + w];
    v /= blocks * 8; m /= blocks * 8;
    const int n_mfma = 12;
    printf("%-8s valu %d mfma %d: kernel %.3f ms | VALU wave: %.2f cyc/instr | MFMA wave: %.2f cyc/instr\n",
           name, valu_on, mfma_on, ms, v / (iters * 16.0), m / (iters * (double)n_mfma));
    hipFree(d); hipFree(c);
}

int main() {
    run<0>("4x4x1", 1, 0);
    run<0>("4x4x1", 0, 1);
    run<0>("4x4x1", 1, 1);
    run<1>("16x16x4", 0, 1);
    run<1>("16x16x4", 1, 1);
    run<2>("32x32x2", 0, 1);
    run<2>("32x32x2", 1, 1);
    printf("# 2 waves of each role per SIMD.  Alone, 2 VALU waves per SIMD issue one instruction per\n"
           "# ~4.75 ticks each; an MFMA wave's figure / 2 = pipe occupancy per instruction per SIMD.\n");
    return 0;
}
