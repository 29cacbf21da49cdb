// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction for
// v_fma_f32 / v_add_f32 / v_pk_fma_f32 at 1, 2, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, int iters, float a, float b) {
    float x[16];
    v2f y[8];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < 8; ++i) y[i] = (v2f){x[2 * i], x[2 * i + 1]};
    v2f av = {a, a}, bv = {b, b};
    long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], a, b);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = x[i] + a;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) y[i] = __builtin_elementwise_fma(y[i], av, bv);
        }
    }
    long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    for (int i = 0; i < 8; ++i) s += y[i].x + y[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}

template <int MODE>
void run(const char* name, int threads, int n_instr_per_iter) {
    float* d;
    hipMalloc(&d, 1 << 22);
    int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, threads>>>(d, 100, 1.0001f, 0.5f);
    hipEventRecord(e0);
    k<MODE><<<256, threads>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float clk; hipMemcpy(&clk, d, 4, hipMemcpyDeviceToHost);
    double waves_per_simd = threads / 64.0 / 4.0;
    double instr_per_simd = (double)iters * n_instr_per_iter * waves_per_simd;
    printf("%-14s waves/SIMD=%.0f  shader-clk cycles/instr(per SIMD)=%.2f  wall-derived@2.4GHz=%.2f\n",
           name, waves_per_simd, clk / instr_per_simd, ms * 1e-3 * 2.4e9 / instr_per_simd);
    hipFree(d);
}

int main() {
    for (int th : {256, 512, 1024}) {
        run<0>("v_fma_f32", th, 16);
        run<1>("v_add_f32", th, 16);
        run<2>("v_pk_fma_f32", th, 8);
    }
    return 0;
}
