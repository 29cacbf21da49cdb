// Plain vs packed fp32 FMA throughput on gfx950, instructions pinned by inline assembly, ONE
// 1024- / 512- / 256-thread workgroup per CU (100 KB of LDS keeps a second one out), 1024
// workgroups.  Reported per instruction: cycles by the waves' own s_memtime (mean over all waves)
// and by wall clock, and the chip's FMA rate by wall clock.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate3 tools/ubench/valu_rate3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP16(X) REP8(X) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, long* cyc, int iters, float a, float b) {
    extern __shared__ float lds[];
    float x[16];
    v2f y[8];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < 8; ++i) y[i] = (v2f){x[2 * i], x[2 * i + 1]};
    v2f av = {a, a}, bv = {b, b};
    lds[threadIdx.x] = a;
    __syncthreads();
    const long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            REP16(X)
#undef X
        } else if (MODE == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(av), "v"(bv));
            REP8(X) REP8(X)
#undef X
        } else if (MODE == 2) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(av));
            REP8(X) REP8(X)
#undef X
        } else {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[i]) : "v"(bv));
            REP8(X) REP8(X)
#undef X
        }
    }
    const long t1 = clock64();
    float s = lds[(threadIdx.x + 1) & 1023];
    for (int i = 0; i < 16; ++i) s += x[i];
    for (int i = 0; i < 8; ++i) s += y[i].x + y[i].y;
    out[(blockIdx.x & 255) * 1024 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads, int fma_per_instr) {
    float* d; long* c;
    const int blocks = 1024, iters = 2000;
    (void)hipMalloc(&d, 256 * 1024 * 4);
    (void)hipMalloc(&c, blocks * 16 * 8);
    auto kf = k<MODE>;
    const int lds = 100 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    kf<<<blocks, threads, lds>>>(d, c, 20, 1.0001f, 0.5f);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    kf<<<blocks, threads, lds>>>(d, c, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long> h(blocks * 16);
    (void)hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
    const int waves = threads / 64;
    double v = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < waves; ++w) v += (double)h[b * 16 + w];
    v /= (double)blocks * waves;
    const double per_wave = v / (iters * 16.0);
    const double instr_total = (double)blocks * waves * iters * 16.0;
    const double tflops = instr_total * 64.0 * fma_per_instr * 2.0 / (ms * 1e-3) / 1e12;
    // wall: 4 rounds of 256 workgroups; instructions per SIMD per round = waves/4 * iters * 16
    const double wall_ns_per_instr_simd = ms * 1e6 / (4.0 * (waves / 4.0) * iters * 16.0);
    printf("%-14s waves/SIMD %d: s_memtime %.2f ticks/instr/wave = %.2f per SIMD | wall %.3f ns/instr/SIMD | %.1f TFLOP/s\n",
           name, waves / 4, per_wave, per_wave / (waves / 4), wall_ns_per_instr_simd, tflops);
    (void)hipFree(d); (void)hipFree(c);
}

int main(int argc, char** argv) {
    // --fma-only: the plain-fp32 issue rate at 1 / 2 / 3 / 4 waves per SIMD (bench.py measures the
    // `at_occupancy` ceilings of its roofline block with it, in the run itself)
    const bool fma_only = argc > 1 && std::string(argv[1]) == "--fma-only";
    for (int th : {256, 512, 768, 1024}) {
        run<0>("v_fma_f32", th, 1);
        if (fma_only) continue;
        run<1>("v_pk_fma_f32", th, 2);
        run<2>("v_pk_mul_f32", th, 2);
        run<3>("v_pk_add_f32", th, 2);
    }
    return 0;
}
