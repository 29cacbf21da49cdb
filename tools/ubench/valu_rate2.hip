// VALU issue rate on gfx950 with the instructions pinned by inline assembly (the C form
// of tools/ubench/valu_rate.hip is SLP-vectorised into v_pk_* by the compiler, so its
// "per instruction" figures are per HALF packed instruction).  Every wave times its own
// loop with s_memtime; printed: cycles per instruction and wave, and per SIMD.
//   indep: 16 independent accumulators per wave    dep: one dependent chain
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate2 tools/ubench/valu_rate2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, long* cyc, int iters, float a, float b) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    extern __shared__ float lds[];
    lds[threadIdx.x] = a;
    __syncthreads();
    const long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            REP16(X)
#undef X
        } else if (MODE == 1) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(a), "v"(b));
            REP16(X)
#undef X
        } else if (MODE == 2) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            REP16(X)
#undef X
        } else if (MODE == 3) {  // DPP move + fma pairs (the Hermitian split's partner fetch)
#define X(i) asm volatile("v_mov_b32_dpp %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
            REP16(X)
#undef X
        } else if (MODE == 4) {  // dependent LDS round trip: write, read back, 16 times
#define X(i)                                                                              \
    asm volatile("ds_write_b32 %1, %0\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)"       \
                 : "+v"(x[0]) : "v"(threadIdx.x * 4) : "memory");
            REP16(X)
#undef X
        } else {  // 16 independent LDS reads, one wait
#define X(i) asm volatile("ds_read_b64 %0, %1 offset:" #i "*8" : "=v"(*(double*)&x[(i & 7) * 2]) : "v"((threadIdx.x & 63) * 128) : "memory");
            REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads) {
    float* d; long* c;
    const int blocks = 256, iters = 4000;
    (void)hipMalloc(&d, blocks * 1024 * 4);
    (void)hipMalloc(&c, blocks * 16 * 8);
    k<MODE><<<blocks, threads, 32768>>>(d, c, 50, 1.0001f, 0.5f);
    k<MODE><<<blocks, threads, 32768>>>(d, c, iters, 1.0001f, 0.5f);
    (void)hipDeviceSynchronize();
    std::vector<long> h(blocks * 16);
    (void)hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
    const int waves = threads / 64;
    double v = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < waves; ++w) v += (double)h[b * 16 + w];
    v /= (double)blocks * waves;
    const double per_wave = v / (iters * 16.0);
    printf("%-22s waves/SIMD %d: %.2f cycles per instruction and wave, %.2f per SIMD\n", name,
           waves / 4, per_wave, per_wave / (waves / 4));
    (void)hipFree(d); (void)hipFree(c);
}

int main() {
    for (int th : {256, 512, 768, 1024}) {
        run<0>("v_fma_f32 indep", th);
        run<1>("v_fma_f32 dep", th);
        run<2>("v_add_f32 indep", th);
        run<3>("v_mov_b32_dpp indep", th);
        run<4>("ds_write+read dep", th);
        run<5>("ds_read_b64 x16 + wait", th);
    }
    return 0;
}
