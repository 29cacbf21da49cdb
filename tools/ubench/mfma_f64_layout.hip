// Operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950, found by experiment:
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_f64_layout.hip -o tools/ubench/mfma_f64_layout && tools/ubench/mfma_f64_layout
// Every lane supplies one A and one B value (random); the host forms D = A B under the
// hypothesis A[i][k] = a[lane i + 16 k], B[k][j] = b[lane j + 16 k] and looks every result
// register up in D.  Also times a dependent-free stream of MFMAs (cycles per instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void probe(const double* a, const double* b, double* d) {
    const int l = threadIdx.x;
    v4d c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) d[l * 4 + v] = c[v];
}

__global__ void rate(double* out, long long* cyc, int iters) {
    const int l = threadIdx.x;
    v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = 1.0 + l, b = 0.5 * l;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    long long t1 = clock64();
    out[blockIdx.x * 64 + l] = c0[0] + c1[1] + c2[2] + c3[3];
    if (l == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    double ha[64], hb[64], hd[256], *da, *db, *dd;
    srand(1);
    for (int i = 0; i < 64; ++i) { ha[i] = rand() % 97 + 1; hb[i] = rand() % 89 + 1; }
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dd, sizeof hd);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice);
    hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(da, db, dd);
    hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
    double D[16][16];
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            D[i][j] = 0;
            for (int k = 0; k < 4; ++k) D[i][j] += ha[i + 16 * k] * hb[j + 16 * k];
        }
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int v = 0; v < 4; ++v) {
            int fi = -1, fj = -1, n = 0;
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j)
                    if (D[i][j] == hd[l * 4 + v]) { fi = i; fj = j; ++n; }
            if (n != 1) ++bad;
            if (l < 20 || l % 16 == 0) printf("lane %2d reg %d -> D[%2d][%2d] (%d match)\n", l, v, fi, fj, n);
        }
    printf("unmatched registers: %d\n", bad);
    long long* dc; double* dout; long long hc;
    hipMalloc(&dc, 8); hipMalloc(&dout, 256 * 4 * 64 * 8);
    const int iters = 10000;
    rate<<<1, 64>>>(dout, dc, iters);
    hipMemcpy(&hc, dc, 8, hipMemcpyDeviceToHost);
    printf("1 wave: %.1f clock64 ticks per MFMA\n", (double)hc / (4.0 * iters));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rate<<<1024, 256>>>(dout, dc, iters);
    hipEventRecord(e0);
    rate<<<1024, 256>>>(dout, dc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("full chip: %.1f TFLOP/s fp64 (1024 x 4 waves x %d MFMAs of 2048 flop in %.3f ms)\n",
           1024.0 * 4 * 4 * iters * 2048 / (ms * 1e-3) / 1e12, 4 * iters, ms);
    return 0;
}
