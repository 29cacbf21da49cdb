// Operand / result layout of v_mfma_f32_4x4x1_16b_f32 on gfx950, found by probing:
// A = indicator of lane la, B = indicator of lane lb, where does the 1 land in D?
// Hypothesis checked: block = lane / 4, A row i = la % 4, B column j = lb % 4,
// D[i][j] of block k in register i of lane 4 k + j.
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma_layout tools/ubench/mfma_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void probe(int* out) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            v4f d = {0.f, 0.f, 0.f, 0.f};
            d = __builtin_amdgcn_mfma_f32_4x4x1f32(lane == la ? 1.f : 0.f, lane == lb ? 1.f : 0.f, d, 0, 0, 0);
            for (int r = 0; r < 4; ++r)
                if (d[r] != 0.f) out[la * 64 + lb] = r * 64 + lane;
        }
}

int main() {
    int* d;
    hipMalloc(&d, 4096 * 4);
    hipMemset(d, 0xff, 4096 * 4);
    probe<<<1, 64>>>(d);
    std::vector<int> h(4096);
    hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const int want = (la / 4 == lb / 4) ? (la % 4) * 64 + 4 * (lb / 4) + lb % 4 : -1;
            if (h[la * 64 + lb] != want) {
                if (bad < 12) printf("la %d lb %d: got reg %d lane %d, expected %d\n", la, lb,
                                     h[la * 64 + lb] / 64, h[la * 64 + lb] % 64, want);
                ++bad;
            }
        }
    printf("mfma_f32_4x4x1 layout hypothesis: %d mismatches of 4096\n", bad);
    return 0;
}
