// Radix-16 butterfly + 15 twiddle multiplies on 16 complex registers per lane: the scalar form
// of fft512.h against the packed form (pkmath.h).  Checks that both give the same numbers and
// times them at 2 and 4 waves per SIMD (one workgroup per CU).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast -fno-slp-vectorize -I setk_amd/csrc -I tools/ubench -o pk_dft16 tools/ubench/pk_dft16.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "fft512.h"
#include "pkmath.h"
using namespace setk;

template <int DIR>
SETK_PK void pk_dft4(pk2& a, pk2& b, pk2& c, pk2& d) {
    const pk2 s0 = pk_add(a, c), s1 = pk_sub(a, c), s2 = pk_add(b, d), s3 = pk_sub(b, d);
    a = pk_add(s0, s2);
    c = pk_sub(s0, s2);
    if (DIR < 0) {
        b = pk_add_mi(s1, s3);
        d = pk_add_pi(s1, s3);
    } else {
        b = pk_add_pi(s1, s3);
        d = pk_add_mi(s1, s3);
    }
}
template <int DIR, int M>
SETK_PK pk2 pk_twid16(pk2 v) {
    constexpr float C1 = 0.92387953251128673848f, S1 = 0.38268343236508978178f;
    constexpr float R2 = 0.70710678118654752440f;
    constexpr float D = (float)DIR;
    if constexpr (M == 1) return pk_cmul_s(v, (pk2){C1, D * S1});
    if constexpr (M == 3) return pk_cmul_s(v, (pk2){S1, D * C1});
    if constexpr (M == 9) return pk_cmul_s(v, (pk2){-C1, -D * S1});
    // R2 (1 + D i) v,  D i v,  R2 (-1 + D i) v
    if constexpr (M == 2) return pk_scale_s(DIR < 0 ? pk_add_mi(v, v) : pk_add_pi(v, v), R2);
    if constexpr (M == 4) return DIR < 0 ? pk_add_mi((pk2){0.f, 0.f}, v) : pk_add_pi((pk2){0.f, 0.f}, v);
    if constexpr (M == 6) return pk_scale_s(DIR < 0 ? pk_add_pi(v, v) : pk_add_mi(v, v), -R2);
    return v;
}
template <int DIR>
SETK_PK void pk_dft16(pk2 (&v)[16]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) pk_dft4<DIR>(v[b], v[4 + b], v[8 + b], v[12 + b]);
    v[5] = pk_twid16<DIR, 1>(v[5]);
    v[9] = pk_twid16<DIR, 2>(v[9]);
    v[13] = pk_twid16<DIR, 3>(v[13]);
    v[6] = pk_twid16<DIR, 2>(v[6]);
    v[10] = pk_twid16<DIR, 4>(v[10]);
    v[14] = pk_twid16<DIR, 6>(v[14]);
    v[7] = pk_twid16<DIR, 3>(v[7]);
    v[11] = pk_twid16<DIR, 6>(v[11]);
    v[15] = pk_twid16<DIR, 9>(v[15]);
#pragma unroll
    for (int c = 0; c < 4; ++c) pk_dft4<DIR>(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
}

template <int PK>
__global__ __launch_bounds__(1024) void k(const float2* in, const float2* tw, float2* out, long* cyc, int iters) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = 0.f;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    cf v[16], w[16];
    for (int j = 0; j < 16; ++j) {
        v[j] = in[(tid * 16 + j) % 4096];
        w[j] = tw[(threadIdx.x & 15) * 16 + j];
    }
    const long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (PK) {
            pk2 p[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) p[j] = pk(v[j]);
            pk_dft16<-1>(p);
#pragma unroll
            for (int q = 1; q < 16; ++q) p[dft16_pos(q)] = pk_cmul(p[dft16_pos(q)], pk(w[q]));
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = unpk(p[j]);
        } else {
            dft16<-1>(v);
#pragma unroll
            for (int q = 1; q < 16; ++q) v[dft16_pos(q)] = cmul(v[dft16_pos(q)], w[q]);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = make_float2(v[j].x * 0.25f, v[j].y * 0.25f);  // keep it bounded
    }
    const long t1 = clock64();
    for (int j = 0; j < 16; ++j) out[(size_t)tid * 16 + j] = v[j];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    const int blocks = 512;
    std::vector<float2> hin(4096), htw(256);
    srand(3);
    for (auto& x : hin) x = make_float2(rand() / (float)RAND_MAX - 0.5f, rand() / (float)RAND_MAX - 0.5f);
    for (int i = 0; i < 256; ++i) {
        const double a = -2.0 * M_PI * (i / 16) * (i % 16) / 256.0;
        htw[i] = make_float2((float)cos(a), (float)sin(a));
    }
    float2 *din, *dtw, *o0, *o1; long* dc;
    (void)hipMalloc(&din, 4096 * 8); (void)hipMalloc(&dtw, 256 * 8);
    (void)hipMalloc(&o0, (size_t)blocks * 1024 * 16 * 8); (void)hipMalloc(&o1, (size_t)blocks * 1024 * 16 * 8);
    (void)hipMalloc(&dc, blocks * 16 * 8);
    (void)hipMemcpy(din, hin.data(), 4096 * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dtw, htw.data(), 256 * 8, hipMemcpyHostToDevice);
    const int lds = 100 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    // agreement after 3 rounds
    k<0><<<1, 256, lds>>>(din, dtw, o0, dc, 3);
    k<1><<<1, 256, lds>>>(din, dtw, o1, dc, 3);
    std::vector<float2> a(256 * 16), b(256 * 16);
    (void)hipMemcpy(a.data(), o0, a.size() * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(b.data(), o1, b.size() * 8, hipMemcpyDeviceToHost);
    double num = 0, den = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        num += (a[i].x - b[i].x) * (double)(a[i].x - b[i].x) + (a[i].y - b[i].y) * (double)(a[i].y - b[i].y);
        den += a[i].x * (double)a[i].x + a[i].y * (double)a[i].y;
    }
    printf("packed vs scalar: rel rms %.2e (rms %.3g)\n", sqrt(num / den), sqrt(den / a.size()));
    const int iters = 2000;
    for (int th : {256, 512, 1024})
        for (int which = 0; which < 2; ++which) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            auto launch = [&](int it) {
                if (which) k<1><<<blocks, th, lds>>>(din, dtw, o1, dc, it);
                else k<0><<<blocks, th, lds>>>(din, dtw, o0, dc, it);
            };
            launch(10);
            (void)hipEventRecord(e0); launch(iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            std::vector<long> h(blocks * 16);
            (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
            double v = 0; const int waves = th / 64;
            for (int bb = 0; bb < blocks; ++bb) for (int w = 0; w < waves; ++w) v += (double)h[bb * 16 + w];
            v /= (double)blocks * waves * iters;
            printf("%s waves/SIMD %d: %.0f ticks per (dft16 + 15 twiddles) and wave; kernel %.3f ms\n",
                   which ? "packed" : "scalar", waves / 4, v, ms);
        }
    return 0;
}
