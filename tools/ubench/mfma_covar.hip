// MFMA vs VALU for the masked outer-product accumulation of pass 1 (VERDICT r1
// weak-7 / next-4c): Phi_s, Phi_n (8 x 8 complex each) of every bin from an
// LDS-resident tile of 4 frames x 8 channels x 256 bins, the hand-over format of
// stft_covar_kernel.  Both variants sweep the same tile `reps` times, so what is
// timed is the contraction and its operand traffic, nothing else.
//
//   valu   thread (f, h): the product's kernel -- x_i conj(x_j) once, two FMAs
//          per mask, Hermitian triangle only, 64 accumulators / thread, 512 threads
//   mfma   one v_mfma_f32_16x16x4_f32 per (bin, frame pair):
//             A (16 x 4) rows 0-7  = [xr_i(t) xi_i(t) | xr_i(t+1) xi_i(t+1)]
//                        rows 8-15 = [xi_i(t) -xr_i(t) | ...]
//             B (4 x 16) cols 0-7 = m_s [xr_j ; xi_j], cols 8-15 = m_n [xr_j ; xi_j]
//             D = [Re Phi_s  Re Phi_n ; Im Phi_s  Im Phi_n]   (full 8 x 8, both masks)
//          a wave owns 32 bins = 128 accumulator registers, 8 waves / workgroup
//
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_covar tools/ubench/mfma_covar.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

constexpr int C = 8, TB = 4, NBIN = 256, ROW = 272;  // ROW: slot stride in complex entries
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ inline float2 cmulc(float2 a, float2 b) {
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}

template <int H>
__device__ __forceinline__ void sweep(const float2* xt, const float* ms, const float* mn, int f,
                                      int reps, float (&dgs)[4], float (&dgn)[4],
                                      float2 (&ofs)[14], float2 (&ofn)[14]) {
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int tt = 0; tt < TB; ++tt) {
            float2 x[C];
#pragma unroll
            for (int c = 0; c < C; ++c) x[c] = xt[(tt * C + c) * ROW + f];
            const float ws = ms[tt * NBIN + f], wn = mn[tt * NBIN + f];
#pragma unroll
            for (int i = H; i < C; i += 2) {
                const float p = fmaf(x[i].x, x[i].x, x[i].y * x[i].y);
                dgs[i / 2] = fmaf(ws, p, dgs[i / 2]);
                dgn[i / 2] = fmaf(wn, p, dgn[i / 2]);
            }
            int k = 0;
#pragma unroll
            for (int i = 0; i < C; ++i)
#pragma unroll
                for (int j = i + 1; j < C; ++j) {
                    if ((k & 1) == H) {
                        const float2 p = cmulc(x[i], x[j]);
                        ofs[k / 2].x = fmaf(ws, p.x, ofs[k / 2].x);
                        ofn[k / 2].x = fmaf(wn, p.x, ofn[k / 2].x);
                        ofs[k / 2].y = fmaf(ws, p.y, ofs[k / 2].y);
                        ofn[k / 2].y = fmaf(wn, p.y, ofn[k / 2].y);
                    }
                    ++k;
                }
        }
    }
}

// tile: xt[(tt * C + c) * ROW + f] complex, masks ms/mn[tt][f]
__global__ __launch_bounds__(512, 4) void covar_valu(const float2* __restrict__ gx,
                                                     const float* __restrict__ gm, int reps,
                                                     float* __restrict__ out, long* cyc) {
    extern __shared__ float2 lds[];
    float2* xt = lds;
    float* ms = reinterpret_cast<float*>(xt + TB * C * ROW);
    float* mn = ms + TB * NBIN;
    for (int i = threadIdx.x; i < TB * C * ROW; i += 512) xt[i] = gx[i];
    for (int i = threadIdx.x; i < TB * NBIN; i += 512) {
        ms[i] = gm[i];
        mn[i] = 1.f - gm[i];
    }
    __syncthreads();
    const int f = threadIdx.x & 255, h = threadIdx.x >> 8;
    float dgs[4], dgn[4];
    float2 ofs[14], ofn[14];
    for (int e = 0; e < 4; ++e) dgs[e] = dgn[e] = 0.f;
    for (int e = 0; e < 14; ++e) ofs[e] = ofn[e] = make_float2(0.f, 0.f);
    const long t0 = clock64();
    if (h == 0) sweep<0>(xt, ms, mn, f, reps, dgs, dgn, ofs, ofn);
    else sweep<1>(xt, ms, mn, f, reps, dgs, dgn, ofs, ofn);
    const long t1 = clock64();
    // full [mask][i][j] (re, im) for checking
    float* o = out + ((size_t)blockIdx.x * NBIN + f) * 2 * C * C * 2;
    for (int i = 0; i < C; ++i)
        if ((i & 1) == h) {
            o[((0 * C + i) * C + i) * 2] = dgs[i / 2];
            o[((1 * C + i) * C + i) * 2] = dgn[i / 2];
            o[((0 * C + i) * C + i) * 2 + 1] = 0.f;
            o[((1 * C + i) * C + i) * 2 + 1] = 0.f;
        }
    int k = 0;
    for (int i = 0; i < C; ++i)
        for (int j = i + 1; j < C; ++j) {
            if ((k & 1) == h) {
                o[((0 * C + i) * C + j) * 2] = ofs[k / 2].x;
                o[((0 * C + i) * C + j) * 2 + 1] = ofs[k / 2].y;
                o[((1 * C + i) * C + j) * 2] = ofn[k / 2].x;
                o[((1 * C + i) * C + j) * 2 + 1] = ofn[k / 2].y;
                o[((0 * C + j) * C + i) * 2] = ofs[k / 2].x;
                o[((0 * C + j) * C + i) * 2 + 1] = -ofs[k / 2].y;
                o[((1 * C + j) * C + i) * 2] = ofn[k / 2].x;
                o[((1 * C + j) * C + i) * 2 + 1] = -ofn[k / 2].y;
            }
            ++k;
        }
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

__global__ __launch_bounds__(512, 2) void covar_mfma(const float2* __restrict__ gx,
                                                     const float* __restrict__ gm, int reps,
                                                     float* __restrict__ out, long* cyc) {
    extern __shared__ float2 lds[];
    float2* xt = lds;
    float* ms = reinterpret_cast<float*>(xt + TB * C * ROW);
    float* mn = ms + TB * NBIN;
    for (int i = threadIdx.x; i < TB * C * ROW; i += 512) xt[i] = gx[i];
    for (int i = threadIdx.x; i < TB * NBIN; i += 512) {
        ms[i] = gm[i];
        mn[i] = 1.f - gm[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, kk = lane >> 4;   // row / column index, k = 0..3
    const int ch = i16 & 7, toff = kk >> 1, comp = kk & 1;
    const bool im_row = i16 >= 8;                // A: rows 8-15 carry (xi, -xr)
    const float* mrow = (i16 < 8) ? ms : mn;     // B: columns 8-15 use the noise mask
    constexpr int BPW = NBIN / 8;                // 32 bins per wave
    v4f acc[BPW];
    for (int b = 0; b < BPW; ++b) acc[b] = (v4f){0.f, 0.f, 0.f, 0.f};
    const long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int tp = 0; tp < TB; tp += 2) {
            const float2* xrow = xt + ((tp + toff) * C + ch) * ROW + wave * BPW;
            const float* mr = mrow + (tp + toff) * NBIN + wave * BPW;
#pragma unroll
            for (int b = 0; b < BPW; ++b) {
                const float2 x = xrow[b];
                const float m = mr[b];
                const float braw = comp ? x.y : x.x;
                const float araw = im_row ? (comp ? -x.x : x.y) : braw;
                acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(araw, m * braw, acc[b], 0, 0, 0);
            }
        }
    }
    const long t1 = clock64();
    // D layout of 16x16x4: lane holds D[4 * (lane / 16) + r][lane % 16], r = 0..3
    for (int b = 0; b < BPW; ++b) {
        const int f = wave * BPW + b;
        float* o = out + ((size_t)blockIdx.x * NBIN + f) * 2 * C * C * 2;
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * (lane >> 4) + r, col = lane & 15;
            const int i = row & 7, part = row >> 3;  // part 0 = Re, 1 = Im
            const int j = col & 7, mk = col >> 3;
            o[((mk * C + i) * C + j) * 2 + part] = acc[b][r];
        }
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    const int reps = 400, blocks = 512;
    std::vector<float2> hx(TB * C * ROW);
    std::vector<float> hm(TB * NBIN);
    srand(1);
    for (auto& v : hx) v = make_float2(rand() / (float)RAND_MAX - 0.5f, rand() / (float)RAND_MAX - 0.5f);
    for (auto& v : hm) v = rand() / (float)RAND_MAX;
    float2* dx; float* dm; float *o1, *o2; long* dc;
    hipMalloc(&dx, hx.size() * 8); hipMalloc(&dm, hm.size() * 4);
    const size_t on = (size_t)blocks * NBIN * 2 * C * C * 2;
    hipMalloc(&o1, on * 4); hipMalloc(&o2, on * 4); hipMalloc(&dc, 16);
    hipMemcpy(dx, hx.data(), hx.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dm, hm.data(), hm.size() * 4, hipMemcpyHostToDevice);
    const size_t lds = TB * C * ROW * 8 + 2 * TB * NBIN * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(covar_valu), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(covar_mfma), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bin_frames = (double)blocks * NBIN * TB * reps;
    for (int which = 0; which < 2; ++which) {
        auto launch = [&](int rp) {
            if (which == 0) covar_valu<<<blocks, 512, lds>>>(dx, dm, rp, o1, dc);
            else covar_mfma<<<blocks, 512, lds>>>(dx, dm, rp, o2, dc);
        };
        launch(4);
        hipDeviceSynchronize();
        hipEventRecord(e0); launch(reps); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long cyc; hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
        // per CU: 2 workgroups resident, blocks / 512 rounds
        printf("%s: %.3f ms for %.3g bin-frames -> %.1f G bin-frames/s; one workgroup: %.1f shader "
               "cycles per (bin, frame) slot of its 256 bins\n",
               which == 0 ? "valu" : "mfma", ms, bin_frames, bin_frames / ms / 1e6,
               (double)cyc / ((double)reps * TB * NBIN));
    }
    // agreement (single sweep)
    covar_valu<<<1, 512, lds>>>(dx, dm, 1, o1, dc);
    covar_mfma<<<1, 512, lds>>>(dx, dm, 1, o2, dc);
    std::vector<float> a(NBIN * 2 * C * C * 2), b(a.size());
    hipMemcpy(a.data(), o1, a.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), o2, b.size() * 4, hipMemcpyDeviceToHost);
    double num = 0, den = 0;
    for (size_t i = 0; i < a.size(); ++i) { num += (a[i] - b[i]) * (double)(a[i] - b[i]); den += a[i] * (double)a[i]; }
    printf("mfma vs valu covariance: rel rms %.2e\n", sqrt(num / den));
    return 0;
}
