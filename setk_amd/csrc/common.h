// common.h -- argument blocks and launcher prototypes shared by the kernel
// translation units and the C-ABI front end (capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace setk {

constexpr int kNfft = 512;        // fused kernels are specialised for n_fft = 512
constexpr int kBins = 257;        // n_fft / 2 + 1
constexpr int kBinsPad = 264;     // row pitch of per-bin planes (multiple of 8)
constexpr int kMaxChannels = 8;   // register-resident covariance accumulators
constexpr int kMaxChannels16 = 16; // modular (unfused) operators: covariance, solve, beamform
// pass 2: frames per super-tile = quad-rows per workgroup (16 lanes each) and the
// waves per SIMD its register budget is set for (tools/mk_abl.sh -DSETK_P2_ST=.. -DSETK_P2_WAVES=..)
#ifndef SETK_P2_ST
#define SETK_P2_ST 16
#endif
#ifndef SETK_P2_WAVES
#define SETK_P2_WAVES 2
#endif
constexpr int kSuperTile = SETK_P2_ST;    // frames per inverse-transform batch (pass 2)
constexpr int kPass2Threads = 16 * kSuperTile;
constexpr int kMaxKeep = 7;       // ceil(512 / hop) - 1 for hop >= 64

// number of Hermitian pairs (i <= j)
__host__ __device__ constexpr int npairs(int c) { return c * (c + 1) / 2; }
// pass 1: frames per tile -- 32/C transforms fill the 32 transform quad-rows,
// capped at 8 (small C then rotates several quad-row sets)
__host__ __device__ constexpr int pass1_tile_frames(int c) { return (32 / c) < 8 ? (32 / c) : 8; }
// planes of one packed covariance pair set: [s.re | s.im | n.re | n.im | sums(2)]
__host__ __device__ constexpr int nplanes_partial(int c) { return 4 * npairs(c) + 2; }
// upper-triangular row-major pair index, i <= j
__host__ __device__ constexpr int pair_index(int i, int j, int c) {
    return i * c - i * (i - 1) / 2 + (j - i);
}

// sample formats of UttDesc::audio
constexpr int kAudioF32 = 0;    // float32 [C][ch_stride]   (the reference's C x N, read_wav's int16 / 32768)
constexpr int kAudioPcm16 = 1;  // int16   [C][ch_stride]   planar 16-bit PCM as stored in the wave file
                                // (setk_pcm16_deinterleave_batch); the kernels scale by 2^-15, exactly
                                // libs/utils.py:80-90's dtype="float32" read.  ch_stride is even: a
                                // pair of samples (x[2n], x[2n+1]) is ONE aligned dword

struct UttDesc {
    const float* audio;   // float32 [C][num_samples], or (audio_fmt) int16 [C][ch_stride]
    const float* mask_s;  // [T][F]
    const float* mask_n;  // [T][F] or null
    float* wave_f32;      // pass-2 float output [out_len]
    void* wave_out;       // final output (float32 or int16), may alias wave_f32
    int num_samples;
    int num_frames;
    int out_len;
    int part0;   // first partial slab of this utterance
    int nparts;  // number of partial slabs
    int audio_fmt;  // kAudioF32 | kAudioPcm16
    int ch_stride;  // kAudioPcm16: samples between channels (even, >= num_samples)
    int pad_;
};

struct WorkItem {
    int utt;
    int t0, t1;  // frame range [t0, t1)
    int part;    // partial slab index (pass 1) / unused (pass 2)
    int last;    // 1 if t1 == num_frames
};

struct StftGeom {
    int hop;
    int pad;      // n_fft/2 when center else 0
    int center;
    int keep;     // ceil(n_fft / hop) - 1
};

struct Pass1Args {
    const UttDesc* utts;
    const WorkItem* items;
    float* partials;        // [nparts_total][nplanes][kBinsPad]
    const float* window;    // [512] analysis window (padded, centred); PCM16 launches: x 2^-15
    const float2* tw256;    // [16][16]  exp(-2 pi i la q / 256) at [q*16+la]
    const float2* tw512;    // [129]     exp(-2 pi i k / 512)
    unsigned* norm_bits;    // [n_utts] max |audio| as float bits (atomicMax)
    float* spec_dump;       // DUMP mode: [C][T][F]
    int dump_pitch;         // DUMP mode: row pitch in complex entries (0 = F)
    StftGeom g;
    int flags;
    // matrix-core transforms (pass1_mc.hip / mcdft.h)
    const unsigned* mc_tab;  // [mc::kTabWords][64] operand tiles
    const float* mc_win;     // [8][64] analysis window rows x 2^10 / peak
};

struct FinalizeArgs {
    const UttDesc* utts;
    const float* partials;
    float* covar;  // [n_utts][4*NP (+2*NP when with_ry)][kBinsPad]
    int num_channels;
    int with_ry;   // also emit Ry = (Rs_num + Rn_num) / T  (MPDR)
    float num_scale;  // numerators x this (matrix-core pass 1: (peak / 2^10)^2; else 1)
};

struct SolveArgs {
    const float* covar;   // packed planes, see FinalizeArgs
    float* weight;        // [n_utts][C][kBinsPad] float2
    int* status;          // [n_utts] (atomicMax) or per-bin when per_bin != 0
    int* bin_status;      // [n_problems] or null
    double* snr_acc;      // PMWF ref<0: [n_problems][C][2] per-bin (ps, pn)
    float* wmat;          // PMWF ref<0: [n_problems][C][C] float2 (column major)
    int n_utts;
    int num_bins;
    int num_channels;
    int planes;           // planes per utterance in covar
    int kind, flags, rank1, pmwf_ref;
    float pmwf_beta;
    // fused partial reduction (enhance_batch with few slabs per utterance and no covariance
    // taps): the solve sums pass 1's partial slabs itself, exactly as covar_finalize_kernel
    // would have (same order, same float32 expressions), and that launch falls away
    const float* partials;  // null: read `covar`
    const UttDesc* utts;
    float num_scale;
};

struct Pass2Args {
    const UttDesc* utts;
    const WorkItem* items;
    const float* weight;     // [n_utts][C][kBinsPad] float2
    const float* spec_in;    // ISTFT mode: [B][T][F] float2
    const float* window;     // analysis window [512]
    const float* synwin;     // synthesis window [512] (same padded window)
    const float* winsq;      // window^2 [512]
    const float2* tw256;
    const float2* tw512;
    unsigned* outmax_bits;   // [n_utts]
    const unsigned* norm_bits;  // [n_utts] max |audio| (pass 1): pass2_mc's input range (may be null)
    StftGeom g;
    int flags;
    // matrix-core transforms (pass2_mc.hip / mcdft.h)
    const unsigned* mc_tab;  // [mc::kTabWords][64]
    const float* mc_win;     // [8][64] analysis window rows x 2^10 / peak
    const float* mc_syn;     // [8][64] synthesis window rows x peak / 2^10 / 512 / sum(window^2)
    const float* mc_edge;    // [8][64] corrections of the single-contribution blocks (first | last)
};

struct ScaleArgs {
    const UttDesc* utts;
    const unsigned* norm_bits;
    const unsigned* outmax_bits;
    const float* norm_override;  // istft API: per-batch norm (<=0: none) or null
    int pcm16;
};

// launchers (implemented in the kernel TUs)
hipError_t launch_pass1(int C, bool dump, const Pass1Args& a, int n_items, hipStream_t s, bool pcm16 = false);
hipError_t launch_pass1_mc(int C, const Pass1Args& a, int n_items, hipStream_t s);
bool pass1_mc_supported(int C, int hop);
hipError_t launch_pass2_mc(int C, const Pass2Args& a, int n_items, hipStream_t s, bool pcm16 = false);
int pass2_mc_wgs_per_cu(int C, bool pcm16 = false);
// STFT of whole utterances into the bin-major [F][C][Tp] layout of cgmm_bin.hip
// (items: 64-frame blocks; UttDesc::wave_out = the utterance's output)
hipError_t launch_stft_binmajor(int C, const Pass1Args& a, int n_items, hipStream_t s);
hipError_t launch_finalize(const FinalizeArgs& a, int n_utts, hipStream_t s);
hipError_t launch_solve(const SolveArgs& a, hipStream_t s);
hipError_t launch_pmwf_select(const SolveArgs& a, int* ref_out, hipStream_t s);
hipError_t launch_pass2(int C, bool istft_only, const Pass2Args& a, int n_items, hipStream_t s);
hipError_t launch_scale(const ScaleArgs& a, int n_utts, int max_len, hipStream_t s);
size_t pass2_lds_bytes(int C, int keep);

// modular helpers
hipError_t launch_covar_spec(int C, const float* spec, const float* mask, int T, int F,
                             float* partials, int t_split, hipStream_t s);
hipError_t launch_covar_spec_finalize(int C, const float* partials, int nparts, int F,
                                      float* covar_fcc, hipStream_t s);
hipError_t launch_pack_covar(const float* fcc, int F, int C, int Cp, float pad_diag,
                             float* planes, int plane0, hipStream_t s);
hipError_t launch_unpack_weight(const float* wplanes, int F, int C, float* w_fc, hipStream_t s);
hipError_t launch_unpack_covar(const float* planes, int n_utts, int n_planes, int plane0, int F,
                               int C, float* out, hipStream_t s);
hipError_t launch_unpack_weight_batch(const float* wplanes, int n_utts, int F, int C, float* w,
                                      hipStream_t s);
// n_fft that is not a power of two: Bluestein tables (device), M = 0 when unused.
// `tw` of the generic launchers then holds exp(-2 pi i k / M), k < M / 2.
struct BluesteinPlan {
    int M;                 // convolution length, power of two >= 2 n_fft - 1
    const float* chirp;    // [n_fft] float2 exp(-i pi k^2 / n_fft)
    const float* bhat_br;  // [M] float2 FFT_M of the wrapped conj(chirp), bit-reversed order
};
hipError_t launch_stft_generic(const float* audio, int C, int n_samp, int T, int n_fft, int hop,
                               int pad, const float* window, const float* tw, float* spec,
                               const BluesteinPlan* bp, hipStream_t s);
hipError_t launch_istft_generic(const float* spec, int B, int T, int n_fft, int hop, int pad,
                                int out_len, const float* window, const float* winsq,
                                const float* tw, float* frames, float* wave, unsigned* outmax,
                                const float* norm, int T_eff, const BluesteinPlan* bp,
                                hipStream_t s);
size_t cgmm_fill_args(void* args_out, int C, const float* spec, int T, int F,
                      const float* init_mask, float* gamma_opt, float* mask_out, void* scratch,
                      int update_alpha, int spec_pitch);
size_t cgmm_args_bytes();
size_t cgmm_scratch_bytes(int C, int T, int F);
hipError_t launch_cgmm_batch(int C, const void* d_tbl, int n_utts, int F, int max_frames,
                             int num_iters, hipStream_t s);
// general EM: any number of classes (<= 4), up to 16 channels (cgmm_k.hip)
size_t cgmm_k_work_bytes(int K, int T, int F);
bool cgmm_k_supported(int C, int K);
hipError_t launch_cgmm_k(const float* spec, const double* gamma0, const float* init_mask, float* gamma_out,
                         double* work, int* status, int C, int T, int F, int K, int num_iters,
                         int update_alpha, hipStream_t s);
// bin-resident EM (cgmm_bin.hip)
size_t cgmm_bin_args_bytes();
int cgmm_bin_timing_slots();
int cgmm_bin_pitch(int T);
int cgmm_bin_threads(int C, int max_frames);
void cgmm_bin_fill_args(void* out, const float* xb, const float* init_mask, float* gamma_bm, int T,
                        int F, int update_alpha, int nout, void* timing);
hipError_t launch_cgmm_bin(int C, const void* d_tbl, const float* const* d_spec_ptrs, int spec_pitch,
                           float* const* d_mask_ptrs, float* const* d_gamma_ptrs, int n_utts, int F,
                           int max_frames, int num_iters, int nout, hipStream_t s);
hipError_t launch_ban(const float* w, const float* Rn, int F, int C, float* out, hipStream_t s);
hipError_t launch_rank1(const float* pv, const float* Rs, const float* Rn, int F, int C,
                        float* out, hipStream_t s);
hipError_t launch_directional_feats(const float* spec, const float* sv, const int* pairs, int n_pairs,
                                    int C, int T, int F, float* out, hipStream_t s);
hipError_t launch_maxabs(const UttDesc* utts, int C, unsigned* norm_bits, int n_utts, int max_samples,
                         hipStream_t s);
hipError_t launch_pack_fixed_weights(const float* sets, const int* index, int n_utts, int C,
                                     float* out, hipStream_t s);
hipError_t launch_float_to_pcm16(const float* in, int C, int N, int16_t* out, hipStream_t s);
hipError_t launch_pcm16_to_float(const int16_t* pcm, int C, int N, float* out, hipStream_t s);
size_t cm_item_bytes();
void cm_item_fill(void* tbl, int i, const void* src, float* dst, float vmin, float vrange, int rows, int cols,
                  int kind, int transpose);
hipError_t launch_kaldi_cm_decode_batch(const void* d_items, int n, long max_elems, hipStream_t s);
size_t pcm_item_bytes();
void pcm_item_fill(void* dst, int i, const int16_t* pcm, float* out, int n);
hipError_t launch_pcm16_to_float_batch(const void* d_items, int n_utts, int C, int max_n,
                                       double* power0, hipStream_t s);
void pcm_item_fill_planar(void* dst, int i, const int16_t* pcm, int16_t* out, int n, int stride);
hipError_t launch_pcm16_deinterleave_batch(const void* d_items, int n_utts, int C, int max_n,
                                           double* power0, hipStream_t s);
hipError_t launch_beamform_spec(const float* w_fc, const float* spec, int C, int T, int F,
                                float* out, hipStream_t s);

// WPE (wpe.hip)
bool wpe_supported(int N, int taps);
const char* wpe_limit_message(int N, int taps);
hipError_t launch_wpe_transpose(const float* in, int C, int T, int F, float* out, bool to_fct,
                                hipStream_t s);
hipError_t launch_wpe_lambda(const float* d_fct, int C, int T, int F, int ctx, double* lam,
                             hipStream_t s);
hipError_t launch_wpe_lambda_from_enh(const float* enh_tf, int T, int F, double* lam,
                                      hipStream_t s);
hipError_t launch_wpe_inv_lambda(const double* lam, int T, int F, float* out, hipStream_t s);
size_t wpe_args_bytes();
void wpe_fill_args(void* dst, const float* x_fct, const double* lam, float* out_fct, int* status,
                   int N, int T, int taps, int delay, long long* timing = nullptr,
                   void* rwork = nullptr);
// bytes of global scratch per (utterance, bin) when channels x taps does not fit LDS, else 0
size_t wpe_wide_bytes_per_bin(int N, int taps);
hipError_t launch_wpe_step_batch(const void* d_tbl, int n_utts, int N, int F, int taps,
                                 hipStream_t s);

}  // namespace setk
