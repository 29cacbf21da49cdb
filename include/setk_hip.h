/*
 * setk_hip.h -- C ABI of libsetk_hip.so: the MI355X (gfx950) implementation of
 * setk's mask-based adaptive-beamformer hot path
 *
 *     STFT -> masked spatial covariance -> MVDR/GEV/PMWF/MPDR weights
 *          -> beamform -> iSTFT (+ max-abs renorm)
 *
 * The reference (funcwj/setk) has no FFI for this path: the seam is the python
 * call surface of scripts/sptk/libs/{utils,beamformer}.py.  Each entry point
 * below names the reference function it replaces; the python mirror in
 * setk_amd/libs binds them through ctypes (see INTEGRATION.md for the stub a
 * maintainer would add on the reference side).
 *
 * Conventions
 *   - plain C, no exceptions; every call returns an int status:
 *       0            SETK_OK
 *       < 0          API misuse / unsupported  (python: ValueError/RuntimeError)
 *       > 0          never returned; numerical failures are reported per
 *                    frequency bin / per utterance through `status` arrays
 *                    (python: numpy.linalg.LinAlgError)
 *   - all data pointers are caller-owned and contiguous.  Every pointer may be
 *     DEVICE memory (hipMalloc / torch tensor data_ptr) or ordinary HOST
 *     memory; the library detects which (hipPointerGetAttributes) and stages
 *     host buffers through its own device arena.  setk_enhance_batch takes
 *     device pointers only.
 *   - complex values are interleaved float pairs (numpy complex64).
 *   - spectrogram layout is "time major": X[c][t][f], f fastest, F = n_fft/2+1.
 *     (numpy view of the reference's N x F x T array:
 *      np.ascontiguousarray(obs.transpose(0, 2, 1)))
 *   - masks are T x F row major (the layout apply_adaptive_beamformer.py
 *     normalises to, :146-151), covariances F x C x C, weights F x C.
 *   - all work is enqueued on the hipStream_t passed as `stream` (NULL = the
 *     default stream).  When any OUTPUT pointer is host memory the call
 *     synchronises the stream before returning; otherwise it is asynchronous.
 *   - a handle is used by one host thread at a time; no global state.
 */
#ifndef SETK_HIP_H_
#define SETK_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SETK_ABI_VERSION 1

/* return codes */
#define SETK_OK 0
#define SETK_ERR_INVALID (-1)     /* bad argument / shape mismatch            */
#define SETK_ERR_UNSUPPORTED (-2) /* valid request outside the built kernels  */
#define SETK_ERR_HIP (-3)         /* HIP runtime error, see setk_last_error   */
#define SETK_ERR_NOMEM (-4)

/* per-bin / per-utterance numerical status values (written to status arrays) */
#define SETK_NUM_OK 0
#define SETK_NUM_SINGULAR 1 /* noise covariance not positive definite        */
#define SETK_NUM_NOCONV 2   /* Jacobi sweep limit reached                    */
#define SETK_NUM_NONFINITE 3
#define SETK_NUM_RANKDEF 4  /* WPE only, NOT an error: the tap correlation was rank deficient
                              (fewer frames than channels x taps, a silent or duplicated channel),
                              columns at the noise level were dropped and a filter was produced;
                              numpy.linalg.solve in the reference also goes through on such
                              input, with a different, noise-determined filter */

/* beamformer kinds (apply_adaptive_beamformer.py:22, --beamformer) */
#define SETK_BF_MVDR 0
#define SETK_BF_GEVD 1
#define SETK_BF_PMWF 2        /* beta / ref channel / rank1 via setk_bf_opts */
#define SETK_BF_MPDR 3
#define SETK_BF_MPDR_WHITEN 4

/* rank-1 approximation of Rs for PMWF (libs/beamformer.py:66-84, 642-645) */
#define SETK_RANK1_NONE 0
#define SETK_RANK1_EIG 1
#define SETK_RANK1_GEV 2

/* flags */
#define SETK_FLAG_BAN 0x1        /* blind analytic normalisation, do_ban      */
#define SETK_FLAG_CLAMP_MASK 0x2 /* speech mask <- min(mask, 1)   (:141)      */
#define SETK_FLAG_POST_MASK 0x4  /* enh <- enh * mask^T           (:174-175)  */
#define SETK_FLAG_NO_GAUGE 0x8   /* leave eigenvector phase as computed       */
#define SETK_FLAG_OUT_PCM16 0x10 /* enhance_batch writes int16 PCM, not f32   */
#define SETK_FLAG_NO_RENORM 0x20 /* apply_weights_batch: inverse_stft(norm=None) */
/* Reference failure semantics (setk_weights, setk_enhance_batch; CLI --strict-reference true).
 * The reference solves Rn x = b with numpy.linalg.solve (LAPACK ?gesv: LU with partial pivoting
 * in the matrix's own precision, complex64) and raises LinAlgError("Singular matrix") only on an
 * EXACT zero pivot (libs/beamformer.py:536 MVDR, :568 MPDR on Ry, :646 PMWF; the CLI logs and
 * skips the utterance, apply_adaptive_beamformer.py:170-172); its GEV never raises (hegvd's
 * refusal is caught and scipy.linalg.eig takes over, libs/beamformer.py:54-59).  With this flag
 * the same decision is taken on the device: a complex64 LU with partial pivoting (cabs1 pivot
 * search, no fused multiply-adds) of the matrix the reference would hand to solve, per bin, and
 * SETK_NUM_SINGULAR where a pivot column is exactly zero -- a duplicated / silent / power-of-two
 * scaled channel, an all-zero noise covariance; GEV goes through even on an all-zero Rn (as the
 * pencil (Rs, I)).  Without the flag (the default) such input is regularised (floored / loaded
 * Cholesky) and only an all-zero or non-finite covariance is refused.  The weights of the bins
 * that pass are the same in both modes. */
#define SETK_FLAG_STRICT_REFERENCE 0x40
/* setk_enhance_batch(_taps): audio[u] is 16-bit PCM, planar int16 [C][setk_pcm16_channel_stride(
 * num_samples[u])] as setk_pcm16_deinterleave_batch lays it out, not float32 [C][N].  Both
 * streaming kernels then read 2 bytes per sample and scale by 2^-15 inside their transforms
 * (folded into the window tables: a power of two, so the results are bit for bit those of the
 * float32 call on pcm / 32768, i.e. on read_wav's dtype="float32" samples, libs/utils.py:80-90).
 * Needs hop = n_fft / 2 (the CLI default 512 / 256); otherwise SETK_ERR_UNSUPPORTED -- convert
 * with setk_pcm16_to_float_batch. */
#define SETK_FLAG_IN_PCM16 0x80

typedef struct setk_context* setk_handle_t;

typedef struct setk_bf_opts {
    int kind;        /* SETK_BF_*                                            */
    int flags;       /* SETK_FLAG_*                                          */
    float pmwf_beta; /* 0 -> pmwf-0, 1 -> pmwf-1                             */
    int pmwf_ref;    /* < 0: pick the channel with max estimated SNR         */
    int rank1;       /* SETK_RANK1_*                                         */
} setk_bf_opts;

/* ---- lifetime ---------------------------------------------------------- */
int setk_abi_version(void);
int setk_create(setk_handle_t* out, int device_ordinal);
int setk_destroy(setk_handle_t h);
/* text of the last error on this handle (never NULL) */
const char* setk_last_error(setk_handle_t h);
/* PCI bus id of the handle's device ("0000:c1:00.0", NUL terminated, into out[len]): what a
 * host needs to place its reader threads and pinned staging on the GPU's NUMA node
 * (/sys/bus/pci/devices/<id>/numa_node; setk_amd/numa.py, --numa auto).  The reference has no
 * such notion: its processes are CPU only (scripts/run_adapt_beamformer.sh:69-92). */
int setk_device_pci_bus_id(setk_handle_t h, char* out, int len);

/* ---- host-memory plumbing (used by the streaming host pipeline) -----------
 * Pin an existing host range (e.g. the mmap of a wave file in the page cache) so
 * that setk_memcpy_h2d_async DMAs straight out of it, and release it again once
 * the copy has completed.  Thread safe; independent of the handle's arena. */
int setk_host_register(setk_handle_t h, void* ptr, size_t bytes);
int setk_host_unregister(setk_handle_t h, void* ptr);
int setk_memcpy_h2d_async(setk_handle_t h, void* dst, const void* src, size_t bytes,
                          void* stream);
int setk_memcpy_d2h_async(setk_handle_t h, void* dst, const void* src, size_t bytes,
                          void* stream);
/* Buffers, streams and events on the handle's device, for a host program that brings no HIP
 * runtime binding of its own (setk_amd/pipeline.py runs without importing torch): device and
 * page-locked host memory, non-blocking streams (pass them as the `stream` argument of any
 * entry point), events without timing.  Thread safe.  What the reference has in their place:
 * nothing -- it is single-threaded numpy (apply_adaptive_beamformer.py:130-178). */
int setk_device_alloc(setk_handle_t h, size_t bytes, void** out);
int setk_device_free(setk_handle_t h, void* ptr);
int setk_host_alloc(setk_handle_t h, size_t bytes, void** out);
int setk_host_free(setk_handle_t h, void* ptr);
/* The read stage of the command line's pipeline for one batch in one call (no handle: host code
 * only).  Payload i is nbytes[i] bytes at offsets[i] of the file paths[i] -- a wave file's 16-bit
 * frames, a mask's float32 rows, exactly as stored -- and is copied to dst[i] (the batch's
 * page-locked slab) by a process-wide pool of n_threads native threads: a MADV_SEQUENTIAL
 * mapping + memcpy for payloads of at least mmap_min_bytes, pread below.  status[i]: 0 or the
 * errno of the failing call (EIO: the file ends inside the payload).  Thread safe; concurrent
 * calls share the pool.  What the reference has in its place: WaveReader / ScriptReader._load,
 * one utterance at a time on the interpreter's thread (libs/data_handler.py:345-413). */
int setk_host_read_payloads(int n, const char* const* paths, const long long* offsets,
                            const long long* nbytes, void* const* dst, int n_threads,
                            long long mmap_min_bytes, int* status);
int setk_stream_create(setk_handle_t h, void** out);
int setk_stream_destroy(setk_handle_t h, void* stream);
int setk_stream_synchronize(setk_handle_t h, void* stream);
int setk_stream_wait_event(setk_handle_t h, void* stream, void* event);
int setk_event_create(setk_handle_t h, void** out);
int setk_event_destroy(setk_handle_t h, void* event);
int setk_event_record(setk_handle_t h, void* event, void* stream);
int setk_event_synchronize(setk_handle_t h, void* event);

/* ---- STFT plan ---------------------------------------------------------
 * Mirrors the arguments of forward_stft / inverse_stft
 * (scripts/sptk/libs/utils.py:96-173) after the host resolved
 * n_fft = nextpow2(frame_len) | frame_len and evaluated the window
 * (scipy.signal.get_window(name, frame_len, fftbins=True) or sqrt-hann).
 * `window` = frame_len host floats, NULL = periodic hann.
 * n_fft must be even, in [16, 4096]; n_fft == 512 selects the register/LDS
 * radix-16 kernels, other powers of two (>= 64) a generic LDS radix-2 kernel,
 * anything else (--round-power-of-two false with e.g. frame_len 400) Bluestein's
 * chirp-z form on top of the radix-2 kernel. */
int setk_stft_plan(setk_handle_t h, int frame_len, int frame_hop, int n_fft,
                   int center, const float* window);
/* frames produced for `num_samples` input samples, < 0 on error */
int setk_stft_num_frames(setk_handle_t h, int num_samples);
/* samples produced by the inverse for `num_frames` frames (nsamps < 0: the
 * librosa default length) */
int setk_istft_num_samples(setk_handle_t h, int num_frames, int nsamps);

/* ---- modular operators -------------------------------------------------- */

/* forward_stft per channel (libs/utils.py:96-138; SpectrogramReader._load,
 * libs/data_handler.py:492-503).  audio[C][N] float32 -> spec[C][T][F]. */
int setk_stft(setk_handle_t h, const float* audio, int num_channels,
              int num_samples, float* spec, void* stream);

/* The same for a batch of utterances in ONE launch (device pointers, host tables;
 * spec[u] = [C][T_u][spec_pitch], F entries used per row; spec_pitch = 0 means F, a
 * multiple of 16 entries keeps every row 128-byte aligned for a streaming consumer;
 * n_fft = 512 plan, C <= 8).  Asynchronous on `stream`. */
int setk_stft_batch(setk_handle_t h, int n_utts, int num_channels,
                    const float* const* audio, const int* num_samples,
                    float* const* spec, int spec_pitch, void* stream);

/* inverse_stft (libs/utils.py:142-173) for `batch` independent spectrograms
 * spec[B][T][F] -> wave[B][L], L = setk_istft_num_samples(h, T, nsamps).
 * norm: NULL or B floats; norm[b] > 0 rescales wave b to that max-abs
 * (samps * norm / (max|samps| + eps), :166-168). */
int setk_istft(setk_handle_t h, const float* spec, int batch, int num_frames,
               int nsamps, const float* norm, float* wave, void* stream);

/* compute_covar (libs/beamformer.py:87-103):
 * covar[f] = sum_t m[t][f] x x^H / max(sum_t m[t][f], 1e-6).
 * spec[C][T][F], mask[T][F] -> covar[F][C][C] complex64.  1 <= C <= 16 (the wide
 * kernel serves 9 - 16 channels). */
int setk_covar(setk_handle_t h, const float* spec, const float* mask,
               int num_channels, int num_frames, int num_bins, float* covar,
               void* stream);

/* solve_pevd (libs/beamformer.py:31-63): principal eigenvector of Rs (Rn NULL)
 * or of the pencil (Rs, Rn).  Output pvec[F][C] complex64, unit 2-norm
 * (Rn NULL) or Rn-normalised v^H Rn v = 1.  Gauge (unless SETK_FLAG_NO_GAUGE):
 * component 0 real >= 0; for the pencil the rule applies to y = L^H v,
 * Rn = L L^H.  status[F] (may be NULL) receives SETK_NUM_*. */
int setk_pevd(setk_handle_t h, const float* Rs, const float* Rn, int num_bins,
              int num_channels, int flags, float* pvec, int* status,
              void* stream);

/* {Mvdr,Gevd,Pmwf,Mpdr}Beamformer.weight (+ do_ban), libs/beamformer.py:
 * 14-28, 527-539, 555-571, 632-659, 674-682.  Rs, Rn: [F][C][C]; Ry only for
 * MPDR kinds (covariance of the all-ones mask).  weight[F][C] complex64.
 * status[F] may be NULL.  ref_out (may be NULL) receives the PMWF reference
 * channel actually used. */
int setk_weights(setk_handle_t h, const setk_bf_opts* opts, const float* Rs,
                 const float* Rn, const float* Ry, int num_bins,
                 int num_channels, float* weight, int* status, int* ref_out,
                 void* stream);

/* read_wav with dtype float32 + transpose (libs/utils.py:65-92,
 * data_handler.py:372-393) for 16-bit PCM: interleaved frames pcm[n][c] ->
 * audio[c][n] = pcm / 32768 (soundfile's scaling, exact in float32).  Lets a
 * caller upload the 2-byte samples of a wav as they lie in the file. */
int setk_pcm16_to_float(setk_handle_t h, const int16_t* pcm, int num_channels,
                        int num_samples, float* audio, void* stream);

/* The same for a batch in ONE launch (device pointers; the tables are host
 * arrays): what the streaming host pipeline runs on a staging slab of wav
 * payloads.  power0 (device double[n_utts], may be NULL) receives sum(x0^2) of
 * channel 0 -- SpectrogramReader.power * N (data_handler.py:398-403), which the
 * CLI only logs.  Asynchronous on `stream`. */
int setk_pcm16_to_float_batch(setk_handle_t h, int n_utts, int num_channels,
                              const int16_t* const* pcm, const int* num_samples,
                              float* const* audio, double* power0, void* stream);

/* 16-bit PCM ingest WITHOUT a float32 copy: interleaved frames pcm[u][n][c] (a wave file's
 * data chunk) -> planar int16 out[u][c][stride], stride = setk_pcm16_channel_stride(num_samples[u])
 * (the samples between channels: num_samples rounded up to a multiple of 8, the padding
 * zeroed), the layout SETK_FLAG_IN_PCM16 reads.  4 C N bytes of traffic where the float32
 * conversion moves 6 C N, and half the bytes in both streaming passes afterwards.  power0 as
 * setk_pcm16_to_float_batch.  Device pointers, host tables, asynchronous on `stream`. */
int setk_pcm16_channel_stride(int num_samples);
int setk_pcm16_deinterleave_batch(setk_handle_t h, int n_utts, int num_channels,
                                  const int16_t* const* pcm, const int* num_samples,
                                  int16_t* const* out, double* power0, void* stream);

/* Kaldi CompressedMatrix bodies -> float32 matrices on the device, a batch per launch: the masks
 * of the streaming command line travel as the 1 - 2 bytes per element the archive holds.
 * Replaces `uncompress` (scripts/sptk/libs/kaldi_io.py:248-292; formats CM / CM2 / CM3 =
 * kOneByteWithColHeaders / kTwoByte / kOneByte) with the same float32 operations in the same
 * order: results equal numpy's bit for bit.  Per matrix i: kinds[i] (SETK_KALDI_CM..CM3), the global
 * header's (vmin, vrange, rows, cols), src[i] = the bytes that follow that 16-byte header (CM: the
 * per-column headers, then the column-major bytes), dst[i] = rows x cols float32 row-major -- or, with
 * transpose[i] != 0, cols x rows: the transpose (a mask stored F x T, apply_adaptive_beamformer.py:
 * 146-151).  Device pointers (CM / CM2 bodies 2-byte aligned), host tables, asynchronous on `stream`. */
#define SETK_KALDI_CM 1
#define SETK_KALDI_CM2 2
#define SETK_KALDI_CM3 3
int setk_kaldi_cm_decode_batch(setk_handle_t h, int n, const int* kinds, const float* vmin, const float* vrange,
                               const int* rows, const int* cols, const int* transpose,
                               const void* const* src, float* const* dst, void* stream);

/* The way back for a multi-channel result (apply_wpe.py:58-61 -> write_wav,
 * libs/utils.py:45-62 -> soundfile.write, float -> PCM_16): float32 rows audio[C][N] ->
 * interleaved frames pcm[N][C], lrint(x * 32767) without clipping (libsndfile's default;
 * out-of-range values wrap).  Host or device pointers. */
int setk_float_to_pcm16(setk_handle_t h, const float* audio, int num_channels, int num_samples,
                        int16_t* pcm, void* stream);

/* do_ban (libs/beamformer.py:14-28) on an arbitrary weight:
 * out[f] = w[f] * sqrt(|w^H Rn Rn w|) / max(Re w^H Rn w, eps_f32). */
int setk_ban(setk_handle_t h, const float* weight, const float* Rn, int num_bins,
             int num_channels, float* out, void* stream);

/* rank1_constraint (libs/beamformer.py:66-84): p = principal eigenvector of Rs
 * (Rn NULL) or Rn v for the pencil (Rs, Rn);
 * out[f] = tr(Rs) / max(tr(p p^H), eps_f32) * p p^H,  [F][C][C] complex64. */
int setk_rank1(setk_handle_t h, const float* Rs, const float* Rn, int num_bins,
               int num_channels, float* out, int* status, void* stream);

/* Beamformer.beamform (libs/beamformer.py:220-234):
 * out[t][f] = sum_c conj(w[f][c]) spec[c][t][f]. */
int setk_beamform(setk_handle_t h, const float* weight, const float* spec,
                  int num_channels, int num_frames, int num_bins, float* out,
                  void* stream);

/* ---- CGMM mask estimation (BASELINE configs[4]) ---------------
 * CgmmTrainer(obs, 2, gamma=init).train(num_iters) of
 * scripts/sptk/libs/cluster.py:396-465 as used by estimate_cgmm_masks.py:44-64:
 * K = 2 complex-Gaussian mixture, alpha fixed at 1/2 (or SETK_CGMM_UPDATE_ALPHA), deterministic start
 * (Rs = x x^H / T, Rn = I) or an initial speech mask.
 * spec[C][T][F] complex64, init_mask[T][F] or NULL.
 * gamma_out (may be NULL) receives the posteriors [2][T][F]; mask_out[T][F]
 * receives gamma[0] (the speech mask the CLI saves).  1 <= C <= 8. */
#define SETK_CGMM_UPDATE_ALPHA 0x1 /* --update-alpha: alpha_k = mean_t gamma_k in every M-step
                                     (Cgmm.update, cluster.py:246-257) instead of 1/2 */
int setk_cgmm_masks(setk_handle_t h, const float* spec, int num_channels, int num_frames,
                    int num_bins, int num_iters, const float* init_mask, float* gamma_out,
                    float* mask_out, int flags, void* stream);

/* The general form: num_classes K in [2, 4], 1 <= C <= 16 channels -- CgmmTrainer with
 * num_classes != 2 (cluster.py:427-434), and the K = 2 starts for arrays wider than 8.
 * gamma0[K][F][T] float64 (host or device) is the start: for K > 2 the reference's
 * np.random.uniform(size=[K, F, T]) / sum over K, drawn from the legacy global generator that
 * estimate_cgmm_masks.py:28 seeds with --seed (the caller draws it: setk_amd/libs/cluster.py);
 * NULL with K = 2 selects init_mask[T][F] or the deterministic start as in setk_cgmm_masks.
 * gamma_out[K][T][F] float32 receives every class's posteriors.  float64 throughout, like
 * the reference; a straightforward kernel (cgmm_k.hip), not the tuned K = 2 path. */
int setk_cgmm_masks_k(setk_handle_t h, const float* spec, int num_channels, int num_frames,
                      int num_bins, int num_classes, int num_iters, const double* gamma0,
                      const float* init_mask, float* gamma_out, int flags, void* stream);
/* The same, reporting per bin what the reference's np.linalg.eigh would have refused
 * (cluster.py:104-113, uncaught by estimate_cgmm_masks.py: its run ends with LinAlgError):
 * status[F] (host or device, may be NULL) receives the worst SETK_NUM_* over classes and
 * iterations -- SETK_NUM_NONFINITE for a covariance with NaN / inf, SETK_NUM_NOCONV when the
 * Jacobi sweep limit was reached.  The posteriors are written either way. */
int setk_cgmm_masks_k_status(setk_handle_t h, const float* spec, int num_channels, int num_frames,
                             int num_bins, int num_classes, int num_iters, const double* gamma0,
                             const float* init_mask, float* gamma_out, int flags, int* status,
                             void* stream);

/* Batched form: n_utts utterances per EM stage launch (device pointers only;
 * spec[u] = [C][num_frames[u]][spec_pitch] (0 = F), mask_out[u] = [num_frames[u]][F],
 * init_mask NULL or per-utterance NULL-able table).  Asynchronous on `stream`. */
int setk_cgmm_masks_batch(setk_handle_t h, int n_utts, int num_channels,
                          const float* const* spec, const int* num_frames, int num_bins,
                          int num_iters, const float* const* init_mask, float* const* mask_out,
                          int flags, int spec_pitch, void* stream);

/* The compute loop of estimate_cgmm_masks.py:44-64 for a batch, audio in, mask out:
 * STFT (n_fft = 512 plan) written straight into the bin-major layout of the
 * bin-resident EM, the EM itself, masks as float32 [T][F].  audio[u] device float32
 * [C][num_samples[u]]; init_mask / mask_out / flags as setk_cgmm_masks_batch.
 * SETK_ERR_UNSUPPORTED when one bin of the longest utterance does not fit a CU
 * (then: setk_stft_batch + setk_cgmm_masks_batch). */
int setk_cgmm_estimate_batch(setk_handle_t h, int n_utts, int num_channels,
                             const float* const* audio, const int* num_samples, int num_iters,
                             const float* const* init_mask, float* const* mask_out, int flags,
                             void* stream);

/* directional_feats (libs/spatial.py:184-208, compute_df_on_mask.py:40-54):
 * out[t][f] = mean over the n_pairs microphone pairs (i, j) = (pairs[2p],
 * pairs[2p+1], host array) of cos((arg X_i - arg X_j) - (arg v_i - arg v_j)),
 * spec [C][T][F] and steer_vector [F][C] complex64, out [T][F] float32. */
int setk_directional_feats(setk_handle_t h, const float* spec, const float* steer_vector,
                           const int* pairs, int n_pairs, int num_channels, int num_frames,
                           int num_bins, float* out, void* stream);

/* FixedBeamformer.run + inverse_stft for a batch (apply_fixed_beamformer.py:
 * 38-48, libs/beamformer.py:323-340): wave[u] = istft(sum_c conj(w[f][c]) X_c),
 * rescaled to max |audio[u]| (SpectrogramReader.maxabs).  weights: n_sets
 * filters in the reference layout [set][F = 257][C] complex64 (host or device);
 * weight_index[u] (host, may be NULL = set 0) picks the beam of utterance u.
 * audio / wave: device pointers as for setk_enhance_batch; flags:
 * SETK_FLAG_OUT_PCM16, SETK_FLAG_NO_RENORM (the wave as inverse_stft(norm=None)
 * leaves it: apply_classic_beamformer.py:109-110 without --normalize).  Needs the
 * n_fft = 512 plan. */
int setk_apply_weights_batch(setk_handle_t h, int n_utts, int num_channels,
                             const float* const* audio, const int* num_samples,
                             const float* weights, int n_sets, const int* weight_index,
                             void* const* wave, int flags, void* stream);

/* ---- WPE dereverberation --------------------------------------
 * wpe_step of scripts/sptk/libs/wpe.py:58-81 (tap-stacked correlation, solve,
 * filter; fp64 like the reference) iterated `num_iters` times as wpe() does
 * (:84-110): spec / out [C][T][F] complex64.  lambda of the first iteration is
 * compute_lambda(spec, context) (:32-55) or, when lambda_enh != NULL,
 * max(|lambda_enh[t][f]|^2, eps) -- the previous enhanced signal of
 * facted_wpd() (:146-149); later iterations use compute_lambda(dereverb).
 * inv_lambda_out (may be NULL) receives 1 / lambda of the last iteration as a
 * float32 [T][F] array: passed to setk_covar as the mask it yields the
 * power-weighted covariance of facted_wpd (:160-162, up to a per-bin scale that
 * cancels in the MVDR weight).  status[F] (may be NULL) receives SETK_NUM_*;
 * SETK_NUM_SINGULAR is numpy's LinAlgError, SETK_NUM_RANKDEF a note (columns at the noise
 * level were dropped; the result is finite).  Limits: channels <= 16, NK = channels * taps
 * <= 256.  While R fits LDS (NK^2 + NK N + 16 (NK + N) complex128 <= 160 KB: up to 8
 * channels x 10 taps, 16 x 4) a workgroup never leaves its CU; beyond that (8 x 12, 16 x 6,
 * 16 x 10 ...) R is factored in global scratch of the handle's arena (NK^2 complex128 per
 * bin and utterance in flight).  NK > 256: SETK_ERR_UNSUPPORTED names the bound. */
int setk_wpe(setk_handle_t h, const float* spec, int num_channels, int num_frames,
             int num_bins, int taps, int delay, int context, int num_iters,
             const float* lambda_enh, float* out, float* inv_lambda_out,
             int* status, void* stream);

/* One wpe_step (scripts/sptk/libs/wpe.py:58-81) with the caller's variances used AS
 * GIVEN: lambda_ft is float64 [F][T] (the reference's F x T), no floor, no
 * re-estimation.  spec / out [C][T][F] complex64; taps / delay describe the tap
 * matrix compute_tap_mat(reverb, taps, delay) (:13-29) that the kernel indexes in
 * place.  status as setk_wpe. */
int setk_wpe_step(setk_handle_t h, const float* spec, int num_channels, int num_frames,
                  int num_bins, int taps, int delay, const double* lambda_ft, float* out,
                  int* status, void* stream);

/* wpe() (libs/wpe.py:84-110) for n_utts utterances of the same channel count in one
 * call: every iteration is ONE launch over (bin, utterance) instead of 257 workgroups
 * per utterance.  spec[u] / out[u] [C][num_frames[u]][F] complex64 (host or device),
 * status [n_utts][F] or NULL.  Same limits as setk_wpe. */
int setk_wpe_batch(setk_handle_t h, int n_utts, const float* const* spec, int num_channels,
                   const int* num_frames, int num_bins, int taps, int delay, int context,
                   int num_iters, float* const* out, int* status, void* stream);

/* setk_wpe_batch for facted_wpd's batch (libs/wpe.py:113-177, apply_wpd.py:20-60): the
 * variances of iteration 0 come per utterance from lambda_enh[u] (complex64 [T_u][F], |.|^2 of
 * the previous enhanced signal; a NULL array or entry: compute_lambda of the input, as
 * setk_wpe_batch), and inv_lambda_out[u] (float32 [T_u][F], NULL array: not wanted) receives
 * 1 / lambda of the LAST iteration -- the per-utterance arguments of setk_wpe, for n_utts
 * utterances behind one launch per iteration. */
int setk_wpe_batch_var(setk_handle_t h, int n_utts, const float* const* spec, int num_channels,
                       const int* num_frames, int num_bins, int taps, int delay, int context,
                       int num_iters, const float* const* lambda_enh, float* const* out,
                       float* const* inv_lambda_out, int* status, void* stream);

/* The same with spec[u] / out[u] in the reference's own layout, F x N x T_u complex64 (the
 * `reverb` / return value of wpe(), libs/wpe.py:84-110): it is the layout the step kernel
 * works in, so no transposition happens on either side.  out[u] must not alias spec[u]. */
int setk_wpe_batch_fnt(setk_handle_t h, int n_utts, const float* const* spec, int num_channels,
                       const int* num_frames, int num_bins, int taps, int delay, int context,
                       int num_iters, float* const* out, int* status, void* stream);

/* ---- fused hot path ------------------------------------------------------
 * The compute body of apply_adaptive_beamformer.py:130-178 for a batch of
 * utterances that share the channel count, in four kernel stages:
 *   1. windowed rFFT + masked outer-product accumulation (never stores X)
 *   2. batched per-bin weight solve
 *   3. rFFT recompute + w^H x + irFFT + overlap-add
 *   4. max-abs renorm (to max|input|) and emit float32 (or PCM16)
 * audio[u]   device float32 [C][num_samples[u]]; with SETK_FLAG_IN_PCM16 device int16
 *            [C][setk_pcm16_channel_stride(num_samples[u])] (cast to const float*).
 *            Float samples of any magnitude are accepted: the matrix-core transforms of
 *            stage 3 bring an utterance whose max |x| exceeds 1 into their fp16 operand range
 *            by a power of two taken from stage 1's max |x| (exact, undone on the way out)
 * mask_s[u]  device float32 [T_u][F]
 * mask_n     NULL, or per-utterance interferer masks (--itf-mask)
 * wave[u]    device float32 (or int16) [hop*(T_u-1)] when center, see
 *            setk_istft_num_samples
 * status[u]  int, SETK_NUM_* (worst bin of the utterance); host memory (the call
 *            then synchronises the stream), device memory (asynchronous) or NULL
 * The pointer tables and num_samples are HOST arrays of n_utts entries.
 * Requires the n_fft = 512 plan and 1 <= C <= 8. */
int setk_enhance_batch(setk_handle_t h, const setk_bf_opts* opts, int n_utts,
                       int num_channels, const float* const* audio,
                       const int* num_samples, const float* const* mask_s,
                       const float* const* mask_n, void* const* wave,
                       int* status, void* stream);

/* The same call with taps on the intermediate results of the fused kernels
 * (parity tests compare them with compute_covar / the weight classes directly;
 * a caller may also keep the weights).  Every member may be NULL; device or
 * host memory:
 *   Rs, Rn   [n_utts][F][C][C] complex64  normalised covariances out of the fused
 *            STFT+covariance kernel and its reduction (libs/beamformer.py:87-103)
 *   weight   [n_utts][F][C]    complex64  beamformer weights (libs/beamformer.py
 *            weight() of the selected class, after BAN when requested)
 *   maxabs   [n_utts] float32  max |audio| (WaveReader.maxabs, the renorm target) */
typedef struct setk_batch_taps {
    float* Rs;
    float* Rn;
    float* weight;
    float* maxabs;
} setk_batch_taps;
int setk_enhance_batch_taps(setk_handle_t h, const setk_bf_opts* opts, int n_utts,
                            int num_channels, const float* const* audio,
                            const int* num_samples, const float* const* mask_s,
                            const float* const* mask_n, void* const* wave,
                            int* status, const setk_batch_taps* taps, void* stream);

/* ---- multi-GPU: the work-queue barrier on RCCL ------------------------------
 * The path shards by utterance with no data exchange (the reference: split_scp.pl + run.pl
 * JOB=1:nj, scripts/run_adapt_beamformer.sh:69-92).  One process per GPU needs a start / finish
 * barrier and the sums of its "Processed N utterances" counters: an all-reduce of a few
 * doubles over xGMI.  librccl is dlopen'ed on first use.  Rendezvous: rank 0 calls
 * setk_comm_unique_id, the caller carries the SETK_COMM_ID_BYTES to every rank (setk_amd/
 * dist.py: a TCP socket on MASTER_ADDR:MASTER_PORT), every rank calls setk_comm_create
 * (collective).  values: host array, reduced in place over all ranks (n <= 64).
 * SETK_ERR_UNSUPPORTED: librccl is not available (setk_comm_last_error has the reason). */
#define SETK_COMM_ID_BYTES 128
#define SETK_COMM_SUM 0
#define SETK_COMM_MAX 1
typedef struct setk_comm* setk_comm_t;
int setk_comm_unique_id(char out[SETK_COMM_ID_BYTES]);
int setk_comm_create(setk_comm_t* out, int device_ordinal, const char id[SETK_COMM_ID_BYTES],
                     int rank, int world);
int setk_comm_allreduce_f64(setk_comm_t c, double* values, int n, int op);
int setk_comm_barrier(setk_comm_t c);
int setk_comm_destroy(setk_comm_t c);
const char* setk_comm_last_error(void);

/* Stage timings (ms, hipEvent on `stream`) of the most recent
 * setk_enhance_batch when profiling was enabled with setk_set_profiling(h,1):
 * out[0] = the fused STFT+covariance kernel alone, out[1] = partial reduction +
 * weight solve, out[2] = beamform+iSTFT kernel, out[3] = renorm kernel.  */
int setk_set_profiling(setk_handle_t h, int enable);
int setk_last_stage_ms(setk_handle_t h, float out[4]);

#ifdef __cplusplus
}
#endif
#endif /* SETK_HIP_H_ */
