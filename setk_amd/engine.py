"""
Batched front end of the fused hot path (setk_enhance_batch): takes decoded
utterances (numpy), keeps them resident in HBM through torch tensors, runs the
four kernel stages for the whole batch and hands back PCM16 / float32 waves.

This is the compute body of apply_adaptive_beamformer.py:130-178 for many
utterances at once -- per-utterance work is microseconds on an MI355X, so the
engineering unit is the batch, not the utterance.
"""
import os
import threading

import numpy as np

from . import _ffi
from .libs.utils import nextpow2, stft_window, cmat_abs

BEAMFORMER_KINDS = {
    # name -> (kind, pmwf_beta)
    "mvdr": (_ffi.BF_MVDR, 0.0),
    "mpdr": (_ffi.BF_MPDR, 0.0),
    "mpdr-whiten": (_ffi.BF_MPDR_WHITEN, 0.0),
    "gevd": (_ffi.BF_GEVD, 0.0),
    "pmwf-0": (_ffi.BF_PMWF, 0.0),
    "pmwf-1": (_ffi.BF_PMWF, 1.0),
}
RANK1 = {"": _ffi.RANK1_NONE, "none": _ffi.RANK1_NONE, "eig": _ffi.RANK1_EIG,
         "gev": _ffi.RANK1_GEV}


class Pcm16Frames(object):
    """Interleaved 16-bit PCM frames [N, C] of one utterance, exactly as stored in
    its wav (WaveReader.read_pcm16).  BatchEnhancer uploads the 2-byte samples and
    converts / transposes them on the device (setk_pcm16_to_float)."""

    def __init__(self, frames):
        frames = np.asarray(frames)
        if frames.dtype != np.int16 or frames.ndim != 2:
            raise ValueError("Pcm16Frames expects an int16 array of shape N x C")
        # (a private, writable copy: np.frombuffer views of file bytes are read-only
        # and torch.from_numpy wants to own writable memory)
        self.frames = np.array(frames, dtype=np.int16, order="C", copy=True)

    @property
    def num_channels(self):
        return self.frames.shape[1]

    @property
    def size(self):
        return self.frames.size

    def to_float(self):
        """The reference's host view: C x N float32 (soundfile scaling)."""
        return np.ascontiguousarray(self.frames.T.astype(np.float32) / np.float32(32768.0))


def _channels_and_size(samps):
    if isinstance(samps, Pcm16Frames):
        return samps.num_channels, samps.size
    samps = np.asarray(samps)
    return (1 if samps.ndim == 1 else samps.shape[0]), samps.size


def compute_vad_masks(spectrogram, proportion):
    """Energy based VAD mask of apply_adaptive_beamformer.py:50-71: keep
    proportion*100 % of the energy.  spectrogram F x T -> (T x F bool, index).
    The cumulative sum replaces the reference's python while-loop."""
    energy = cmat_abs(spectrogram)
    vec = np.sort(energy.flatten())
    filter_energy = np.sum(vec) * (1 - proportion)
    csum = np.cumsum(vec)
    index = int(np.searchsorted(csum, filter_energy, side="right"))
    threshold = vec[min(index, vec.shape[0] - 1)] if vec.shape[0] else 0
    return (energy < threshold).transpose(), index


class _Slabs(object):
    """Grow-only working set of the engines' torch-free batch paths: a page-locked input slab
    and its device twin, a device scratch for converted samples, a device output slab and its
    page-locked twin, one stream -- all from the library (setk_host_alloc / setk_device_alloc
    / setk_stream_create)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.cap = [0, 0, 0]
        self.h_in = self.d_in = self.d_f32 = self.h_out = self.d_out = 0
        self.np_in = self.np_out = None
        self.stream = ctx.stream_create()

    def reserve(self, n_in, n_f32, n_out):
        ctx = self.ctx
        ctx.stream_synchronize(self.stream)  # the previous call's work has left the slabs
        if n_in > self.cap[0]:
            self.np_in = None
            if self.h_in:
                ctx.host_free(self.h_in)
                ctx.device_free(self.d_in)
            n = int(n_in * 1.25)
            self.h_in, self.np_in = ctx.host_alloc(n)
            self.d_in = ctx.device_alloc(n)
            self.cap[0] = n
        if n_f32 > self.cap[1]:
            if self.d_f32:
                ctx.device_free(self.d_f32)
            n = int(n_f32 * 1.25)
            self.d_f32 = ctx.device_alloc(n)
            self.cap[1] = n
        if n_out > self.cap[2]:
            self.np_out = None
            if self.h_out:
                ctx.host_free(self.h_out)
                ctx.device_free(self.d_out)
            n = int(n_out * 1.25)
            self.h_out, self.np_out = ctx.host_alloc(n)
            self.d_out = ctx.device_alloc(n)
            self.cap[2] = n

    def stage_audio(self, utts, C, extra_out):
        """Lay the utterances (C x N float32 arrays or Pcm16Frames) out in the input slab,
        copy it up in one piece and convert the 16-bit ones on the device.  extra_out(N) ->
        output bytes of an utterance.  Returns (device sample pointers, lengths, output
        offsets, output bytes)."""
        ctx = self.ctx
        al = lambda v: (v + 255) & ~255  # noqa: E731
        lay, n_in, n_f32, n_out = [], 0, 0, 0
        for s in utts:
            pcm = isinstance(s, Pcm16Frames)
            N = s.frames.shape[0] if pcm else _channels_and_size(s)[1] // C
            lay.append((pcm, N, n_in, n_f32, n_out))
            n_in = al(n_in + (2 if pcm else 4) * C * N)
            if pcm:
                n_f32 = al(n_f32 + 4 * C * N)
            n_out = al(n_out + extra_out(N))
        self.reserve(max(n_in, 256), max(n_f32, 256), max(n_out, 256))
        aptr, ns, pcm_jobs = [], [], []
        for s, (pcm, N, o_in, o_f32, _) in zip(utts, lay):
            if pcm:
                self.np_in[o_in:o_in + 2 * C * N] = np.frombuffer(s.frames, dtype=np.uint8)
                pcm_jobs.append((self.d_in + o_in, N, self.d_f32 + o_f32))
                aptr.append(self.d_f32 + o_f32)
            else:
                a = np.ascontiguousarray(s, dtype=np.float32)
                self.np_in[o_in:o_in + a.nbytes] = np.frombuffer(a, dtype=np.uint8)
                aptr.append(self.d_in + o_in)
            ns.append(N)
        ctx.memcpy_h2d_async(self.d_in, self.h_in, n_in, self.stream)
        if pcm_jobs:
            ctx.pcm16_to_float_batch(C, [p for p, _, _ in pcm_jobs], [n for _, n, _ in pcm_jobs],
                                     [o for _, _, o in pcm_jobs], stream=self.stream)
        return aptr, ns, [l[4] for l in lay], n_out

    def fetch(self, n_out):
        """One copy down; returns the page-locked view (valid until the next reserve())."""
        self.ctx.memcpy_d2h_async(self.h_out, self.d_out, n_out, self.stream)
        self.ctx.stream_synchronize(self.stream)
        return self.np_out

    def close(self):
        ctx = self.ctx
        if not self.stream:
            return
        ctx.stream_synchronize(self.stream)
        self.np_in = self.np_out = None
        for p in (self.h_in, self.h_out):
            if p:
                ctx.host_free(p)
        for p in (self.d_in, self.d_f32, self.d_out):
            if p:
                ctx.device_free(p)
        ctx.stream_destroy(self.stream)
        self.stream = 0
        self.h_in = self.d_in = self.d_f32 = self.h_out = self.d_out = 0


class BatchEnhancer(object):
    def __init__(self, beamformer="mvdr", frame_len=512, frame_hop=256, center=True,
                 round_power_of_two=True, window="hann", ban=False, pmwf_ref=-1, rank1_appro="",
                 post_mask=False, vad_proportion=1, pcm16=False, device=None, ctx=None,
                 max_batch_samples=1 << 28, strict_reference=False):
        """strict_reference: refuse (status SETK_NUM_SINGULAR -> the CLI's LinAlgError branch)
        exactly where the reference's numpy.linalg.solve meets an exactly zero pivot
        (SETK_FLAG_STRICT_REFERENCE, include/setk_hip.h); default: regularise and go through."""
        if beamformer not in BEAMFORMER_KINDS:
            raise ValueError(f"unknown beamformer {beamformer}")
        # no GPU / no library: setk_create fails here, loudly (there is no CPU fallback).
        # torch is the plumbing of enhance() only -- the streaming pipeline brings its own
        # buffers and streams -- and is imported when enhance() first needs it.
        self.ctx = ctx or _ffi.default_context(device)
        self._torch = None
        n_fft = nextpow2(frame_len) if round_power_of_two else frame_len
        self.stft = dict(frame_len=frame_len, frame_hop=frame_hop, n_fft=n_fft, center=center,
                         window=stft_window(window, frame_len))
        self.num_bins = n_fft // 2 + 1
        kind, beta = BEAMFORMER_KINDS[beamformer]
        flags = 0
        if ban:
            flags |= _ffi.FLAG_BAN
        if post_mask:
            flags |= _ffi.FLAG_POST_MASK
        if pcm16:
            flags |= _ffi.FLAG_OUT_PCM16
        if strict_reference:
            flags |= _ffi.FLAG_STRICT_REFERENCE
        self.base_flags = flags
        self.opts_kw = dict(kind=kind, pmwf_beta=beta, pmwf_ref=int(pmwf_ref),
                            rank1=RANK1[rank1_appro])
        self.pcm16 = pcm16
        # wave files' 16-bit samples go into the fused kernels as stored (de-interleaved, never
        # widened to float32) when the geometry is the matrix-core pass 2's: hop = n_fft / 2
        # (the two library switches are read the way csrc/capi.hip reads them -- atoi -- so that
        #  e.g. SETK_MC_PASS2=false disables the form on both sides of the ABI)
        self.pcm_direct_ok = n_fft == 512 and 2 * frame_hop == n_fft and \
            _ffi.env_atoi("SETK_PCM16_DIRECT", 1) != 0 and \
            _ffi.env_atoi("SETK_MC_PASS2", 1) != 0 and _ffi.env_atoi("SETK_LEGACY_FFT", 0) == 0
        self.vad_proportion = vad_proportion
        self.max_batch_samples = max_batch_samples

    @property
    def torch(self):
        if self._torch is None:
            torch = _ffi.import_torch(type(self).__name__ + ': this input (more than 8 channels / an unfused geometry)')
            if not torch.cuda.is_available():
                raise _ffi.SetkError("BatchEnhancer needs an MI355X (no CPU fallback)")
            self._torch = torch
        return self._torch

    @property
    def dev(self):
        return self.torch.device("cuda", self.ctx.device)

    def _plan(self):
        s = self.stft
        self.ctx.stft_plan(s["frame_len"], s["frame_hop"], s["n_fft"], s["center"], s["window"])

    def frames_and_length(self, num_samples):
        """(T, L) of the planned transform for a signal of num_samples, evaluated on
        the host (librosa's framing: SURVEY appendix A)."""
        s = self.stft
        hop, n_fft = s["frame_hop"], s["n_fft"]
        if s["center"]:
            if num_samples < n_fft // 2 + 1:
                raise ValueError("signal shorter than n_fft/2+1 (reflect padding)")
            T = 1 + num_samples // hop
            return T, hop * (T - 1)
        if num_samples < n_fft:
            raise ValueError("signal shorter than n_fft")
        T = 1 + (num_samples - n_fft) // hop
        return T, n_fft + hop * (T - 1)

    def condition_mask(self, mask, num_frames):
        """apply_adaptive_beamformer.py:146-151: masks arrive T x F or F x T."""
        F = self.num_bins
        mask = np.asarray(mask)
        if mask.ndim != 2:
            raise ValueError(f"mask must be 2D, got {mask.shape}")
        if mask.shape[0] == F and mask.shape[1] != F:
            mask = np.transpose(mask)
        if mask.shape[1] != F:
            raise ValueError("Input mask matrix should be shape as " +
                             f"[num_frames x num_bins], now is {mask.shape}")
        if mask.shape[0] != num_frames:
            raise ValueError("Shape of input obs do not match with mask matrix, " +
                             f"{num_frames} frames vs {mask.shape}")
        return mask

    def enhance(self, utts):
        """utts: list of (samps C x N float32 | Pcm16Frames, speech mask, interferer mask|None).
        Returns list of (wave ndarray | None, status) in input order; status != 0
        is the reference's LinAlgError case (the utterance is to be skipped)."""
        self._plan()
        results = [None] * len(utts)
        groups = {}
        for i, (samps, _, itf) in enumerate(utts):
            groups.setdefault((_channels_and_size(samps)[0], itf is not None), []).append(i)
        for (C, has_itf), idx in groups.items():
            batch, nsamp = [], 0
            for i in idx:
                n = _channels_and_size(utts[i][0])[1]
                if batch and nsamp + n > self.max_batch_samples:
                    self._run(utts, batch, C, has_itf, results)
                    batch, nsamp = [], 0
                batch.append(i)
                nsamp += n
            if batch:
                self._run(utts, batch, C, has_itf, results)
        return results

    def _run_unfused(self, utts, batch, C, has_itf, results):
        """n_fft != 512 or more than 8 channels: the same stages through the
        stand-alone operators (setk_stft -> setk_covar x2 -> setk_weights ->
        setk_beamform -> setk_istft), everything resident on the device, one
        utterance at a time."""
        torch, ctx, dev, F = self.torch, self.ctx, self.dev, self.num_bins
        mpdr = self.opts_kw["kind"] in (_ffi.BF_MPDR, _ffi.BF_MPDR_WHITEN)
        for i in batch:
            samps, mask, itf = utts[i]
            if isinstance(samps, Pcm16Frames):
                samps = samps.to_float()
            samps = np.ascontiguousarray(samps, dtype=np.float32)
            if samps.ndim == 1:
                samps = samps[None]
            N = samps.shape[1]
            T = ctx.num_frames(N)
            mask = self.condition_mask(mask, T)
            if has_itf:
                itf = self.condition_mask(itf, T)
            else:
                mask = np.minimum(mask, 1)
            a = torch.from_numpy(samps).to(dev)
            spec = torch.empty((C, T, F), dtype=torch.complex64, device=dev)
            ctx.stft(a, spec)
            if 0.5 < self.vad_proportion < 1:
                vad, _ = compute_vad_masks(spec[0].cpu().numpy().T, self.vad_proportion)
                mask = np.where(vad, 1.0e-4, mask)
                if has_itf:
                    itf = np.where(vad, 1.0e-4, itf)
            ms = torch.from_numpy(np.ascontiguousarray(mask, dtype=np.float32)).to(dev)
            mn = torch.from_numpy(np.ascontiguousarray(itf, dtype=np.float32)).to(dev) \
                if has_itf else (1 - ms).contiguous()
            Rs = torch.empty((F, C, C), dtype=torch.complex64, device=dev)
            Rn = torch.empty_like(Rs)
            ctx.covar(spec, mn, C, T, F, Rn)
            ctx.covar(spec, ms, C, T, F, Rs)
            Ry = None
            if mpdr:
                Ry = torch.empty_like(Rs)
                ctx.covar(spec, torch.ones_like(ms), C, T, F, Ry)
            w = torch.empty((F, C), dtype=torch.complex64, device=dev)
            status = np.zeros(F, dtype=np.int32)
            flags = self.base_flags & (_ffi.FLAG_BAN | _ffi.FLAG_STRICT_REFERENCE)
            ctx.weights(_ffi.BfOpts(flags=flags, **self.opts_kw), Rs, Rn, Ry, F, C, w, status)
            if status.any():
                results[i] = (None, int(status.max()))
                continue
            enh = torch.empty((T, F), dtype=torch.complex64, device=dev)
            ctx.beamform(w, spec, C, T, F, enh)
            if self.base_flags & _ffi.FLAG_POST_MASK:
                enh = (enh * ms).contiguous()
            L = ctx.istft_num_samples(T)
            wave = torch.empty((1, L), dtype=torch.float32, device=dev)
            norm = a.abs().max().reshape(1).contiguous()
            ctx.istft(enh.reshape(1, T, F), 1, T, None, norm, wave)
            out = wave[0]
            if self.pcm16:
                out = torch.round(out * 32767.0).to(torch.int16)
            results[i] = (out.cpu().numpy(), 0)

    def _run(self, utts, batch, C, has_itf, results):
        if self.stft["n_fft"] != 512 or C > 8:
            # the fused kernels are specialised for n_fft = 512 and C <= 8
            return self._run_unfused(utts, batch, C, has_itf, results)
        kind = self.opts_kw["kind"]
        if has_itf and kind == _ffi.BF_MPDR_WHITEN:
            # the fused kernel forms Ry from mask_s + (1 - mask_s); with a separate
            # interferer mask Rn and Ry are independent (libs/beamformer.py:573-590)
            return self._run_unfused(utts, batch, C, has_itf, results)
        drop_itf = has_itf and kind == _ffi.BF_MPDR  # plain MPDR never reads mask_n
        torch, ctx, dev = self.torch, self.ctx, self.dev
        audio, masks, itfs, waves, ns = [], [], [], [], []
        flags = self.base_flags | (0 if has_itf else _ffi.FLAG_CLAMP_MASK)
        # 16-bit PCM all the way into the kernels (SETK_FLAG_IN_PCM16): no float32 copy exists
        direct = self.pcm_direct_ok and all(isinstance(utts[i][0], Pcm16Frames) for i in batch) \
            and not (0.5 < self.vad_proportion < 1)
        if direct:
            flags |= _ffi.FLAG_IN_PCM16
        staged = []  # (interleaved frames on the device, N, planar destination)
        for i in batch:
            samps, mask, itf = utts[i]
            if isinstance(samps, Pcm16Frames) and direct:
                pcm = torch.from_numpy(samps.frames).to(dev)
                N = samps.frames.shape[0]
                a = torch.empty((C, ctx.pcm16_channel_stride(N)), dtype=torch.int16, device=dev)
                staged.append((pcm, N, a))
            elif isinstance(samps, Pcm16Frames):
                # the wav's 2-byte frames go up as they are; scaling and the
                # transpose to C x N happen on the device
                pcm = torch.from_numpy(samps.frames).to(dev)
                N = samps.frames.shape[0]
                a = torch.empty((C, N), dtype=torch.float32, device=dev)
                ctx.pcm16_to_float(pcm, C, N, a)
            else:
                samps = np.ascontiguousarray(samps, dtype=np.float32)
                if samps.ndim == 1:
                    samps = samps[None]
                N = samps.shape[1]
                a = torch.from_numpy(samps).to(dev)
            T = ctx.num_frames(N)
            mask = self.condition_mask(mask, T)
            if has_itf:
                itf = self.condition_mask(itf, T)
            if 0.5 < self.vad_proportion < 1:
                spec0 = torch.empty((1, T, self.num_bins), dtype=torch.complex64, device=dev)
                ctx.stft(a[:1], spec0)
                vad, _ = compute_vad_masks(spec0[0].cpu().numpy().T, self.vad_proportion)
                if not has_itf:
                    mask = np.minimum(mask, 1)
                mask = np.where(vad, 1.0e-4, mask)
                if has_itf:
                    itf = np.where(vad, 1.0e-4, itf)
            audio.append(a)
            masks.append(torch.from_numpy(np.ascontiguousarray(mask, dtype=np.float32)).to(dev))
            if has_itf and not drop_itf:
                itfs.append(torch.from_numpy(np.ascontiguousarray(itf, dtype=np.float32)).to(dev))
            L = ctx.istft_num_samples(T)
            waves.append(torch.empty(L, dtype=torch.int16 if self.pcm16 else torch.float32,
                                     device=dev))
            ns.append(N)
        if staged:
            ctx.pcm16_deinterleave_batch(C, [p.data_ptr() for p, _, _ in staged], [n for _, n, _ in staged],
                                         [a.data_ptr() for _, _, a in staged])
        opts = _ffi.BfOpts(flags=flags, **self.opts_kw)
        status = ctx.enhance_batch(opts, C, [t.data_ptr() for t in audio], ns,
                                   [t.data_ptr() for t in masks],
                                   [t.data_ptr() for t in itfs] if itfs else None,
                                   [t.data_ptr() for t in waves], want_status=True)
        for j, i in enumerate(batch):
            results[i] = (waves[j].cpu().numpy() if status[j] == 0 else None, status[j])


class FixedBatchBeamformer(object):
    """apply_fixed_beamformer.py:38-48 for a batch: FixedBeamformer.run +
    inverse_stft with the renorm to max |audio|.  weights: B x F x M complex (the
    reference's layout); run() takes [(samps C x N float32 | Pcm16Frames, beam)]
    and returns the waveforms (int16 when pcm16 else float32) in input order."""

    def __init__(self, weights, frame_len=512, frame_hop=256, center=True,
                 round_power_of_two=True, window="hann", pcm16=False, device=None,
                 max_batch_samples=1 << 29, renorm=True):
        # no GPU / no library: setk_create fails here, loudly.  The fused batch path brings
        # its own buffers and stream; torch is the plumbing of the stand-alone operators only
        # (n_fft != 512, more than 8 channels) and is imported when they are first needed.
        self.ctx = _ffi.default_context(device)
        self._torch = None
        self._slabs = None
        self._dw = 0
        # renorm=False: inverse_stft(norm=None), apply_classic_beamformer.py:109-110
        self.renorm = bool(renorm)
        weights = np.asarray(weights)
        if weights.ndim == 2:
            weights = weights[None]
        self.weights = np.ascontiguousarray(weights, dtype=np.complex64)  # B x F x M
        n_fft = nextpow2(frame_len) if round_power_of_two else frame_len
        self.stft = dict(frame_len=frame_len, frame_hop=frame_hop, n_fft=n_fft, center=center,
                         window=stft_window(window, frame_len))
        self.n_fft = n_fft
        if self.weights.shape[1] != n_fft // 2 + 1:
            raise ValueError(f"weights have {self.weights.shape[1]} bins, the transform "
                             f"{n_fft // 2 + 1}")
        self.pcm16 = pcm16
        self.max_batch_samples = max_batch_samples

    def set_weights(self, weights):
        """Swap the weight table (B x F x M) and keep everything else -- the pinned slabs, the
        device twin, the stream: what a caller does whose table grows from batch to batch
        (apply_classic_beamformer: one entry per DoA seen so far)."""
        weights = np.asarray(weights)
        if weights.ndim == 2:
            weights = weights[None]
        weights = np.ascontiguousarray(weights, dtype=np.complex64)
        if weights.shape[1:] != self.weights.shape[1:]:
            raise ValueError(f"weights {weights.shape[1:]}, engine built for {self.weights.shape[1:]}")
        if self._dw:
            if self._slabs is not None:
                self.ctx.stream_synchronize(self._slabs.stream)  # the old table may still be read
            self.ctx.device_free(self._dw)
            self._dw = 0
        self.weights = weights

    @property
    def torch(self):
        if self._torch is None:
            torch = _ffi.import_torch(type(self).__name__ + ': this input (more than 8 channels / an unfused geometry)')
            if not torch.cuda.is_available():
                raise _ffi.SetkError("setk_amd needs an MI355X GPU (no CPU fallback)")
            self._torch = torch
        return self._torch

    @property
    def dev(self):
        return self.torch.device("cuda", self.ctx.device)

    def _plan(self):
        s = self.stft
        self.ctx.stft_plan(s["frame_len"], s["frame_hop"], s["n_fft"], s["center"], s["window"])

    def close(self):
        """Give the slabs and the device copy of the weights back."""
        b, self._slabs = self._slabs, None
        if b:
            b.close()
        if self._dw:
            self.ctx.device_free(self._dw)
            self._dw = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, utts):
        self._plan()
        results = [None] * len(utts)
        groups = {}
        for i, (samps, _) in enumerate(utts):
            groups.setdefault(_channels_and_size(samps)[0], []).append(i)
        for C, idx in groups.items():
            if C != self.weights.shape[2]:
                raise ValueError(f"Input obs do not match with weight, {self.weights.shape[1:]} "
                                 f"vs {C} channels")
            batch, nsamp = [], 0
            for i in idx:
                n = _channels_and_size(utts[i][0])[1]
                if batch and nsamp + n > self.max_batch_samples:
                    self._run(utts, batch, C, results)
                    batch, nsamp = [], 0
                batch.append(i)
                nsamp += n
            if batch:
                self._run(utts, batch, C, results)
        return results

    def _upload(self, samps, C):
        torch, dev = self.torch, self.dev
        if isinstance(samps, Pcm16Frames):
            pcm = torch.from_numpy(samps.frames).to(dev)
            N = samps.frames.shape[0]
            a = torch.empty((C, N), dtype=torch.float32, device=dev)
            self.ctx.pcm16_to_float(pcm, C, N, a)
            return a, N
        samps = np.ascontiguousarray(samps, dtype=np.float32)
        if samps.ndim == 1:
            samps = samps[None]
        return torch.from_numpy(samps).to(dev), samps.shape[1]

    def _run(self, utts, batch, C, results):
        ctx = self.ctx
        if self.n_fft != 512 or C > 8:
            return self._run_unfused(utts, batch, C, results)
        # one slab up, setk_apply_weights_batch, one slab down -- on the library's own
        # buffers and stream
        if self._slabs is None:
            self._slabs = _Slabs(ctx)
        b = self._slabs
        esz = 2 if self.pcm16 else 4
        aptr, ns, off_out, n_out = b.stage_audio(
            [utts[i][0] for i in batch], C, lambda N: esz * ctx.istft_num_samples(ctx.num_frames(N)))
        if not self._dw:
            self._dw = ctx.device_alloc(self.weights.nbytes)
            ctx.memcpy_h2d_async(self._dw, self.weights.ctypes.data, self.weights.nbytes, b.stream)
        ctx.apply_weights_batch(C, aptr, ns, self._dw, self.weights.shape[0],
                                [int(utts[i][1]) for i in batch], [b.d_out + o for o in off_out],
                                flags=(_ffi.FLAG_OUT_PCM16 if self.pcm16 else 0) |
                                (0 if self.renorm else _ffi.FLAG_NO_RENORM), stream=b.stream)
        host = b.fetch(n_out)
        for k, i in enumerate(batch):
            L = ctx.istft_num_samples(ctx.num_frames(ns[k]))
            results[i] = np.frombuffer(host[off_out[k]:off_out[k] + esz * L],
                                       dtype=np.int16 if self.pcm16 else np.float32).copy()

    def _run_unfused(self, utts, batch, C, results):
        """n_fft != 512: setk_stft -> setk_beamform -> setk_istft per utterance."""
        torch, ctx, dev = self.torch, self.ctx, self.dev
        F = self.n_fft // 2 + 1
        for i in batch:
            samps, beam = utts[i]
            a, N = self._upload(samps, C)
            T = ctx.num_frames(N)
            spec = torch.empty((C, T, F), dtype=torch.complex64, device=dev)
            ctx.stft(a, spec)
            w = torch.from_numpy(self.weights[int(beam)]).to(dev)
            enh = torch.empty((T, F), dtype=torch.complex64, device=dev)
            ctx.beamform(w, spec, C, T, F, enh)
            L = ctx.istft_num_samples(T)
            wave = torch.empty((1, L), dtype=torch.float32, device=dev)
            norm = a.abs().max().reshape(1).contiguous() if self.renorm else None
            ctx.istft(enh.reshape(1, T, F), 1, T, None, norm, wave)
            out = wave[0]
            if self.pcm16:
                out = torch.round(out * 32767.0).to(torch.int16)
            results[i] = out.cpu().numpy()


class CgmmEstimator(object):
    """Batched blind mask estimation (estimate_cgmm_masks.py:19-71, K = 2): STFT
    and all EM iterations on the device, n utterances per kernel launch
    (setk_cgmm_masks_batch).  n_fft must be 512 for the device STFT used here."""

    def __init__(self, frame_len=512, frame_hop=256, center=True, round_power_of_two=True,
                 window="hann", num_iters=20, device=None, ctx=None, update_alpha=False):
        self.update_alpha = bool(update_alpha)
        # no GPU / no library: setk_create fails here, loudly.  torch is the plumbing of
        # estimate_device() (tensors in, tensors out); estimate() brings its own buffers.
        self.ctx = ctx or _ffi.default_context(device)
        self._torch = None
        self._bufs = None
        n_fft = nextpow2(frame_len) if round_power_of_two else frame_len
        self.stft = dict(frame_len=frame_len, frame_hop=frame_hop, n_fft=n_fft, center=center,
                         window=stft_window(window, frame_len))
        self.num_bins = n_fft // 2 + 1
        self.num_iters = num_iters
        import os
        self.force_streaming = os.environ.get("SETK_CGMM_STREAMING", "") not in ("", "0")

    @property
    def torch(self):
        if self._torch is None:
            torch = _ffi.import_torch(type(self).__name__ + ': this input (more than 8 channels / an unfused geometry)')
            if not torch.cuda.is_available():
                raise _ffi.SetkError("CgmmEstimator needs an MI355X (no CPU fallback)")
            self._torch = torch
        return self._torch

    @property
    def dev(self):
        return self.torch.device("cuda", self.ctx.device)

    def _plan(self):
        s = self.stft
        self.ctx.stft_plan(s["frame_len"], s["frame_hop"], s["n_fft"], s["center"], s["window"])

    def estimate_device(self, audio, init_masks=None):
        """audio: list of device float32 tensors C x N (same C).  Returns the
        list of device speech masks T x F (float32)."""
        torch, ctx, dev, F = self.torch, self.ctx, self.dev, self.num_bins
        self._plan()
        C = audio[0].shape[0]
        if self.stft["n_fft"] == 512 and C <= 8 and not self.force_streaming:
            # audio -> masks in one call: the spectrograms are written in the layout the
            # bin-resident EM reads (no [C][T][F] intermediate, no transpose pass)
            masks = [torch.empty((ctx.num_frames(a.shape[1]), F), dtype=torch.float32, device=dev)
                     for a in audio]
            init = None
            if init_masks is not None:
                init = [0 if m is None else m.data_ptr() for m in init_masks]
            try:
                ctx.cgmm_estimate_batch(C, [a.data_ptr() for a in audio],
                                        [a.shape[1] for a in audio], self.num_iters, init,
                                        [t.data_ptr() for t in masks],
                                        update_alpha=self.update_alpha)
                # no host synchronisation: the scratch lives in the handle's arena, whose
                # reuse by the next call is ordered on the stream
                return masks
            except _ffi.SetkUnsupported:
                pass  # a bin of the longest utterance does not fit a CU: streaming kernels
        if C > 8:
            # 9 - 16 channels: the general float64 EM (setk_cgmm_masks_k), one utterance at a time
            # on spectrograms of the stand-alone transform (any n_fft the plan accepts)
            masks = []
            for k, a in enumerate(audio):
                T = ctx.num_frames(a.shape[1])
                spec = torch.empty((C, T, F), dtype=torch.complex64, device=dev)
                ctx.stft(a, spec)
                gamma = torch.empty((2, T, F), dtype=torch.float32, device=dev)
                init = None if init_masks is None else init_masks[k]
                ctx.cgmm_masks_k(spec, C, T, F, 2, self.num_iters, None, init, gamma,
                                 update_alpha=self.update_alpha)
                masks.append(gamma[0])
            torch.cuda.current_stream().synchronize()
            return masks
        specs, masks, frames = [], [], []
        # rows padded to 128 bytes: the EM kernels stream 32-bin (256-byte) segments per
        # wavefront and a 2056-byte row pitch makes every segment straddle an extra line
        Fp = (F + 15) // 16 * 16
        for a in audio:
            T = ctx.num_frames(a.shape[1])
            specs.append(torch.empty((C, T, Fp), dtype=torch.complex64, device=dev))
            masks.append(torch.empty((T, F), dtype=torch.float32, device=dev))
            frames.append(T)
        # all spectrograms in one launch
        ctx.stft_batch(C, [a.data_ptr() for a in audio], [a.shape[1] for a in audio],
                       [t.data_ptr() for t in specs], spec_pitch=Fp)
        init = None
        if init_masks is not None:
            init = [0 if m is None else m.data_ptr() for m in init_masks]
        ctx.cgmm_masks_batch(C, [t.data_ptr() for t in specs], frames, F, self.num_iters, init,
                             [t.data_ptr() for t in masks], update_alpha=self.update_alpha,
                             spec_pitch=Fp)
        torch.cuda.current_stream().synchronize()  # specs must outlive the launches
        return masks

    def estimate(self, utts):
        """utts: list of C x N float32 numpy arrays or Pcm16Frames (16-bit frames as stored:
        converted on the device) -> list of T x F float32 masks.  Per channel count: the
        samples go up in ONE copy out of a page-locked slab, the masks come down in one; the
        buffers, the stream and the copies are the library's (no torch in this path).  Shapes
        the one-call estimator does not take (n_fft != 512, more than 8 channels, a bin that
        does not fit a CU) go through estimate_device()."""
        out = [None] * len(utts)
        groups = {}
        for i, s in enumerate(utts):
            groups.setdefault(_channels_and_size(s)[0], []).append(i)
        for C, idx in groups.items():
            if self.stft["n_fft"] != 512 or C > 8 or self.force_streaming:
                self._estimate_torch(utts, C, idx, out)
                continue
            try:
                self._estimate_native(utts, C, idx, out)
            except _ffi.SetkUnsupported:
                self._estimate_torch(utts, C, idx, out)
        return out

    def _estimate_torch(self, utts, C, idx, out):
        torch, ctx, dev = self.torch, self.ctx, self.dev
        audio, pcm = [], []
        for i in idx:
            s = utts[i]
            if isinstance(s, Pcm16Frames):
                a = torch.empty((C, s.frames.shape[0]), dtype=torch.float32, device=dev)
                pcm.append((torch.from_numpy(s.frames).to(dev), a))
            else:
                s = np.ascontiguousarray(s, dtype=np.float32)
                a = torch.from_numpy(s[None] if s.ndim == 1 else s).to(dev)
            audio.append(a)
        if pcm:
            ctx.pcm16_to_float_batch(C, [p.data_ptr() for p, _ in pcm], [a.shape[1] for _, a in pcm],
                                     [a.data_ptr() for _, a in pcm])
        masks = self.estimate_device(audio)
        host = torch.cat([m.reshape(-1) for m in masks]).cpu().numpy()
        off = 0
        for i, m in zip(idx, masks):
            out[i] = host[off:off + m.numel()].reshape(m.shape)
            off += m.numel()

    def close(self):
        """Give the slabs of estimate() back (also done when the estimator is collected)."""
        b, self._bufs = self._bufs, None
        if b:
            b.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _estimate_native(self, utts, C, idx, out):
        ctx, F = self.ctx, self.num_bins
        self._plan()
        if self._bufs is None:
            self._bufs = _Slabs(ctx)
        b = self._bufs
        aptr, ns, off_out, n_out = b.stage_audio([utts[i] for i in idx], C,
                                                 lambda N: 4 * ctx.num_frames(N) * F)
        ctx.cgmm_estimate_batch(C, aptr, ns, self.num_iters, None, [b.d_out + o for o in off_out],
                                stream=b.stream, update_alpha=self.update_alpha)
        host = b.fetch(n_out)
        for k, i in enumerate(idx):
            T = ctx.num_frames(ns[k])
            out[i] = np.frombuffer(host[off_out[k]:off_out[k] + 4 * T * F],
                                   dtype=np.float32).reshape(T, F).copy()


class BatchDereverb(object):
    """apply_wpe.py:30-66 for a batch, resident on the device: the STFT of every channel,
    num_iters WPE steps over every (bin, utterance) per launch (setk_wpe_batch, fp64) and
    the inverse STFT of every channel (inverse_stft with norm = None), one upload of the
    samples and one download of the waveforms per batch.  run() takes a list of C x N
    float32 arrays or Pcm16Frames with the same channel count and returns C x L float32
    arrays (pcm16: L x C int16 frames, ready for the wav writer), None where the tap
    correlation of a bin is singular (the reference's LinAlgError, apply_wpe.py:55-57)."""

    def __init__(self, taps=10, delay=3, context=1, num_iters=3, frame_len=512, frame_hop=256,
                 center=True, round_power_of_two=True, window="hann", device=None, pcm16=False):
        # no GPU / no library: setk_create fails here, loudly.  The n_fft = 512 path brings its
        # own buffers and stream; torch is the plumbing of the other transform sizes only.
        self.ctx = _ffi.default_context(device)
        self._torch = None
        self._slabs = None
        self._scratch, self._scratch_cap = 0, 0
        # pcm16: hand back interleaved int16 frames L x C, quantised on the device by the
        # writer's rule (wavio.float_to_pcm16: rint(x * 32767) in float64, wrapping)
        self.pcm16 = bool(pcm16)
        self.taps, self.delay, self.context, self.num_iters = taps, delay, context, num_iters
        self.rank_deficient_bins = 0  # SETK_NUM_RANKDEF notes seen so far (apply_wpe logs them)
        n_fft = nextpow2(frame_len) if round_power_of_two else frame_len
        self.stft = dict(frame_len=frame_len, frame_hop=frame_hop, n_fft=n_fft, center=center,
                         window=stft_window(window, frame_len))
        self.num_bins = n_fft // 2 + 1

    @property
    def torch(self):
        if self._torch is None:
            torch = _ffi.import_torch(type(self).__name__ + ': this input (more than 8 channels / an unfused geometry)')
            if not torch.cuda.is_available():
                raise _ffi.SetkError("setk_amd needs an MI355X GPU (no CPU fallback)")
            self._torch = torch
        return self._torch

    @property
    def dev(self):
        return self.torch.device("cuda", self.ctx.device)

    def close(self):
        """Give the slabs and the spectrogram scratch of run() back."""
        b, self._slabs = self._slabs, None
        if b:
            b.close()
        if self._scratch:
            self.ctx.device_free(self._scratch)
            self._scratch, self._scratch_cap = 0, 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, utts):
        ctx = self.ctx
        if not len(utts):
            return []
        s = self.stft
        ctx.stft_plan(s["frame_len"], s["frame_hop"], s["n_fft"], s["center"], s["window"])
        C = _channels_and_size(utts[0])[0]
        if any(_channels_and_size(u)[0] != C for u in utts):
            raise ValueError("BatchDereverb.run needs the same channel count in every utterance")
        if s["n_fft"] == 512 and C <= 8:
            return self._run_native(utts, C)
        return self._run_torch(utts, C)

    def _run_native(self, utts, C):
        """One slab of samples up, setk_stft_batch -> setk_wpe_batch -> setk_istft (->
        setk_float_to_pcm16), one slab of waveforms down: the library's own buffers and stream."""
        ctx, F = self.ctx, self.num_bins
        if self._slabs is None:
            self._slabs = _Slabs(ctx)
        b = self._slabs
        esz = 2 if self.pcm16 else 4
        aptr, ns, off_out, n_out = b.stage_audio(
            utts, C, lambda N: esz * C * ctx.istft_num_samples(ctx.num_frames(N)))
        frames = [ctx.num_frames(N) for N in ns]
        lens = [ctx.istft_num_samples(T) for T in frames]
        al = lambda v: (v + 255) & ~255  # noqa: E731
        # scratch: spectrogram in / out per utterance, float waveforms when PCM16 goes out
        need, spec_in, spec_out, wav32 = 0, [], [], []
        for T, L in zip(frames, lens):
            spec_in.append(need)
            need = al(need + 8 * C * T * F)
            spec_out.append(need)
            need = al(need + 8 * C * T * F)
            wav32.append(need)
            if self.pcm16:
                need = al(need + 4 * C * L)
        if need > self._scratch_cap:
            ctx.stream_synchronize(b.stream)
            if self._scratch:
                ctx.device_free(self._scratch)
            self._scratch_cap = int(need * 1.25)
            self._scratch = ctx.device_alloc(self._scratch_cap)
        base = self._scratch
        ctx.stft_batch(C, aptr, ns, [base + o for o in spec_in], stream=b.stream)
        status = np.zeros((len(utts), F), dtype=np.int32)
        ctx.wpe_batch([base + o for o in spec_in], C, frames, F, self.taps, self.delay, self.context,
                      self.num_iters, [base + o for o in spec_out], status=status, stream=b.stream)
        for k, (T, L) in enumerate(zip(frames, lens)):
            if self.pcm16:
                ctx.istft(base + spec_out[k], C, T, None, None, base + wav32[k], stream=b.stream)
                ctx.float_to_pcm16(base + wav32[k], C, L, b.d_out + off_out[k], stream=b.stream)
            else:
                ctx.istft(base + spec_out[k], C, T, None, None, b.d_out + off_out[k], stream=b.stream)
        host = b.fetch(n_out)
        out = []
        self.rank_deficient_bins += int(np.count_nonzero(status == _ffi.NUM_RANKDEF))
        for k, L in enumerate(lens):
            if _ffi.wpe_failed(status[k]).any():
                out.append(None)
            elif self.pcm16:
                out.append(np.frombuffer(host[off_out[k]:off_out[k] + 2 * C * L],
                                         dtype=np.int16).reshape(L, C).copy())
            else:
                out.append(np.frombuffer(host[off_out[k]:off_out[k] + 4 * C * L],
                                         dtype=np.float32).reshape(C, L).copy())
        return out

    def _run_torch(self, utts, C):
        torch, ctx, dev, F = self.torch, self.ctx, self.dev, self.num_bins
        s = self.stft
        audio, ns = [], []
        for samps in utts:
            if isinstance(samps, Pcm16Frames):
                N = samps.frames.shape[0]
                a = torch.empty((C, N), dtype=torch.float32, device=dev)
                ctx.pcm16_to_float(torch.from_numpy(samps.frames).to(dev), C, N, a)
            else:
                samps = np.ascontiguousarray(samps, dtype=np.float32)
                a = torch.from_numpy(samps[None] if samps.ndim == 1 else samps).to(dev)
                N = a.shape[1]
            audio.append(a)
            ns.append(N)
        frames = [ctx.num_frames(N) for N in ns]
        specs = [torch.empty((C, T, F), dtype=torch.complex64, device=dev) for T in frames]
        if s["n_fft"] == 512 and C <= 8:
            ctx.stft_batch(C, [a.data_ptr() for a in audio], ns, [t.data_ptr() for t in specs])
        else:
            for a, t in zip(audio, specs):
                ctx.stft(a, t)
        outs = [torch.empty_like(t) for t in specs]
        status = np.zeros((len(utts), F), dtype=np.int32)
        ctx.wpe_batch(specs, C, frames, F, self.taps, self.delay, self.context, self.num_iters,
                      outs, status=status)
        lens = [ctx.istft_num_samples(T) for T in frames]
        waves = torch.empty((C * sum(lens),), dtype=torch.float32, device=dev)
        views, off = [], 0
        for t, T, L in zip(outs, frames, lens):
            w = waves[off:off + C * L].view(C, L)
            ctx.istft(t, C, T, None, None, w)
            views.append((off, L))
            off += C * L
        if self.pcm16:
            q = torch.round(waves.double() * 32767.0).to(torch.int64).to(torch.int16)
            # channel-major C x L per utterance -> interleaved frames L x C
            host = torch.cat([q[o:o + C * L].view(C, L).t().reshape(-1) for o, L in views]).cpu().numpy()
            return [None if _ffi.wpe_failed(status[u]).any() else host[o:o + C * L].reshape(L, C)
                    for u, (o, L) in enumerate(views)]
        host = waves.cpu().numpy()
        return [None if _ffi.wpe_failed(status[u]).any() else host[o:o + C * L].reshape(C, L)
                for u, (o, L) in enumerate(views)]


class _Twin(object):
    """A grow-only page-locked host buffer with a device twin (one memcpy up or down per batch)."""

    def __init__(self, ctx):
        self.ctx, self.cap, self.h, self.d, self.view = ctx, 0, 0, 0, None

    def reserve(self, nbytes, stream):
        if nbytes > self.cap:
            self.ctx.stream_synchronize(stream)
            self.close()
            self.cap = int(nbytes * 1.25) + 256
            self.h, self.view = self.ctx.host_alloc(self.cap)
            self.d = self.ctx.device_alloc(self.cap)

    def close(self):
        if self.h:
            self.view = None
            self.ctx.host_free(self.h)
            self.ctx.device_free(self.d)
        self.cap, self.h, self.d = 0, 0, 0


class _DfLane(object):
    """One of the two in-flight halves of BatchDirectionalFeatures: its own library handle (a handle
    orders its calls on one stream at a time), slabs, mask twin and device scratch."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.slabs = _Slabs(ctx)
        self.masks = _Twin(ctx)
        self.scratch, self.scratch_cap = 0, 0
        self.pending = None

    def close(self):
        self.slabs.close()
        self.masks.close()
        if self.scratch:
            self.ctx.device_free(self.scratch)
            self.scratch, self.scratch_cap = 0, 0


class BatchDirectionalFeatures(object):
    """Directional features from TF masks for a batch of utterances, resident on the device.

    Replaces the per-utterance body of funcwj/setk scripts/sptk/compute_df_on_mask.py:40-54
    (SpectrogramReader -> compute_covar -> solve_pevd -> directional_feats, libs/spatial.py:
    184-208): the samples of a chunk go up in one slab, ONE setk_stft_batch launch writes every
    spectrogram, and per utterance setk_covar -> setk_pevd -> setk_directional_feats run on
    device pointers -- the spectrogram (31 MB at 8 ch x 30 s), the covariance and the steer
    vector never visit the host; one slab of T x F features and the per-bin status words comes
    down per chunk.  Two chunks are in flight (two library handles, two streams, each driven by
    its own host thread): the staging copies, the upload and the launches of one overlap the
    kernels and the download of the other -- the path is bound by the host's copies into and out
    of the page-locked slabs and the PCIe transfer of samples, masks and feature maps.  run() takes [(samps C x N
    float32 | Pcm16Frames, mask T x F or F x T)] and returns [(features T x F float32 | None,
    status)]: status != 0 is numpy's LinAlgError case (np.linalg.eigh on a non-finite
    covariance).  Other transform sizes and more than 8 channels go through the stand-alone
    operators of setk_amd.libs (numpy in, numpy out)."""

    def __init__(self, df_pair, frame_len=512, frame_hop=256, center=True, round_power_of_two=True,
                 window="hann", device=None, max_batch_samples=1 << 28, chunk_utts=8):
        pairs = [(int(i), int(j)) for i, j in df_pair]
        if not pairs:
            raise ValueError("no microphone pair given")
        self.pairs = pairs
        self.ctx = _ffi.default_context(device)
        n_fft = nextpow2(frame_len) if round_power_of_two else frame_len
        self.stft = dict(frame_len=frame_len, frame_hop=frame_hop, n_fft=n_fft, center=center,
                         window=stft_window(window, frame_len))
        self.window_name = window
        self.round_power_of_two = round_power_of_two
        self.num_bins = n_fft // 2 + 1
        self.max_batch_samples = max_batch_samples
        self.chunk_utts = max(1, int(chunk_utts))
        self._lanes = []

    def close(self):
        lanes, self._lanes = self._lanes, []
        for k, lane in enumerate(lanes):
            lane.close()
            if k > 0:
                lane.ctx.close()  # (lane 0 runs on the process-wide context)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def condition_mask(self, mask, T, out=None):
        """compute_df_on_mask.py:44-47: F x T masks are turned, values above one clipped.  With
        `out` (T x F float32, e.g. a view of the page-locked upload buffer) the clipped mask is
        written there in one pass."""
        F = self.num_bins
        m = np.asarray(mask)
        if m.ndim != 2:
            raise ValueError(f"mask must be 2-D, got {m.shape}")
        if m.shape[0] == F and m.shape != (T, F):
            m = m.T
        if m.shape != (T, F):
            raise ValueError(f"mask {np.asarray(mask).shape} does not fit {T} frames x {F} bins")
        if out is not None:
            return np.minimum(m, 1, out=out, casting="unsafe")
        return np.minimum(m, 1).astype(np.float32, copy=False)

    def _lane(self, k):
        s = self.stft
        while len(self._lanes) <= k:
            ctx = self.ctx if not self._lanes else _ffi.Context(self.ctx.device)
            ctx.stft_plan(s["frame_len"], s["frame_hop"], s["n_fft"], s["center"], s["window"])
            self._lanes.append(_DfLane(ctx))
        return self._lanes[k]

    def run(self, utts):
        s = self.stft
        self.ctx.stft_plan(s["frame_len"], s["frame_hop"], s["n_fft"], s["center"], s["window"])
        out = [None] * len(utts)
        by_channels = {}
        for i, (samps, _) in enumerate(utts):
            by_channels.setdefault(_channels_and_size(samps)[0], []).append(i)
        chunks = []  # (channel count, utterance indices)
        for C, idx in by_channels.items():
            if any(max(p) >= C or min(p) < 0 for p in self.pairs):
                raise ValueError(f"microphone pair out of range for {C} channels: {self.pairs}")
            if s["n_fft"] != 512 or C > 8:
                for i in idx:
                    out[i] = self._one_by_operators(*utts[i])
                continue
            chunk, load = [], 0
            for i in idx + [None]:
                n = 0 if i is None else _channels_and_size(utts[i][0])[1]
                if chunk and (i is None or len(chunk) >= self.chunk_utts or load + n > self.max_batch_samples):
                    chunks.append((C, chunk))
                    chunk, load = [], 0
                if i is not None:
                    chunk.append(i)
                    load += n
        if len(chunks) == 1:
            lane = self._lane(0)
            self._submit(lane, utts, chunks[0][1], chunks[0][0])
            self._collect(lane, out)
        elif chunks:
            # two lanes, each driven by its own thread (the staging copies and the library calls
            # release the interpreter lock): chunk k goes to lane k & 1, a lane stages, launches and
            # fetches one chunk at a time while the other lane does the same half a period apart
            lanes = [self._lane(0), self._lane(1)]
            errors = []

            def drive(k):
                try:
                    for C, chunk in chunks[k::2]:
                        self._submit(lanes[k], utts, chunk, C)
                        self._collect(lanes[k], out)
                except BaseException as e:  # noqa: B902 (re-raised in the caller's thread)
                    errors.append(e)

            other = threading.Thread(target=drive, args=(1,), name="setk-df-lane1")
            other.start()
            drive(0)
            other.join()
            if errors:
                raise errors[0]
        return out

    def _submit(self, lane, utts, batch, C):
        """Everything of one chunk, enqueued on the lane's stream; nothing waits here."""
        ctx, F = lane.ctx, self.num_bins
        b, mk = lane.slabs, lane.masks
        al = lambda v: (v + 255) & ~255  # noqa: E731
        # out-slab per utterance: [features T x F float32 | status int32[F]]
        aptr, ns, off_out, n_out = b.stage_audio(
            [utts[i][0] for i in batch], C, lambda N: al(4 * ctx.num_frames(N) * F) + 4 * F)
        frames = [ctx.num_frames(N) for N in ns]
        moff, need_m = [], 0
        for T in frames:
            moff.append(need_m)
            need_m = al(need_m + 4 * T * F)
        mk.reserve(need_m, b.stream)
        for i, T, o in zip(batch, frames, moff):
            # (clipped straight into the page-locked buffer: one pass over the mask)
            self.condition_mask(utts[i][1], T, out=mk.view[o:o + 4 * T * F].view(np.float32).reshape(T, F))
        ctx.memcpy_h2d_async(mk.d, mk.h, need_m, b.stream)
        # device scratch: spectrogram [C][T][F], covariance [F][C][C], steer vector [F][C] per utterance
        need, spec, cov, sv = 0, [], [], []
        for T in frames:
            spec.append(need)
            need = al(need + 8 * C * T * F)
            cov.append(need)
            need = al(need + 8 * F * C * C)
            sv.append(need)
            need = al(need + 8 * F * C)
        if need > lane.scratch_cap:
            ctx.stream_synchronize(b.stream)
            if lane.scratch:
                ctx.device_free(lane.scratch)
            lane.scratch_cap = int(need * 1.25)
            lane.scratch = ctx.device_alloc(lane.scratch_cap)
        base = lane.scratch
        ctx.stft_batch(C, aptr, ns, [base + o for o in spec], stream=b.stream)
        for k, T in enumerate(frames):
            o_df = b.d_out + off_out[k]
            o_st = o_df + al(4 * T * F)
            ctx.covar(base + spec[k], mk.d + moff[k], C, T, F, base + cov[k], stream=b.stream)
            ctx.pevd(base + cov[k], None, F, C, 0, base + sv[k], o_st, stream=b.stream)
            ctx.directional_feats(base + spec[k], base + sv[k], self.pairs, C, T, F, o_df, stream=b.stream)
        ctx.memcpy_d2h_async(b.h_out, b.d_out, n_out, b.stream)
        lane.pending = (batch, frames, off_out)

    def _collect(self, lane, out):
        if lane.pending is None:
            return
        batch, frames, off_out = lane.pending
        lane.pending = None
        F = self.num_bins
        al = lambda v: (v + 255) & ~255  # noqa: E731
        lane.ctx.stream_synchronize(lane.slabs.stream)
        host = lane.slabs.np_out
        for k, (i, T) in enumerate(zip(batch, frames)):
            o_df = off_out[k]
            o_st = o_df + al(4 * T * F)
            status = np.frombuffer(host[o_st:o_st + 4 * F], dtype=np.int32)
            code = int(status.max()) if F else 0
            df = None
            if code == 0:
                df = np.frombuffer(host[o_df:o_df + 4 * T * F], dtype=np.float32).reshape(T, F).copy()
            out[i] = (df, code)

    def _one_by_operators(self, samps, mask):
        """n_fft != 512 or more than 8 channels: the mirrored operators, one utterance at a time."""
        from .libs.beamformer import compute_covar, solve_pevd
        from .libs.spatial import directional_feats
        from .libs.utils import forward_stft
        if isinstance(samps, Pcm16Frames):
            samps = samps.to_float()
        samps = np.ascontiguousarray(samps, dtype=np.float32)
        if samps.ndim == 1:
            samps = samps[None]
        s = self.stft
        obs = np.stack([forward_stft(ch, frame_len=s["frame_len"], frame_hop=s["frame_hop"],
                                     round_power_of_two=self.round_power_of_two, center=s["center"],
                                     window=self.window_name, transpose=False) for ch in samps])
        m = self.condition_mask(mask, obs.shape[2])
        try:
            sv = solve_pevd(compute_covar(obs, m))
        except np.linalg.LinAlgError:
            return None, _ffi.NUM_NONFINITE
        return directional_feats(obs, sv.T, df_pair=self.pairs), 0


class BatchWpd(object):
    """Factorised WPD (joint dereverberation and denoising) for a batch, resident on the device.

    Replaces the per-utterance body of funcwj/setk scripts/sptk/apply_wpd.py:31-57 around
    libs/wpe.py:113-177 (facted_wpd): per outer iteration one WPE step with the variances of the
    previous enhanced signal, a K = 2 CGMM on the dereverberated channels, the power-weighted
    and the mask-weighted covariance, the MVDR weights and the beamformer.  Where the numpy
    mirror (setk_amd.libs.wpe.facted_wpd) carries every intermediate through host arrays, here
    the samples of a batch go up once, setk_stft_batch writes the spectrograms, and every stage
    works on device pointers of one scratch block, outer iteration by outer iteration: setk_wpe per
    utterance, ONE setk_cgmm_masks_batch for the batch, setk_covar x 2 -> setk_weights ->
    setk_beamform per utterance, then setk_istft with the renorm to max |samples|
    (SpectrogramReader.maxabs) and the float -> PCM_16 conversion; one slab comes down per batch:
    [wave | status words | speech mask].  run() takes C x N float32 arrays or Pcm16Frames of one
    channel count and returns [(wave, mask T x F float32) | None]; None is the reference's
    LinAlgError (singular tap correlation or power-weighted covariance).  The transform sizes the
    fused STFT does not serve (n_fft != 512, more than 8 channels) go through the numpy mirror."""

    def __init__(self, taps=10, delay=3, context=1, wpd_iters=3, cgmm_iters=20, update_alpha=False,
                 frame_len=512, frame_hop=256, center=True, round_power_of_two=True, window="hann",
                 device=None, pcm16=False):
        self.ctx = _ffi.default_context(device)
        self.taps, self.delay, self.context = int(taps), int(delay), int(context)
        self.wpd_iters, self.cgmm_iters = int(wpd_iters), int(cgmm_iters)
        self.update_alpha = bool(update_alpha)
        self.pcm16 = bool(pcm16)
        n_fft = nextpow2(frame_len) if round_power_of_two else frame_len
        self.stft = dict(frame_len=frame_len, frame_hop=frame_hop, n_fft=n_fft, center=center,
                         window=stft_window(window, frame_len))
        self.window_name, self.round_power_of_two = window, round_power_of_two
        self.num_bins = n_fft // 2 + 1
        self.rank_deficient_bins = 0
        self._slabs = None
        self._scratch, self._scratch_cap = 0, 0
        self._cgmm_per_utt = os.environ.get("SETK_WPD_CGMM_PER_UTT") == "1"

    def close(self):
        b, self._slabs = self._slabs, None
        if b:
            b.close()
        if self._scratch:
            self.ctx.device_free(self._scratch)
            self._scratch, self._scratch_cap = 0, 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, utts):
        if not len(utts):
            return []
        s = self.stft
        self.ctx.stft_plan(s["frame_len"], s["frame_hop"], s["n_fft"], s["center"], s["window"])
        C = _channels_and_size(utts[0])[0]
        if any(_channels_and_size(u)[0] != C for u in utts):
            raise ValueError("BatchWpd.run needs the same channel count in every utterance")
        if s["n_fft"] == 512 and C <= 8:
            return self._run_resident(utts, C)
        return [self._one_by_mirror(u) for u in utts]

    @staticmethod
    def _peak(samps):
        if isinstance(samps, Pcm16Frames):
            return float(np.abs(samps.frames.astype(np.int32)).max()) / 32768.0 if samps.frames.size else 0.0
        return float(np.max(np.abs(samps))) if np.size(samps) else 0.0

    def _run_resident(self, utts, C):
        ctx, F, K = self.ctx, self.num_bins, self.wpd_iters
        if self._slabs is None:
            self._slabs = _Slabs(ctx)
        b = self._slabs
        al = lambda v: (v + 255) & ~255  # noqa: E731
        esz = 2 if self.pcm16 else 4
        n_status = K * F  # per outer iteration: the MVDR solve (WPE's words come back with its call)

        def out_bytes(N):
            T = ctx.num_frames(N)
            return al(esz * ctx.istft_num_samples(T)) + al(4 * n_status) + al(4 * T * F) + 256

        aptr, ns, off_out, n_out = b.stage_audio(utts, C, out_bytes)
        frames = [ctx.num_frames(N) for N in ns]
        lens = [ctx.istft_num_samples(T) for T in frames]
        # scratch per utterance: spectrogram, dereverberated channels, 1 / lambda,
        # two covariances, weights, enhanced spectrum, float wave (PCM16 output)
        lay, need = [], 0

        def take(nbytes):
            nonlocal need
            o = need
            need = al(need + nbytes)
            return o

        for T, L in zip(frames, lens):
            lay.append(dict(spec=take(8 * C * T * F), der=take(8 * C * T * F), inv=take(4 * T * F),
                            Rd=take(8 * F * C * C), Rs=take(8 * F * C * C),
                            w=take(8 * F * C), enh=take(8 * T * F), wav=take(4 * L), norm=take(256)))
        if need > self._scratch_cap:
            ctx.stream_synchronize(b.stream)
            if self._scratch:
                ctx.device_free(self._scratch)
            self._scratch_cap = int(need * 1.25)
            self._scratch = ctx.device_alloc(self._scratch_cap)
        base, st = self._scratch, b.stream
        ctx.stft_batch(C, aptr, ns, [base + q["spec"] for q in lay], stream=st)
        mvdr = _ffi.BfOpts(kind=_ffi.BF_MVDR)
        peaks = [np.array([self._peak(u)], dtype=np.float32) for u in utts]  # (kept alive until the fetch)
        d_wave = [b.d_out + off_out[k] for k in range(len(utts))]
        d_stat = [d_wave[k] + al(esz * L) for k, L in enumerate(lens)]
        d_mask = [d_stat[k] + al(4 * n_status) for k in range(len(utts))]
        # iteration-major: the CGMM of an outer iteration is ONE launch over (bin, utterance) for
        # the whole batch -- 257 workgroups of one 10 s utterance fill an eighth of the chip, and the
        # EM's 22 passes are latency, not throughput, at that size
        wpe_status = np.zeros((K, len(utts), F), dtype=np.int32)
        for it in range(K):
            # (the WPE step too: one launch over (bin, utterance); its status words come back with
            # the call, which drains the stream as every setk_wpe* call does)
            ctx.wpe_batch_var([base + q["spec"] for q in lay], C, frames, F, self.taps, self.delay,
                              self.context, 1, [base + q["der"] for q in lay],
                              lambda_enh=[base + q["enh"] for q in lay] if it else None,
                              inv_lambda_outs=[base + q["inv"] for q in lay], status=wpe_status[it],
                              stream=st)
            if self._cgmm_per_utt:  # A/B only (SETK_WPD_CGMM_PER_UTT=1): one EM launch per utterance
                for k, (q, T) in enumerate(zip(lay, frames)):
                    ctx.cgmm_masks_batch(C, [base + q["der"]], [T], F, self.cgmm_iters, None, [d_mask[k]],
                                         stream=st, update_alpha=self.update_alpha)
            else:
                ctx.cgmm_masks_batch(C, [base + q["der"] for q in lay], frames, F, self.cgmm_iters, None,
                                     d_mask, stream=st, update_alpha=self.update_alpha)
            for k, (q, T) in enumerate(zip(lay, frames)):
                p = lambda name: base + q[name]  # noqa: E731
                # the mask 1 / lambda gives the power-weighted covariance up to a per-bin scale
                # that cancels in the MVDR weight
                ctx.covar(p("der"), p("inv"), C, T, F, p("Rd"), stream=st)
                ctx.covar(p("der"), d_mask[k], C, T, F, p("Rs"), stream=st)
                ctx.weights(mvdr, p("Rs"), p("Rd"), None, F, C, p("w"), d_stat[k] + 4 * F * it, stream=st)
                ctx.beamform(p("w"), p("der"), C, T, F, p("enh"), stream=st)
        for k, (q, T, L) in enumerate(zip(lay, frames, lens)):
            p = lambda name: base + q[name]  # noqa: E731
            ctx.memcpy_h2d_async(p("norm"), peaks[k].ctypes.data, 4, st)
            if self.pcm16:
                ctx.istft(p("enh"), 1, T, None, p("norm"), p("wav"), stream=st)
                ctx.float_to_pcm16(p("wav"), 1, L, d_wave[k], stream=st)
            else:
                ctx.istft(p("enh"), 1, T, None, p("norm"), d_wave[k], stream=st)
        host = b.fetch(n_out)
        out = []
        for k, (T, L) in enumerate(zip(frames, lens)):
            o_wave = off_out[k]
            o_stat = o_wave + al(esz * L)
            o_mask = o_stat + al(4 * n_status)
            status = np.frombuffer(host[o_stat:o_stat + 4 * n_status], dtype=np.int32).reshape(K, F)
            self.rank_deficient_bins += int(np.count_nonzero(wpe_status[:, k] == _ffi.NUM_RANKDEF))
            if _ffi.wpe_failed(wpe_status[:, k]).any() or status.any():
                out.append(None)
                continue
            wave = np.frombuffer(host[o_wave:o_wave + esz * L], dtype=np.int16 if self.pcm16 else np.float32).copy()
            mask = np.frombuffer(host[o_mask:o_mask + 4 * T * F], dtype=np.float32).reshape(T, F).copy()
            out.append((wave, mask))
        return out

    def _one_by_mirror(self, samps):
        from .libs.utils import forward_stft, inverse_stft
        from .libs.wpe import facted_wpd
        from .libs import wavio
        if isinstance(samps, Pcm16Frames):
            samps = samps.to_float()
        samps = np.ascontiguousarray(samps, dtype=np.float32)
        if samps.ndim == 1:
            samps = samps[None]
        s = self.stft
        kw = dict(frame_len=s["frame_len"], frame_hop=s["frame_hop"], center=s["center"], window=self.window_name)
        obs = np.stack([forward_stft(ch, round_power_of_two=self.round_power_of_two, transpose=True, **kw)
                        for ch in samps])  # N x T x F
        try:
            tf_mask, enh = facted_wpd(obs, wpd_iters=self.wpd_iters, cgmm_iters=self.cgmm_iters,
                                      update_alpha=self.update_alpha, context=self.context,
                                      taps=self.taps, delay=self.delay)
        except np.linalg.LinAlgError:
            return None
        wave = inverse_stft(enh, norm=float(np.max(np.abs(samps))), transpose=True, **kw)
        if self.pcm16:
            wave = wavio.float_to_pcm16(wave)
        return wave, tf_mask[..., 0].astype(np.float32)
