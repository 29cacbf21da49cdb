"""
ctypes binding of libsetk_hip.so (include/setk_hip.h).

There is no CPU fallback: if the HIP library has not been built or no GPU is
visible, every operator raises.  PyTorch is imported first so that its bundled
HIP runtime (same SONAME, libamdhip64.so.7) is the one the library binds to --
torch tensors and this library then share one runtime, one device context and
torch's streams.  A process that needs none of that (the streaming CLI: buffers,
streams and events come from the library itself) sets TORCH_FREE and skips the import.
"""
import ctypes
import os
import sys
from ctypes import (POINTER, c_char_p, c_float, c_int, c_void_p)

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SETK_LIB", os.path.join(_HERE, "libsetk_hip.so"))

SETK_OK = 0
ERR_INVALID, ERR_UNSUPPORTED, ERR_HIP, ERR_NOMEM = -1, -2, -3, -4
NUM_OK, NUM_SINGULAR, NUM_NOCONV, NUM_NONFINITE = 0, 1, 2, 3
NUM_RANKDEF = 4  # WPE: rank-deficient tap correlation, columns dropped -- a note, not an error


def wpe_failed(status):
    """Boolean array: the bins whose WPE status is an error (SETK_NUM_RANKDEF is only a note)."""
    import numpy as _np
    st = _np.asarray(status)
    return (st != NUM_OK) & (st != NUM_RANKDEF)
BF_MVDR, BF_GEVD, BF_PMWF, BF_MPDR, BF_MPDR_WHITEN = 0, 1, 2, 3, 4
RANK1_NONE, RANK1_EIG, RANK1_GEV = 0, 1, 2
FLAG_BAN, FLAG_CLAMP_MASK, FLAG_POST_MASK, FLAG_NO_GAUGE, FLAG_OUT_PCM16 = 1, 2, 4, 8, 16
FLAG_NO_RENORM = 32
FLAG_STRICT_REFERENCE = 64  # numpy.linalg.solve's exact-zero-pivot refusals (setk_hip.h)
FLAG_IN_PCM16 = 128  # enhance_batch: audio[u] is planar int16 [C][pcm16_channel_stride(N)]
CGMM_UPDATE_ALPHA = 1


class BfOpts(ctypes.Structure):
    _fields_ = [("kind", c_int), ("flags", c_int), ("pmwf_beta", c_float),
                ("pmwf_ref", c_int), ("rank1", c_int)]


class BatchTaps(ctypes.Structure):
    _fields_ = [("Rs", c_void_p), ("Rn", c_void_p), ("weight", c_void_p), ("maxabs", c_void_p)]


class SetkError(RuntimeError):
    pass


class SetkUnsupported(SetkError, NotImplementedError):
    pass


_lib = None


# The streaming CLI needs no torch at all (buffers, streams and events come from the library):
# it sets this before the first Context so that the second of `import torch` is not spent.
# The library then runs on the HIP runtime of /opt/rocm; a process that is going to use torch
# tensors with this library must import torch BEFORE the library is loaded (the default).
TORCH_FREE = os.environ.get("SETK_TORCH_FREE", "") not in ("", "0")


def set_torch_free(ok=True):
    """What the CLIs call before the first Context: run torch-free unless the caller knows an
    input needs a torch path (`ok` false), the user said SETK_TORCH_FREE=0, or torch is loaded
    already.  Returns whether the process is torch-free now."""
    global TORCH_FREE
    if not ok or os.environ.get("SETK_TORCH_FREE", "") == "0" or "torch" in sys.modules:
        return TORCH_FREE and "torch" not in sys.modules
    TORCH_FREE = True
    return True


def import_torch(what="this path"):
    """torch for the code paths that run through torch tensors.  In a process that loaded the
    library torch-free the import would come too late (the library is bound to the HIP runtime
    of /opt/rocm by then; torch must be imported BEFORE it): say so instead of importing."""
    if TORCH_FREE and _lib is not None and "torch" not in sys.modules:
        raise SetkUnsupported(
            f"{what} runs through torch tensors, but this process loaded {LIB_PATH} torch-free "
            "(streaming CLI mode); re-run with SETK_TORCH_FREE=0")
    import torch
    return torch


def env_atoi(name, default):
    """An integer environment switch as the library's C side parses it (atoi: optional sign and
    leading digits, anything else is 0); `default` when the variable is unset."""
    import re
    v = os.environ.get(name)
    if v is None:
        return default
    m = re.match(r"\s*([+-]?\d+)", v)
    return int(m.group(1)) if m else 0


def exported_symbols():
    """Names declared in include/setk_hip.h (checked by the CPU test-suite)."""
    return [
        "setk_abi_version", "setk_create", "setk_destroy", "setk_last_error", "setk_device_pci_bus_id",
        "setk_memcpy_d2h_async", "setk_device_alloc", "setk_device_free", "setk_host_alloc",
        "setk_host_free", "setk_stream_create", "setk_stream_destroy", "setk_stream_synchronize",
        "setk_stream_wait_event", "setk_event_create", "setk_event_destroy", "setk_event_record",
        "setk_event_synchronize",
        "setk_host_register", "setk_host_unregister", "setk_memcpy_h2d_async",
        "setk_stft_plan", "setk_stft_num_frames", "setk_istft_num_samples",
        "setk_stft", "setk_stft_batch", "setk_istft", "setk_covar", "setk_pevd", "setk_weights",
        "setk_pcm16_to_float", "setk_pcm16_to_float_batch", "setk_pcm16_channel_stride", "setk_pcm16_deinterleave_batch", "setk_kaldi_cm_decode_batch", "setk_float_to_pcm16", "setk_ban", "setk_rank1", "setk_beamform", "setk_cgmm_masks", "setk_cgmm_masks_k", "setk_cgmm_masks_k_status",
        "setk_cgmm_masks_batch", "setk_cgmm_estimate_batch", "setk_enhance_batch", "setk_enhance_batch_taps",
        "setk_apply_weights_batch",
        "setk_directional_feats", "setk_wpe", "setk_wpe_step", "setk_wpe_batch", "setk_wpe_batch_fnt", "setk_wpe_batch_var", "setk_set_profiling",
        "setk_last_stage_ms",
        "setk_comm_unique_id", "setk_comm_create", "setk_comm_allreduce_f64", "setk_comm_barrier",
        "setk_comm_destroy", "setk_comm_last_error", "setk_host_read_payloads"
    ]


def load_library():
    """dlopen libsetk_hip.so and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SetkError(
            f"{LIB_PATH} is missing: build it with `python -m setk_amd.build` "
            "(there is no CPU fallback)")
    if not TORCH_FREE or "torch" in sys.modules:
        try:
            import torch  # noqa: F401  (loads torch's libamdhip64.so.7 first)
        except Exception:  # pragma: no cover - torch is plumbing, not required
            pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    if hasattr(lib, "hoststub_report") and os.environ.get("SETK_ALLOW_HOSTSTUB") != "1":
        # tools/hoststub: the library linked against a host-memory stand-in for HIP whose
        # kernels do nothing -- test infrastructure for the host side, never a way to run
        raise SetkError(f"{LIB_PATH} is the test build against the HIP stand-in (tools/hoststub); "
                        "it computes nothing.  Tests that mean to load it set SETK_ALLOW_HOSTSTUB=1")
    H = c_void_p
    fp = c_void_p  # raw data pointers (host or device)
    lib.setk_abi_version.restype = c_int
    lib.setk_create.argtypes = [POINTER(H), c_int]
    lib.setk_destroy.argtypes = [H]
    lib.setk_last_error.argtypes = [H]
    lib.setk_last_error.restype = c_char_p
    lib.setk_device_pci_bus_id.argtypes = [H, ctypes.c_char_p, c_int]
    lib.setk_device_pci_bus_id.restype = c_int
    lib.setk_host_register.argtypes = [H, c_void_p, ctypes.c_size_t]
    lib.setk_host_unregister.argtypes = [H, c_void_p]
    lib.setk_memcpy_h2d_async.argtypes = [H, c_void_p, c_void_p, ctypes.c_size_t, c_void_p]
    lib.setk_memcpy_d2h_async.argtypes = [H, c_void_p, c_void_p, ctypes.c_size_t, c_void_p]
    for name in ("setk_device_alloc", "setk_host_alloc"):
        getattr(lib, name).argtypes = [H, ctypes.c_size_t, POINTER(c_void_p)]
    for name in ("setk_device_free", "setk_host_free", "setk_stream_destroy", "setk_stream_synchronize",
                 "setk_event_destroy", "setk_event_synchronize"):
        getattr(lib, name).argtypes = [H, c_void_p]
    for name in ("setk_stream_create", "setk_event_create"):
        getattr(lib, name).argtypes = [H, POINTER(c_void_p)]
    lib.setk_stream_wait_event.argtypes = [H, c_void_p, c_void_p]
    lib.setk_event_record.argtypes = [H, c_void_p, c_void_p]
    lib.setk_set_profiling.argtypes = [H, c_int]
    lib.setk_last_stage_ms.argtypes = [H, POINTER(c_float)]
    lib.setk_stft_plan.argtypes = [H, c_int, c_int, c_int, c_int, fp]
    lib.setk_stft_num_frames.argtypes = [H, c_int]
    lib.setk_istft_num_samples.argtypes = [H, c_int, c_int]
    lib.setk_stft.argtypes = [H, fp, c_int, c_int, fp, c_void_p]
    lib.setk_stft_batch.argtypes = [H, c_int, c_int, POINTER(c_void_p), POINTER(c_int),
                                    POINTER(c_void_p), c_int, c_void_p]
    lib.setk_istft.argtypes = [H, fp, c_int, c_int, c_int, fp, fp, c_void_p]
    lib.setk_covar.argtypes = [H, fp, fp, c_int, c_int, c_int, fp, c_void_p]
    lib.setk_pevd.argtypes = [H, fp, fp, c_int, c_int, c_int, fp, fp, c_void_p]
    lib.setk_weights.argtypes = [H, POINTER(BfOpts), fp, fp, fp, c_int, c_int,
                                 fp, fp, POINTER(c_int), c_void_p]
    lib.setk_ban.argtypes = [H, fp, fp, c_int, c_int, fp, c_void_p]
    lib.setk_pcm16_to_float.argtypes = [H, c_void_p, c_int, c_int, fp, c_void_p]
    lib.setk_float_to_pcm16.argtypes = [H, fp, c_int, c_int, c_void_p, c_void_p]
    lib.setk_pcm16_channel_stride.argtypes = [c_int]
    lib.setk_pcm16_deinterleave_batch.argtypes = [H, c_int, c_int, POINTER(c_void_p), POINTER(c_int),
                                                  POINTER(c_void_p), c_void_p, c_void_p]
    lib.setk_pcm16_to_float_batch.argtypes = [H, c_int, c_int, POINTER(c_void_p), POINTER(c_int),
                                              POINTER(c_void_p), c_void_p, c_void_p]
    lib.setk_kaldi_cm_decode_batch.argtypes = [H, c_int, POINTER(c_int), POINTER(c_float), POINTER(c_float),
                                               POINTER(c_int), POINTER(c_int), POINTER(c_int),
                                               POINTER(c_void_p), POINTER(c_void_p), c_void_p]
    lib.setk_rank1.argtypes = [H, fp, fp, c_int, c_int, fp, fp, c_void_p]
    lib.setk_beamform.argtypes = [H, fp, fp, c_int, c_int, c_int, fp, c_void_p]
    lib.setk_cgmm_masks.argtypes = [H, fp, c_int, c_int, c_int, c_int, fp, fp, fp, c_int, c_void_p]
    lib.setk_cgmm_masks_k.argtypes = [H, fp, c_int, c_int, c_int, c_int, c_int, fp, fp, fp, c_int, c_void_p]
    lib.setk_cgmm_masks_k_status.argtypes = [H, fp, c_int, c_int, c_int, c_int, c_int, fp, fp, fp, c_int, fp,
                                             c_void_p]
    lib.setk_cgmm_masks_batch.argtypes = [H, c_int, c_int, POINTER(c_void_p), POINTER(c_int), c_int,
                                          c_int, POINTER(c_void_p), POINTER(c_void_p), c_int,
                                          c_int, c_void_p]
    lib.setk_cgmm_estimate_batch.argtypes = [H, c_int, c_int, POINTER(c_void_p), POINTER(c_int),
                                             c_int, POINTER(c_void_p), POINTER(c_void_p), c_int,
                                             c_void_p]
    lib.setk_enhance_batch.argtypes = [
        H, POINTER(BfOpts), c_int, c_int, POINTER(c_void_p), POINTER(c_int),
        POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int),
        c_void_p
    ]
    lib.setk_enhance_batch_taps.argtypes = lib.setk_enhance_batch.argtypes[:-1] + [
        POINTER(BatchTaps), c_void_p]
    lib.setk_wpe.argtypes = [H, fp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, fp, fp, fp,
                             fp, c_void_p]
    lib.setk_wpe_step.argtypes = [H, fp, c_int, c_int, c_int, c_int, c_int, fp, fp, fp, c_void_p]
    lib.setk_wpe_batch.argtypes = [H, c_int, POINTER(c_void_p), c_int, POINTER(c_int), c_int, c_int,
                                   c_int, c_int, c_int, POINTER(c_void_p), fp, c_void_p]
    lib.setk_wpe_batch_fnt.argtypes = lib.setk_wpe_batch.argtypes
    lib.setk_host_read_payloads.argtypes = [c_int, POINTER(ctypes.c_char_p), POINTER(ctypes.c_longlong),
                                            POINTER(ctypes.c_longlong), POINTER(c_void_p), c_int,
                                            ctypes.c_longlong, POINTER(c_int)]
    lib.setk_wpe_batch_var.argtypes = [H, c_int, POINTER(c_void_p), c_int, POINTER(c_int), c_int, c_int,
                                       c_int, c_int, c_int, POINTER(c_void_p), POINTER(c_void_p),
                                       POINTER(c_void_p), fp, c_void_p]
    lib.setk_directional_feats.argtypes = [H, fp, fp, POINTER(c_int), c_int, c_int, c_int, c_int,
                                           fp, c_void_p]
    lib.setk_apply_weights_batch.argtypes = [
        H, c_int, c_int, POINTER(c_void_p), POINTER(c_int), fp, c_int, POINTER(c_int),
        POINTER(c_void_p), c_int, c_void_p
    ]
    for name in exported_symbols():
        fn = getattr(lib, name)
        if fn.restype is not c_char_p:
            fn.restype = c_int
    _lib = lib
    return lib


def host_read_payloads(paths, offsets, nbytes, dsts, threads, mmap_min_bytes):
    """setk_host_read_payloads: payload i = nbytes[i] bytes at offsets[i] of paths[i] -> address
    dsts[i], by the library's pool of native reader threads.  Returns the list of errno values
    (0: read).  No handle, no device: the call releases the interpreter lock for the whole batch."""
    n = len(paths)
    if not n:
        return []
    lib = load_library()
    P = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
    O = (ctypes.c_longlong * n)(*[int(v) for v in offsets])
    B = (ctypes.c_longlong * n)(*[int(v) for v in nbytes])
    D = (c_void_p * n)(*[int(v) for v in dsts])
    S = (c_int * n)()
    rc = lib.setk_host_read_payloads(n, P, O, B, D, int(threads), int(mmap_min_bytes), S)
    if rc != 0:
        raise SetkError(f"setk_host_read_payloads: bad arguments ({rc})")
    return list(S)


def _ptr(x):
    """Raw address of a numpy array (host) or torch tensor (host/device)."""
    if x is None:
        return None
    if isinstance(x, int):  # a raw (device) address
        return x
    if isinstance(x, np.ndarray):
        if not x.flags["C_CONTIGUOUS"]:
            raise ValueError("array must be C-contiguous")
        return x.ctypes.data
    if hasattr(x, "data_ptr"):
        if not x.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return x.data_ptr()
    raise TypeError(f"unsupported buffer type {type(x)}")


def _torch():
    """torch, or None in a torch-free process (TORCH_FREE and nobody imported it)."""
    if TORCH_FREE and "torch" not in sys.modules:
        return None
    try:
        import torch
        return torch
    except Exception:  # pragma: no cover
        return None


def current_stream_ptr():
    """torch's current HIP stream as a raw hipStream_t (0 = default stream)."""
    torch = _torch()
    try:
        if torch is not None and torch.cuda.is_available():
            return torch.cuda.current_stream().cuda_stream
    except Exception:
        pass
    return 0


class Context:
    """Owns one setk_handle_t (one GPU, one host thread at a time)."""

    def __init__(self, device=0):
        self._lib = load_library()
        self._h = c_void_p()
        rc = self._lib.setk_create(ctypes.byref(self._h), int(device))
        if rc != SETK_OK:
            self._h = None
            raise SetkError(
                f"setk_create(device={device}) failed with {rc}: no usable MI355X / "
                "HIP runtime (the product path needs the GPU; there is no CPU fallback)")
        self.device = int(device)
        self.plan = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.setk_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc == SETK_OK:
            return
        msg = self._lib.setk_last_error(self._h)
        msg = msg.decode() if msg else ""
        if rc == ERR_INVALID:
            raise ValueError(msg)
        if rc == ERR_UNSUPPORTED:
            raise SetkUnsupported(msg)
        raise SetkError(f"libsetk_hip error {rc}: {msg}")

    # -- host-memory plumbing (thread safe) -----------------------------------
    def host_register(self, ptr, nbytes):
        """Pin [ptr, ptr + nbytes); returns False when the runtime refuses the range."""
        return self._lib.setk_host_register(self._h, c_void_p(int(ptr)), int(nbytes)) == SETK_OK

    def host_unregister(self, ptr):
        self._lib.setk_host_unregister(self._h, c_void_p(int(ptr)))

    # -- buffers, streams, events (a host pipeline without a runtime binding of its own) ----
    def _out_ptr(self, fn, *args):
        p = c_void_p()
        self.check(fn(self._h, *args, ctypes.byref(p)))
        return p.value or 0

    def pci_bus_id(self):
        """PCI address of the handle's device ("0000:23:00.0"): the key to its NUMA node in
        sysfs (setk_amd/numa.py)."""
        buf = ctypes.create_string_buffer(32)
        self.check(self._lib.setk_device_pci_bus_id(self._h, buf, 32))
        return buf.value.decode().lower()

    def device_alloc(self, nbytes):
        return self._out_ptr(self._lib.setk_device_alloc, ctypes.c_size_t(int(nbytes)))

    def device_free(self, ptr):
        self.check(self._lib.setk_device_free(self._h, c_void_p(ptr)))

    def host_alloc(self, nbytes):
        """Page-locked host memory: (address, uint8 numpy view)."""
        nbytes = int(nbytes)
        addr = self._out_ptr(self._lib.setk_host_alloc, ctypes.c_size_t(nbytes))
        view = np.ctypeslib.as_array(ctypes.cast(addr, POINTER(ctypes.c_uint8)), shape=(nbytes,))
        return addr, view

    def host_free(self, addr):
        self.check(self._lib.setk_host_free(self._h, c_void_p(addr)))

    def stream_create(self):
        return self._out_ptr(self._lib.setk_stream_create)

    def stream_destroy(self, stream):
        self.check(self._lib.setk_stream_destroy(self._h, c_void_p(stream)))

    def stream_synchronize(self, stream):
        self.check(self._lib.setk_stream_synchronize(self._h, c_void_p(stream)))

    def stream_wait_event(self, stream, event):
        self.check(self._lib.setk_stream_wait_event(self._h, c_void_p(stream), c_void_p(event)))

    def event_create(self):
        return self._out_ptr(self._lib.setk_event_create)

    def event_destroy(self, event):
        self.check(self._lib.setk_event_destroy(self._h, c_void_p(event)))

    def event_record(self, event, stream):
        self.check(self._lib.setk_event_record(self._h, c_void_p(event), c_void_p(stream)))

    def event_synchronize(self, event):
        self.check(self._lib.setk_event_synchronize(self._h, c_void_p(event)))

    def memcpy_d2h_async(self, dst, src, nbytes, stream):
        self.check(self._lib.setk_memcpy_d2h_async(self._h, c_void_p(dst), c_void_p(src),
                                                   ctypes.c_size_t(int(nbytes)), c_void_p(stream)))

    def memcpy_h2d_async(self, dst, src, nbytes, stream):
        self.check(self._lib.setk_memcpy_h2d_async(self._h, c_void_p(int(dst)), c_void_p(int(src)),
                                                   int(nbytes), stream))

    # -- plan ---------------------------------------------------------------
    def stft_plan(self, frame_len, frame_hop, n_fft, center, window=None):
        key = (int(frame_len), int(frame_hop), int(n_fft), bool(center),
               None if window is None else window.tobytes())
        if self.plan == key:
            return
        wp = None
        if window is not None:
            window = np.ascontiguousarray(window, dtype=np.float32)
            if window.shape != (frame_len,):
                raise ValueError("window must have frame_len entries")
            wp = window.ctypes.data
        self.check(
            self._lib.setk_stft_plan(self._h, frame_len, frame_hop, n_fft,
                                     1 if center else 0, wp))
        self.plan = key

    def num_frames(self, num_samples):
        r = self._lib.setk_stft_num_frames(self._h, int(num_samples))
        if r < 0:
            self.check(r)
        return r

    def istft_num_samples(self, num_frames, nsamps=-1):
        r = self._lib.setk_istft_num_samples(self._h, int(num_frames),
                                             -1 if nsamps is None else int(nsamps))
        if r < 0:
            self.check(r)
        return r

    # -- modular operators (numpy in / numpy out, or device tensors) ---------
    def stft(self, audio, out, stream=None):
        C, N = audio.shape
        self.check(
            self._lib.setk_stft(self._h, _ptr(audio), C, N, _ptr(out),
                                current_stream_ptr() if stream is None else stream))

    def stft_batch(self, C, audio_ptrs, num_samples, spec_ptrs, stream=None, spec_pitch=0):
        """One launch for the spectrograms of a batch (device addresses)."""
        n = len(audio_ptrs)
        A = (c_void_p * n)(*audio_ptrs)
        S = (c_void_p * n)(*spec_ptrs)
        NS = (c_int * n)(*[int(v) for v in num_samples])
        self.check(self._lib.setk_stft_batch(self._h, n, int(C), A, NS, S, int(spec_pitch),
                                             current_stream_ptr() if stream is None else stream))

    def istft(self, spec, batch, num_frames, nsamps, norm, out, stream=None):
        self.check(
            self._lib.setk_istft(self._h, _ptr(spec), batch, num_frames,
                                 -1 if nsamps is None else int(nsamps), _ptr(norm),
                                 _ptr(out),
                                 current_stream_ptr() if stream is None else stream))

    def covar(self, spec, mask, C, T, F, out, stream=None):
        self.check(
            self._lib.setk_covar(self._h, _ptr(spec), _ptr(mask), C, T, F, _ptr(out),
                                 current_stream_ptr() if stream is None else stream))

    def pevd(self, Rs, Rn, F, C, flags, out, status, stream=None):
        self.check(
            self._lib.setk_pevd(self._h, _ptr(Rs), _ptr(Rn), F, C, flags, _ptr(out),
                                _ptr(status),
                                current_stream_ptr() if stream is None else stream))

    def weights(self, opts, Rs, Rn, Ry, F, C, out, status, stream=None):
        ref = c_int(-1)
        self.check(
            self._lib.setk_weights(self._h, ctypes.byref(opts), _ptr(Rs), _ptr(Rn),
                                   _ptr(Ry), F, C, _ptr(out), _ptr(status),
                                   ctypes.byref(ref),
                                   current_stream_ptr() if stream is None else stream))
        return ref.value

    def pcm16_to_float(self, pcm, C, N, out, stream=None):
        """interleaved int16 frames pcm[N][C] -> float32 out[C][N] (= pcm / 32768)."""
        self.check(
            self._lib.setk_pcm16_to_float(self._h, _ptr(pcm), C, N, _ptr(out),
                                          current_stream_ptr() if stream is None else stream))

    def float_to_pcm16(self, audio, C, N, pcm, stream=None):
        """float32 audio[C][N] -> interleaved int16 frames pcm[N][C] (rint(x * 32767), wrapping)."""
        self.check(
            self._lib.setk_float_to_pcm16(self._h, _ptr(audio), C, N, _ptr(pcm),
                                          current_stream_ptr() if stream is None else stream))

    def pcm16_to_float_batch(self, C, pcm_ptrs, num_samples, out_ptrs, power0=None, stream=None):
        """One launch for a batch of interleaved int16 payloads (device addresses)."""
        n = len(pcm_ptrs)
        P = (c_void_p * n)(*pcm_ptrs)
        O = (c_void_p * n)(*out_ptrs)
        NS = (c_int * n)(*[int(v) for v in num_samples])
        self.check(
            self._lib.setk_pcm16_to_float_batch(
                self._h, n, int(C), P, NS, O, _ptr(power0),
                current_stream_ptr() if stream is None else stream))

    def pcm16_channel_stride(self, num_samples):
        return int(self._lib.setk_pcm16_channel_stride(int(num_samples)))

    def pcm16_deinterleave_batch(self, C, pcm_ptrs, num_samples, out_ptrs, power0=None, stream=None):
        """Interleaved int16 frames -> planar int16 [C][pcm16_channel_stride(N)] (device
        addresses), ONE launch: the input layout of FLAG_IN_PCM16."""
        n = len(pcm_ptrs)
        P = (c_void_p * n)(*pcm_ptrs)
        O = (c_void_p * n)(*out_ptrs)
        NS = (c_int * n)(*[int(v) for v in num_samples])
        self.check(
            self._lib.setk_pcm16_deinterleave_batch(
                self._h, n, int(C), P, NS, O, _ptr(power0),
                current_stream_ptr() if stream is None else stream))

    def kaldi_cm_decode_batch(self, items, stream=None):
        """Kaldi CompressedMatrix bodies -> float32 matrices on the device, ONE launch.  items:
        (kind 'CM' | 'CM2' | 'CM3', vmin, vrange, rows, cols, transpose, src address, dst address);
        the bytes at src are what follows the archive's 16-byte global header (libs/kaldi_io.py
        `uncompress`: same float32 operations, same results bit for bit)."""
        n = len(items)
        code = {"CM": 1, "CM2": 2, "CM3": 3}
        K = (c_int * n)(*[code[it[0]] for it in items])
        VMIN = (c_float * n)(*[float(it[1]) for it in items])
        VRNG = (c_float * n)(*[float(it[2]) for it in items])
        R = (c_int * n)(*[int(it[3]) for it in items])
        Cc = (c_int * n)(*[int(it[4]) for it in items])
        TR = (c_int * n)(*[int(bool(it[5])) for it in items])
        S = (c_void_p * n)(*[int(it[6]) for it in items])
        D = (c_void_p * n)(*[int(it[7]) for it in items])
        self.check(self._lib.setk_kaldi_cm_decode_batch(
            self._h, n, K, VMIN, VRNG, R, Cc, TR, S, D,
            current_stream_ptr() if stream is None else stream))

    def ban(self, weight, Rn, F, C, out, stream=None):
        self.check(
            self._lib.setk_ban(self._h, _ptr(weight), _ptr(Rn), F, C, _ptr(out),
                               current_stream_ptr() if stream is None else stream))

    def rank1(self, Rs, Rn, F, C, out, status, stream=None):
        self.check(
            self._lib.setk_rank1(self._h, _ptr(Rs), _ptr(Rn), F, C, _ptr(out), _ptr(status),
                                 current_stream_ptr() if stream is None else stream))

    def beamform(self, weight, spec, C, T, F, out, stream=None):
        self.check(
            self._lib.setk_beamform(self._h, _ptr(weight), _ptr(spec), C, T, F,
                                    _ptr(out),
                                    current_stream_ptr() if stream is None else stream))

    def cgmm_masks(self, spec, C, T, F, num_iters, init_mask, gamma_out, mask_out, stream=None,
                   update_alpha=False):
        self.check(
            self._lib.setk_cgmm_masks(self._h, _ptr(spec), C, T, F, int(num_iters),
                                      _ptr(init_mask), _ptr(gamma_out), _ptr(mask_out),
                                      CGMM_UPDATE_ALPHA if update_alpha else 0,
                                      current_stream_ptr() if stream is None else stream))

    def cgmm_masks_k(self, spec, C, T, F, K, num_iters, gamma0, init_mask, gamma_out, stream=None,
                     update_alpha=False, status=None):
        """General CGMM (K <= 4 classes, C <= 16 channels): spec [C][T][F] complex64, gamma0
        [K][F][T] float64 or None (K = 2: init_mask [T][F] or the deterministic start),
        gamma_out [K][T][F] float32; status int32[F] (numpy or device address) or None: SETK_NUM_*
        per bin (non-finite covariance, Jacobi sweep limit -- numpy.linalg.eigh's LinAlgError)."""
        self.check(
            self._lib.setk_cgmm_masks_k_status(self._h, _ptr(spec), int(C), int(T), int(F), int(K),
                                               int(num_iters), _ptr(gamma0), _ptr(init_mask),
                                               _ptr(gamma_out), CGMM_UPDATE_ALPHA if update_alpha else 0,
                                               _ptr(status), current_stream_ptr() if stream is None else stream))

    def cgmm_masks_batch(self, C, spec_ptrs, num_frames, F, num_iters, init_ptrs, out_ptrs,
                         stream=None, update_alpha=False, spec_pitch=0):
        n = len(spec_ptrs)
        S = (c_void_p * n)(*spec_ptrs)
        O = (c_void_p * n)(*out_ptrs)
        I = (c_void_p * n)(*init_ptrs) if init_ptrs is not None else None
        T = (c_int * n)(*[int(v) for v in num_frames])
        self.check(
            self._lib.setk_cgmm_masks_batch(self._h, n, int(C), S, T, int(F), int(num_iters), I, O,
                                            CGMM_UPDATE_ALPHA if update_alpha else 0, int(spec_pitch),
                                            current_stream_ptr() if stream is None else stream))

    def cgmm_estimate_batch(self, C, audio_ptrs, num_samples, num_iters, init_ptrs, out_ptrs,
                            stream=None, update_alpha=False):
        """audio in, masks out: STFT straight into the bin-major layout + bin-resident EM.
        Raises SetkUnsupported when a bin of the longest utterance does not fit a CU."""
        n = len(audio_ptrs)
        A = (c_void_p * n)(*audio_ptrs)
        O = (c_void_p * n)(*out_ptrs)
        I = (c_void_p * n)(*init_ptrs) if init_ptrs is not None else None
        NS = (c_int * n)(*[int(v) for v in num_samples])
        self.check(
            self._lib.setk_cgmm_estimate_batch(self._h, n, int(C), A, NS, int(num_iters), I, O,
                                               CGMM_UPDATE_ALPHA if update_alpha else 0,
                                               current_stream_ptr() if stream is None else stream))

    # -- fused hot path ---------------------------------------------------------
    def enhance_batch(self, opts, num_channels, audio_ptrs, num_samples, mask_ptrs,
                      itf_ptrs, wave_ptrs, want_status=True, stream=None, taps=None,
                      status_ptr=None):
        """taps: dict with any of Rs, Rn ([n][F][C][C] complex64), weight ([n][F][C]
        complex64), maxabs ([n] float32) -> numpy arrays / device tensors filled by
        setk_enhance_batch_taps."""
        n = len(audio_ptrs)
        A = (c_void_p * n)(*audio_ptrs)
        M = (c_void_p * n)(*mask_ptrs)
        W = (c_void_p * n)(*wave_ptrs)
        I = (c_void_p * n)(*itf_ptrs) if itf_ptrs is not None else None
        NS = (c_int * n)(*[int(v) for v in num_samples])
        ST = (c_int * n)() if want_status else None
        if status_ptr is not None:
            # device int32[n]: filled asynchronously on the stream, nothing returned
            ST = ctypes.cast(c_void_p(int(status_ptr)), POINTER(c_int))
            want_status = False
        st = current_stream_ptr() if stream is None else stream
        if taps:
            tp = BatchTaps(*[_ptr(taps.get(k)) for k in ("Rs", "Rn", "weight", "maxabs")])
            self.check(
                self._lib.setk_enhance_batch_taps(
                    self._h, ctypes.byref(opts), n, int(num_channels), A, NS, M, I, W, ST,
                    ctypes.byref(tp), st))
        else:
            self.check(
                self._lib.setk_enhance_batch(
                    self._h, ctypes.byref(opts), n, int(num_channels), A, NS, M, I, W, ST, st))
        return list(ST) if want_status else None

    def directional_feats(self, spec, sv, pairs, C, T, F, out, stream=None):
        """spec [C][T][F], sv [F][C] complex64, pairs [(i, j), ...] -> out [T][F] float32."""
        flat = [int(v) for p in pairs for v in p]
        P = (c_int * len(flat))(*flat)
        self.check(
            self._lib.setk_directional_feats(self._h, _ptr(spec), _ptr(sv), P, len(flat) // 2, C, T,
                                             F, _ptr(out),
                                             current_stream_ptr() if stream is None else stream))

    def apply_weights_batch(self, num_channels, audio_ptrs, num_samples, weights, n_sets,
                            weight_index, wave_ptrs, flags=0, stream=None):
        """Fixed beamformer + iSTFT + renorm for a batch; weights [n_sets][257][C]
        complex64 (numpy or device tensor), weight_index: list of ints or None."""
        n = len(audio_ptrs)
        A = (c_void_p * n)(*audio_ptrs)
        W = (c_void_p * n)(*wave_ptrs)
        NS = (c_int * n)(*[int(v) for v in num_samples])
        IDX = (c_int * n)(*[int(v) for v in weight_index]) if weight_index is not None else None
        self.check(
            self._lib.setk_apply_weights_batch(
                self._h, n, int(num_channels), A, NS, _ptr(weights), int(n_sets), IDX, W,
                int(flags), current_stream_ptr() if stream is None else stream))

    def wpe(self, spec, C, T, F, taps, delay, context, num_iters, out, lambda_enh=None,
            inv_lambda_out=None, status=None, stream=None):
        """spec / out [C][T][F] complex64; status int32[F] (numpy) or None."""
        self.check(
            self._lib.setk_wpe(self._h, _ptr(spec), int(C), int(T), int(F), int(taps), int(delay),
                               int(context), int(num_iters), _ptr(lambda_enh), _ptr(out),
                               _ptr(inv_lambda_out), _ptr(status),
                               current_stream_ptr() if stream is None else stream))

    def wpe_batch(self, specs, C, frames, F, taps, delay, context, num_iters, outs, status=None,
                  stream=None):
        """specs / outs: lists of [C][T_u][F] complex64 arrays (numpy or device tensors);
        status int32 [n][F] (numpy) or None."""
        n = len(specs)
        sp = (c_void_p * n)(*[_ptr(a) for a in specs])
        op = (c_void_p * n)(*[_ptr(a) for a in outs])
        fr = (c_int * n)(*[int(t) for t in frames])
        self.check(
            self._lib.setk_wpe_batch(self._h, n, sp, int(C), fr, int(F), int(taps), int(delay),
                                     int(context), int(num_iters), op, _ptr(status),
                                     current_stream_ptr() if stream is None else stream))

    def wpe_batch_fnt(self, specs, C, frames, F, taps, delay, context, num_iters, outs, status=None,
                      stream=None):
        """wpe_batch with specs / outs in the reference's layout F x N x T_u (complex64)."""
        n = len(specs)
        sp = (c_void_p * n)(*[_ptr(a) for a in specs])
        op = (c_void_p * n)(*[_ptr(a) for a in outs])
        fr = (c_int * n)(*[int(t) for t in frames])
        self.check(
            self._lib.setk_wpe_batch_fnt(self._h, n, sp, int(C), fr, int(F), int(taps), int(delay),
                                         int(context), int(num_iters), op, _ptr(status),
                                         current_stream_ptr() if stream is None else stream))

    def wpe_batch_var(self, specs, C, frames, F, taps, delay, context, num_iters, outs, lambda_enh=None,
                      inv_lambda_outs=None, status=None, stream=None):
        """wpe_batch with facted_wpd's per-utterance operands: lambda_enh[u] ([T_u][F] complex64 or
        None) gives the variances of iteration 0, inv_lambda_outs[u] ([T_u][F] float32) receives
        1 / lambda of the last one.  Arrays or device addresses; status int32 [n][F] or None."""
        n = len(specs)
        sp = (c_void_p * n)(*[_ptr(a) for a in specs])
        op = (c_void_p * n)(*[_ptr(a) for a in outs])
        le = (c_void_p * n)(*[_ptr(a) for a in lambda_enh]) if lambda_enh is not None else None
        il = (c_void_p * n)(*[_ptr(a) for a in inv_lambda_outs]) if inv_lambda_outs is not None else None
        fr = (c_int * n)(*[int(t) for t in frames])
        self.check(
            self._lib.setk_wpe_batch_var(self._h, n, sp, int(C), fr, int(F), int(taps), int(delay),
                                         int(context), int(num_iters), le, op, il, _ptr(status),
                                         current_stream_ptr() if stream is None else stream))

    def wpe_step(self, spec, C, T, F, taps, delay, lambda_ft, out, status=None, stream=None):
        """One wpe_step with the caller's variances (float64 F x T) used as given."""
        self.check(
            self._lib.setk_wpe_step(self._h, _ptr(spec), int(C), int(T), int(F), int(taps), int(delay),
                                    _ptr(lambda_ft), _ptr(out), _ptr(status),
                                    current_stream_ptr() if stream is None else stream))

    def set_profiling(self, on):
        self.check(self._lib.setk_set_profiling(self._h, 1 if on else 0))

    def last_stage_ms(self):
        out = (c_float * 4)()
        self.check(self._lib.setk_last_stage_ms(self._h, out))
        return list(out)


_default_ctx = {}


def default_context(device=None):
    """Process-wide context for the python API mirror (one per device)."""
    if device is None:
        # SETK_DEVICE, then LOCAL_RANK (torchrun), then torch's current device
        if os.environ.get("SETK_DEVICE") is not None:
            device = int(os.environ["SETK_DEVICE"])
        elif os.environ.get("LOCAL_RANK") is not None:
            device = int(os.environ["LOCAL_RANK"])
        else:
            device = 0
            torch = _torch()
            try:
                if torch is not None and torch.cuda.is_available():
                    device = torch.cuda.current_device()
            except Exception:
                pass
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
