"""Drives every entry point of include/setk_hip.h through the sanitizer build of the host side
(tools/hoststub/build.sh: AddressSanitizer + UBSan, HIP replaced by a host-memory stand-in
whose kernel launches only validate their configuration).  Run by tests/test_host_asan.py:

    LD_PRELOAD=<libclang_rt.asan> SETK_ALLOW_HOSTSTUB=1 SETK_LIB=_abl/libsetk_hostasan.so python tests/host_asan_driver.py

Outputs are not looked at (no kernel runs); what is checked is the host code: argument
validation, descriptor tables, arena sizing, staging copies, launch geometry.  Prints one
JSON line with the stand-in's counters; an ASAN / UBSan report aborts the process."""
import ctypes
import json
import os
import sys
from ctypes import byref, c_char, c_int, c_long, c_size_t, c_void_p

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from setk_amd import _ffi  # noqa: E402

lib = _ffi.load_library()
assert hasattr(lib, "hoststub_report"), "not the host-stub build: set SETK_LIB"
lib.hipMalloc.argtypes = [ctypes.POINTER(c_void_p), c_size_t]
lib.hipFree.argtypes = [c_void_p]
lib.hipMemcpy.argtypes = [c_void_p, c_void_p, c_size_t, c_int]
F = 257
rng = np.random.default_rng(0)
held = []


def dmalloc(nbytes):
    p = c_void_p()
    assert lib.hipMalloc(byref(p), max(int(nbytes), 16)) == 0
    held.append(p.value)
    return p.value


def to_dev(a):
    a = np.ascontiguousarray(a)
    p = dmalloc(a.nbytes)
    assert lib.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0
    return p


def release():
    while held:
        assert lib.hipFree(held.pop()) == 0


def cplx(*shape):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)


def expect(exc, fn, *a, **k):
    try:
        fn(*a, **k)
    except exc:
        return
    raise AssertionError(f"{fn.__name__} did not raise {exc.__name__}")


def operators(ctx, C, T):
    spec = cplx(C, T, F)
    mask = rng.random((T, F)).astype(np.float32)
    R = np.empty((F, C, C), np.complex64)
    ctx.covar(spec, mask, C, T, F, R)
    Rs = cplx(F, C, C)
    Rn = cplx(F, C, C)
    st = np.zeros(F, np.int32)
    sv = np.empty((F, C), np.complex64)
    ctx.pevd(Rs, Rn, F, C, 0, sv, st)
    ctx.pevd(Rs, None, F, C, 0, sv, st)
    for kind in range(4):
        opts = _ffi.BfOpts()
        opts.kind = kind
        opts.pmwf_ref = -1
        w = np.empty((F, C), np.complex64)
        try:
            ctx.weights(opts, Rs, Rn, Rs, F, C, w, st)
        except (ValueError, NotImplementedError):
            pass
    w = cplx(F, C)
    ctx.ban(w, Rn, F, C, np.empty((F, C), np.complex64))
    ctx.rank1(Rs, Rn, F, C, np.empty((F, C, C), np.complex64), st)
    ctx.beamform(w, spec, C, T, F, np.empty((T, F), np.complex64))
    if C >= 2:
        pairs = [(i, j) for i in range(C) for j in range(i + 1, C)]
        ctx.directional_feats(spec, w, pairs, C, T, F, np.empty((T, F), np.float32))
    # the same through device pointers
    d_spec, d_mask, d_R = to_dev(spec), to_dev(mask), dmalloc(R.nbytes)
    ctx.covar(d_spec, d_mask, C, T, F, d_R)
    ctx.beamform(to_dev(w), d_spec, C, T, F, dmalloc(T * F * 8))
    release()


def stft_paths(ctx, C):
    for (fl, hop, nfft, center) in ((512, 256, 512, True), (512, 128, 512, False), (400, 160, 512, True),
                                    (400, 160, 400, True), (1024, 256, 1024, True)):
        ctx.stft_plan(fl, hop, nfft, center)
        N = 5 * fl + 37
        T = ctx.num_frames(N)
        nb = nfft // 2 + 1
        audio = rng.standard_normal((C, N)).astype(np.float32)
        spec = np.empty((C, T, nb), np.complex64)
        ctx.stft(audio, spec)
        ns = ctx.istft_num_samples(T, N)
        ctx.istft(spec, C, T, N, None, np.empty((C, ns), np.float32))
        ctx.istft(spec, C, T, N, np.ones(C, np.float32), np.empty((C, ns), np.float32))
    ctx.stft_plan(512, 256, 512, True)
    expect(ValueError, ctx.stft, np.zeros((C, 10), np.float32)[:, :0].copy(), np.empty((C, 0, F), np.complex64))


def fused(ctx, C, lens, kind):
    ctx.stft_plan(512, 256, 512, True)
    n = len(lens)
    audio = [to_dev(rng.standard_normal((C, N)).astype(np.float32)) for N in lens]
    frames = [ctx.num_frames(N) for N in lens]
    masks = [to_dev(rng.random((t, F)).astype(np.float32)) for t in frames]
    waves = [dmalloc(N * 4) for N in lens]
    opts = _ffi.BfOpts()
    opts.kind = kind
    opts.pmwf_ref = -1
    try:
        ctx.enhance_batch(opts, C, audio, lens, masks, None, waves)
        taps = {"Rs": np.empty((n, F, C, C), np.complex64), "Rn": np.empty((n, F, C, C), np.complex64),
                "weight": np.empty((n, F, C), np.complex64), "maxabs": np.empty(n, np.float32)}
        ctx.enhance_batch(opts, C, audio, lens, masks, masks, waves, taps=taps)
    except NotImplementedError:
        pass
    wts = cplx(2, F, C)
    try:
        for flags in (0, _ffi.FLAG_NO_RENORM):
            ctx.apply_weights_batch(C, audio, lens, wts, 2, [u % 2 for u in range(n)], waves, flags=flags)
        ctx.apply_weights_batch(C, audio, lens, to_dev(wts), 2, None, waves)
    except NotImplementedError:
        assert C > 8
    pcm = [to_dev(rng.integers(-3000, 3000, (N, C)).astype(np.int16)) for N in lens]
    ctx.pcm16_to_float_batch(C, pcm, lens, audio, power0=dmalloc(4 * n * F * max(frames)))
    ctx.pcm16_to_float(rng.integers(-9, 9, (lens[0], C)).astype(np.int16), C, lens[0],
                       np.empty((C, lens[0]), np.float32))
    ctx.float_to_pcm16(rng.standard_normal((C, lens[0])).astype(np.float32), C, lens[0],
                       np.empty((lens[0], C), np.int16))
    ctx.float_to_pcm16(audio[0], C, lens[0], dmalloc(2 * C * lens[0]))
    # Kaldi compressed-matrix bodies (the stand-in runs no kernels: the host side of the call)
    ctx.kaldi_cm_decode_batch([("CM", -1.0, 2.0, 7, 5, False, dmalloc(5 * (8 + 7)), dmalloc(4 * 35)),
                               ("CM2", 0.0, 1.0, 7, 5, True, dmalloc(2 * 35), dmalloc(4 * 35)),
                               ("CM3", 0.0, 1.0, 7, 5, False, dmalloc(35), dmalloc(4 * 35))])
    try:
        ctx.kaldi_cm_decode_batch([("CM3", 0.0, 1.0, 0, 5, False, dmalloc(35), dmalloc(4 * 35))])
        raise AssertionError("an empty matrix must be refused")
    except ValueError:
        pass
    specs = [dmalloc(C * t * F * 8) for t in frames]
    try:
        ctx.stft_batch(C, audio, lens, specs)
    except NotImplementedError:
        assert C > 8
    release()


def cgmm(ctx, C, lens, init, alpha):
    ctx.stft_plan(512, 256, 512, True)
    n = len(lens)
    frames = [ctx.num_frames(N) for N in lens]
    audio = [to_dev(rng.standard_normal((C, N)).astype(np.float32)) for N in lens]
    specs = [to_dev(cplx(C, t, F)) for t in frames]
    inits = [to_dev(rng.random((t, F)).astype(np.float32)) for t in frames] if init else None
    outs = [dmalloc(t * F * 4) for t in frames]
    for env in ("", "1"):
        os.environ["SETK_CGMM_STREAMING"] = env
        if not env:
            del os.environ["SETK_CGMM_STREAMING"]
        ctx.cgmm_masks_batch(C, specs, frames, F, 3, inits, outs, update_alpha=alpha)
    try:
        ctx.cgmm_estimate_batch(C, audio, lens, 3, inits, outs, update_alpha=alpha)
    except NotImplementedError:
        pass
    t = frames[0]
    ctx.cgmm_masks(cplx(C, t, F), C, t, F, 2, None, np.empty((2, t, F), np.float32),
                   np.empty((t, F), np.float32), update_alpha=alpha)
    release()


def cgmm_general(ctx, C, T, K, init):
    """setk_cgmm_masks_k: K classes from a K x F x T float64 start, or K = 2 from a mask / the
    deterministic start (host and device pointers)."""
    spec = cplx(C, T, F)
    gamma = np.empty((K, T, F), np.float32)
    g0 = None
    if K > 2 or init == "gamma0":
        g0 = rng.random((K, F, T))
        g0 /= g0.sum(0, keepdims=True)
    mask = rng.random((T, F)).astype(np.float32) if init == "mask" else None
    ctx.cgmm_masks_k(spec, C, T, F, K, 2, g0, mask, gamma, update_alpha=(K == 3))
    ctx.cgmm_masks_k(to_dev(spec), C, T, F, K, 1, None if g0 is None else to_dev(g0),
                     None if mask is None else to_dev(mask), dmalloc(gamma.nbytes))
    expect((ValueError, NotImplementedError), ctx.cgmm_masks_k, spec, C, T, F, 5, 1, None, None, gamma)
    expect((ValueError, NotImplementedError), ctx.cgmm_masks_k, spec, C, T, F, 3, 1, None, None, gamma)
    release()


def wpe(ctx, C, T, taps, delay):
    spec = cplx(C, T, F)
    out = np.empty_like(spec)
    st = np.zeros(F, np.int32)
    try:
        ctx.wpe(spec, C, T, F, taps, delay, 1, 2, out, status=st)
        ctx.wpe(spec, C, T, F, taps, delay, 0, 1, out, lambda_enh=cplx(T, F),
                inv_lambda_out=np.empty((F, T), np.float32), status=st)
        ctx.wpe_step(spec, C, T, F, taps, delay, rng.random((F, T)) + 0.1, out, status=st)
        fnt = cplx(F, C, T)
        ctx.wpe_batch_fnt([fnt, fnt[:, :, :T // 2].copy()], C, [T, T // 2], F, taps, delay, 1, 3,
                          [np.empty_like(fnt), np.empty((F, C, T // 2), np.complex64)],
                          status=np.zeros((2, F), np.int32))
        ctx.wpe_batch([spec, spec[:, :T // 2].copy()], C, [T, T // 2], F, taps, delay, 1, 2,
                      [out, np.empty((C, T // 2, F), np.complex64)], status=np.zeros((2, F), np.int32))
    except NotImplementedError:
        pass


def runtime_helpers(ctx):
    """setk_device_alloc / host_alloc / streams / events: what the streaming pipeline runs on."""
    d = ctx.device_alloc(4096)
    h, view = ctx.host_alloc(4096)
    view[:] = 7
    s1, s2 = ctx.stream_create(), ctx.stream_create()
    e = ctx.event_create()
    ctx.memcpy_h2d_async(d, h, 4096, s1)
    ctx.event_record(e, s1)
    ctx.stream_wait_event(s2, e)
    view[:] = 0
    ctx.memcpy_d2h_async(h, d, 4096, s2)
    ctx.event_synchronize(e)
    ctx.stream_synchronize(s2)
    assert int(view.sum()) == 7 * 4096
    ctx.event_destroy(e)
    ctx.stream_destroy(s1)
    ctx.stream_destroy(s2)
    ctx.host_free(h)
    ctx.device_free(d)
    expect(ValueError, ctx.device_alloc, 0)


def edge_geometries(ctx):
    """Tiny and awkward lengths through the fused path, the fixed-weights path, the CGMM
    estimator and the stand-alone transforms: every call must end in success or in an error
    code -- never in an empty launch grid (the stand-in counts those) or a bad table."""
    F_ = 257
    for hop in (64, 128, 200, 256, 512):
        for center in (True, False):
            ctx.stft_plan(512, hop, 512, center)
            for N in (1, 2, 255, 256, 257, 511, 512, 513, 767, 1024, 1025):
                try:
                    T = ctx.num_frames(N)
                except ValueError:  # shorter than the reflect padding / than one frame
                    T = 0
                for C in (1, 3, 8):
                    if T <= 0:
                        expect((ValueError, NotImplementedError), ctx.stft,
                               np.zeros((C, N), np.float32), np.empty((C, 1, F_), np.complex64))
                        continue
                    a = to_dev(rng.standard_normal((C, N)).astype(np.float32))
                    m = to_dev(rng.random((T, F_)).astype(np.float32))
                    L = max(ctx.istft_num_samples(T), 1)
                    w = dmalloc(4 * L)
                    opts = _ffi.BfOpts()
                    opts.kind = 0
                    opts.pmwf_ref = -1
                    try:
                        ctx.enhance_batch(opts, C, [a], [N], [m], None, [w])
                        ctx.apply_weights_batch(C, [a], [N], cplx(1, F_, C), 1, None, [w])
                        ctx.cgmm_estimate_batch(C, [a], [N], 2, None, [dmalloc(4 * T * F_)])
                    except (ValueError, NotImplementedError):
                        pass
                    spec = np.empty((C, T, F_), np.complex64)
                    ctx.stft(rng.standard_normal((C, N)).astype(np.float32), spec)
                    ctx.istft(spec, C, T, None, None, np.empty((C, L), np.float32))
                    release()
    ctx.stft_plan(512, 256, 512, True)


def main():
    ctx = _ffi.Context(0)
    runtime_helpers(ctx)
    edge_geometries(ctx)
    if "--selftest-overflow" in sys.argv:
        # the sanitizer must be live: an output buffer one matrix short -> heap-buffer-overflow
        ctx.covar(cplx(2, 8, F), np.ones((8, F), np.float32), 2, 8, F, np.empty((F - 1, 2, 2), np.complex64))
        return 0
    if "--selftest-device-deref" in sys.argv:
        # host code touching "device" memory is a report too (the stand-in keeps it poisoned)
        p = dmalloc(64)
        return int(ctypes.cast(p, ctypes.POINTER(c_int))[0] == 0) * 0 + int(ctypes.string_at(p, 4) == b"x")
    ctx.stft_plan(512, 256, 512, True)
    for C in (1, 2, 3, 4, 6, 8, 12, 16):
        operators(ctx, C, 40 + C)
        stft_paths(ctx, min(C, 4))
    for C, lens, kind in ((8, [480000] * 2, 0), (4, [160000, 1000, 64000, 777], 1), (1, [4096], 0),
                          (6, [48000, 31999, 52001], 2), (2, [600], 3), (12, [20000, 30000], 0),
                          (16, [16000], 1), (8, [16000 + 977 * u for u in range(300)], 0)):
        fused(ctx, C, lens, kind)
    for C, lens, init, alpha in ((6, [480000, 320000], False, False), (2, [16000, 9000, 48000], True, True),
                                 (8, [480000], True, False), (4, [160000] * 3, False, True),
                                 (3, [1200000], False, False), (5, [700], False, False),
                                 (7, [100000], False, False), (6, [4000000], False, False)):
        cgmm(ctx, C, lens, init, alpha)
    for C, T, K, init in ((4, 70, 3, ""), (2, 33, 4, ""), (12, 50, 2, ""), (16, 40, 2, "mask"), (9, 64, 2, "gamma0"),
                          (1, 20, 2, "")):
        cgmm_general(ctx, C, T, K, init)
    # (the last three: R beyond LDS, factored in the arena's global scratch)
    for C, T, taps, delay in ((2, 200, 10, 3), (6, 300, 10, 3), (8, 120, 5, 1), (1, 64, 3, 0), (4, 50, 20, 3),
                              (8, 150, 12, 3), (16, 90, 6, 2), (16, 40, 16, 0)):
        wpe(ctx, C, T, taps, delay)
    assert ":" in ctx.pci_bus_id()
    # argument errors must come back as error codes, not crashes
    expect((ValueError, NotImplementedError), ctx.covar, cplx(2, 4, F), np.ones((4, F), np.float32), 0, 4, F,
           np.empty((F, 2, 2), np.complex64))
    expect((ValueError, NotImplementedError), ctx.covar, cplx(2, 4, F), np.ones((4, F), np.float32), 40, 4, F,
           np.empty((F, 2, 2), np.complex64))
    expect(ValueError, ctx.stft_plan, 512, 0, 512, True)
    expect((ValueError, NotImplementedError), ctx.stft_plan, 512, 256, 256, True)
    ctx.close()
    vals = [c_long() for _ in range(4)]
    first = (c_char * 512)()
    lib.hoststub_report(*[byref(v) for v in vals], first, 512)
    rec = dict(zip(("launches", "violations", "copies", "live_allocations"), [v.value for v in vals]))
    rec["first_violation"] = first.value.decode()
    print(json.dumps(rec))
    return 1 if rec["violations"] else 0


if __name__ == "__main__":
    sys.exit(main())
