"""
Host-side mirror of the reference's ``scripts/sptk/libs`` call surface for the
mask-based adaptive-beamformer path.  Names, argument meaning, array layouts and
error behaviour follow the reference (file:line cited per function); the
arithmetic runs in libsetk_hip.so on the MI355X -- there is no CPU fallback.
"""
