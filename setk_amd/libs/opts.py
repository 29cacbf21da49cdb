"""
Shared STFT command line options of the sptk tools (the parent parser the
reference defines in scripts/sptk/libs/opts.py:21-49: --frame-len 512,
--frame-hop 256, --center true, --round-power-of-two true, --window hann).
"""
import argparse

_TRUE = {"y", "yes", "t", "true", "on", "1"}
_FALSE = {"n", "no", "f", "false", "off", "0"}


def strtobool(val):
    """distutils.util.strtobool semantics (distutils is gone in python 3.12)."""
    v = str(val).lower()
    if v in _TRUE:
        return 1
    if v in _FALSE:
        return 0
    raise ValueError(f"invalid truth value {val!r}")


def str2tuple(string, sep=","):
    """ "1.0,2,0" -> (1.0, 2.0, 0.0) """
    return tuple(float(tok) for tok in string.split(sep))


_STFT_OPTIONS = (
    ("--frame-len", int, 512, "Frame length in number of samples (related to sample frequency)"),
    ("--frame-hop", int, 256, "Frame shift in number of samples (related to sample frequency)"),
    ("--center", strtobool, True, "Value of parameter 'center' in librosa.stft functions"),
    ("--round-power-of-two", strtobool, True, "If true, pad fft size to power of two"),
    ("--window", str, "hann", "Type of window function, see scipy.signal.get_window"),
)


def _build():
    parser = argparse.ArgumentParser(add_help=False)
    for flag, typ, default, text in _STFT_OPTIONS:
        parser.add_argument(flag, type=typ, default=default, help=text)
    return parser


class StftParser(object):
    """argparse parent: ``ArgumentParser(parents=[StftParser.parser])``."""
    parser = _build()
