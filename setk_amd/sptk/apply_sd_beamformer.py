#!/usr/bin/env python
"""Drop-in for funcwj/setk scripts/sptk/apply_sd_beamformer.py: apply_classic_beamformer
with --beamformer sd (:14-17)."""
from setk_amd.sptk.apply_classic_beamformer import build_parser, run as run_classic_beamformer


def run(args):
    args.beamformer = "sd"
    run_classic_beamformer(args)


def main(argv=None):
    parser = build_parser(
        description="Command to apply supperdirective beamformer (linear & circular array).",
        with_beamformer=False)
    run(parser.parse_args(argv))


if __name__ == "__main__":
    main()
