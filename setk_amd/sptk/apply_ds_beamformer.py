#!/usr/bin/env python
"""Drop-in for funcwj/setk scripts/sptk/apply_ds_beamformer.py: apply_classic_beamformer
with --beamformer ds (:14-17)."""
from setk_amd.sptk.apply_classic_beamformer import build_parser, run as run_classic_beamformer


def run(args):
    args.beamformer = "ds"
    run_classic_beamformer(args)


def main(argv=None):
    parser = build_parser(
        description="Command to apply delay and sum beamformer (linear & circular array).",
        with_beamformer=False)
    run(parser.parse_args(argv))


if __name__ == "__main__":
    main()
