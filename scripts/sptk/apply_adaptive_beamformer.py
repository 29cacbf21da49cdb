#!/usr/bin/env python
# coding=utf-8
"""Same path, same command line as funcwj/setk's
scripts/sptk/apply_adaptive_beamformer.py; the work is done on the MI355X by
setk_amd (see setk_amd/sptk/apply_adaptive_beamformer.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from setk_amd.sptk.apply_adaptive_beamformer import main  # noqa: E402

if __name__ == "__main__":
    main()
