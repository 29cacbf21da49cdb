#!/usr/bin/env python
"""Same path and command line as funcwj/setk's scripts/sptk/apply_classic_beamformer.py;
the computation runs on the MI355X (setk_amd/sptk/apply_classic_beamformer.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from setk_amd.sptk.apply_classic_beamformer import main  # noqa: E402

if __name__ == "__main__":
    main()
