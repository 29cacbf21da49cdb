#!/usr/bin/env python
"""Summarise a SETK_CGMM_TIMING dump (csrc/capi.hip): mean shader cycles per pass and bin of
wave 0, per phase.  python tools/cgmm_phases.py <dump> [<dump of the product build>]"""
import sys

import numpy as np

NAMES = {0: "frames pass (all phases)", 2: "barrier wait after the frames", 3: "solve (wave 0 = class 0)",
         4: "barrier wait after the solve", 8: "E0  q_0 = |L_0^-1 x|^2, 8 frames", 9: "E1  q_1",
         10: "P   log2 / exp2 / rcp posterior + weights",
         11: "A1  weighted outer products, first row group (plain build: R, the 21 real sums x 2 classes)",
         12: "A1  halving butterfly + row store",
         13: "A2  second row group (plain build: I, the 15 imaginary sums x 2 classes)",
         14: "A2  butterfly (+ the two posterior sums) + store", 16: "solve: float64 sums of the 4 wave rows",
         17: "solve: trace, power-of-two scale", 18: "solve: Cholesky on lanes (6 LDS round trips)",
         19: "solve: L^-1 columns, eigenvalue-floor certificate", 20: "solve: log det, factor write (fast path)",
         21: "solve: exact path (Jacobi), when taken"}


def load(path):
    a = np.loadtxt(path, comments="#")
    return a[:, 1:]


def main():
    a = load(sys.argv[1])
    passes = a[:, 5].mean()
    print(f"# {sys.argv[1]}: {a.shape[0]} bins, {passes:.0f} timed passes per bin "
          f"(init + EM iterations; the closing posterior pass is not timed), fast-path solves per class "
          f"{a[:, 6].mean():.1f} / {a[:, 7].mean():.1f}")
    tot = sum(a[:, k].mean() for k in (0, 2, 3, 4)) / passes
    print(f"| phase | cycles per pass | share of the pass |\n|---|---|---|")
    for k in sorted(NAMES):
        if k >= a.shape[1]:
            continue
        v = a[:, k].mean() / passes
        if v == 0:
            continue
        print(f"| {k:2d} {NAMES[k]} | {v:8.0f} | {100 * v / tot:5.1f} % |")
    print(f"| pass of one workgroup (0 + 2 + 3 + 4) | {tot:8.0f} | 100 % |")
    if len(sys.argv) > 2:
        b = load(sys.argv[2])
        pb = b[:, 5].mean()
        print(f"\n# product build (no clock reads inside a pass), {sys.argv[2]}: frames {b[:, 0].mean() / pb:.0f}, "
              f"barrier {b[:, 2].mean() / pb:.0f}, solve {b[:, 3].mean() / pb:.0f}, tail {b[:, 4].mean() / pb:.0f} "
              f"cycles per pass")


if __name__ == "__main__":
    main()
