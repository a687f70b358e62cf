"""Stress check of the SPD inverse on the GPU: for a ladder of sizes (single-block groups, groups of two, three and four
blocks; even and odd block counts; ragged last blocks) the inverse must be bit-identical run to run -- the persistent
sweep kernel orders the updates of every tile by flags, so a missing dependency would show up as run-to-run differences
-- symmetric, and satisfy A X v = v on random probes.

    python tools/stress_inverse.py [--repeat 3]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussdca.jl_amd as g  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--sizes", type=int, nargs="*",
                    default=[100, 128, 129, 384, 1000, 2560, 5000, 7424, 8320, 8448, 8576, 9000, 10000, 10112, 10752, 11000, 11600])
    args = ap.parse_args()
    ctx = g.Context(0)
    rng = np.random.default_rng(11)
    bad = 0
    for n in args.sizes:
        B = rng.standard_normal((n, 48))
        A = (B @ B.T) / 48 + np.diag(0.4 + rng.random(n))
        V = rng.standard_normal((n, 4))
        t = time.time()
        X0 = g.inv_cholesky(A, ctx=ctx)
        resid = float(np.max(np.abs(A @ (X0 @ V) - V)) / np.max(np.abs(V)))
        same = True
        for _ in range(args.repeat - 1):
            same = same and np.array_equal(g.inv_cholesky(A, ctx=ctx), X0)
        ok = same and resid < 1e-9 and np.array_equal(X0, X0.T)
        bad += not ok
        print("n=%6d blocks=%3d  residual %.2e  bit-identical over %d runs: %s  symmetric: %s  (%.1fs)%s" %
              (n, -(-n // 128), resid, args.repeat, same, np.array_equal(X0, X0.T), time.time() - t, "" if ok else "  <-- FAIL"),
              flush=True)
    print("FAILED" if bad else "all ok")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
