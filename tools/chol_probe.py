"""The blocked Cholesky fallback (option CHOLESKY=2: every inverse goes through it) against LAPACK's potrf + potri:
agreement, the index reported for matrices that are not positive definite, time per inverse; then the families of small
pseudocounts where the sweep gives up (status and scores next to the oracle).

    python tools/chol_probe.py [--sizes 100 128 300 1000 2689 8600] [--families]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import gaussdca.jl_amd as g
from oracle import gdca_oracle as o

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", nargs="*", type=int, default=[100, 128, 129, 300, 1000, 2689, 8600])
ap.add_argument("--families", action="store_true")
args = ap.parse_args()
ctx = g.Context(0)
ok = True
rng = np.random.default_rng(5)
for n in args.sizes:
    A = rng.standard_normal((n, n + 50))
    C = A @ A.T / (n + 50) + 0.05 * np.eye(n)
    X_l = o.spd_inverse(C)
    ctx.set_option("CHOLESKY", 2)
    t0 = time.perf_counter()
    X_c = g.inv_cholesky(C, ctx=ctx)
    t1 = time.perf_counter()
    X_c2 = g.inv_cholesky(C, ctx=ctx)
    t2 = time.perf_counter()
    ctx.set_option("CHOLESKY", 0)
    X_s = g.inv_cholesky(C, ctx=ctx)
    t3 = time.perf_counter()
    sc = np.abs(X_l).max()
    d_c, d_s = np.abs(X_c - X_l).max() / sc, np.abs(X_s - X_l).max() / sc
    print("n=%5d: cholesky fallback vs LAPACK %.2e, sweep vs LAPACK %.2e; symmetric %s, rerun equal %s; wall (with host copies) %.1f / %.1f ms, sweep %.1f ms"
          % (n, d_c, d_s, np.array_equal(X_c, X_c.T), np.array_equal(X_c, X_c2), (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
    ok &= d_c < 1e-11
    for bad in sorted({0, n // 3, n - 1}):
        Cb = C.copy()
        Cb[bad, bad] = -1.0
        ctx.set_option("CHOLESKY", 2)
        try:
            g.inv_cholesky(Cb, ctx=ctx)
            got = 0
        except g.PosDefException as e:
            got = e.info
        try:
            o.spd_inverse(Cb)
            want = 0
        except o.NotPositiveDefinite as e:
            want = e.info
        if got != want:
            ok = False
            print("   not-PD index: got %d, LAPACK %d" % (got, want))
ctx.set_option("CHOLESKY", 1)
if args.families:
    from gaussdca.jl_amd import synth
    from gdca_testutil import score_close  # noqa

    Zo = synth.synth_family(430, 600, 21, 0x1C0D)
    for pc in (1e-6, 1e-8, 1e-9, 1e-10, 1e-11, 1e-12):
        try:
            S_o = o.scores_from_Z(Zo, 21, pc, "auto", "frob")
            want = "ok"
        except o.NotPositiveDefinite as e:
            want = "not PD (%d)" % e.info
        t0 = time.perf_counter()
        try:
            S, st = ctx.run(np.asfortranarray(Zo.T), 21, pc, -1.0, 0)
            got = "ok"
        except g.PosDefException as e:
            got = "not PD (%d)" % e.info
            st = None
        dt = (time.perf_counter() - t0) * 1e3
        line = "pc=%g: oracle %s, device %s (%.0f ms)" % (pc, want, got, dt)
        if st is not None and want == "ok":
            rel = np.abs(S - S_o).max() / np.abs(S_o).max()
            line += "; refined %d, ||X||_1 %.2e, max |dS| / max |S| = %.2e" % (st["refined"], st["inverse_norm1"], rel)
        print(line, flush=True)
        ok &= (want == "ok") == (got == "ok")
print("OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
