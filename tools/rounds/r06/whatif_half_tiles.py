"""Timing experiment (wrong results on purpose): K = 128 tile items that skip half of their chunks (SWEEP_DEBUG bit 6 of an experiment build) -- how much of
a mid-size inverse's time is the tile items' execution?  Wall time of the operator-level inverse, device-resident matrix."""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
import gaussdca.jl_amd as g
from gaussdca.jl_amd import devops
from gaussdca.jl_amd._lib import DeviceBuffer
ctx = g.Context(0)
ctx.set_option("CHOLESKY", "0"); ctx.set_option("REFINE", "0")
rng = np.random.default_rng(0)
for n in (2560, 4096, 5632):
    B = rng.standard_normal((n, 64))
    A = B @ B.T / 64 + np.diag(0.5 + rng.random(n))
    dA = DeviceBuffer(ctx, 8 * n * n)
    for dbg in (0, 64, 0, 64):
        ctx.set_option("SWEEP_DEBUG", str(dbg))
        ts = []
        for r in range(6):
            dA.upload(A)
            t0 = time.perf_counter()
            try:
                devops.inv_cholesky_dev(ctx, dA, n)
            except Exception:
                pass
            ts.append((time.perf_counter() - t0) * 1e3)
        print("n %d debug %2d wall ms %s" % (n, dbg, " ".join("%.3f" % t for t in ts)), flush=True)
