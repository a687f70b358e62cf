"""Shader clock of the sweep kernel group by group when inverses run back to back (no idle gap, no other kernels between
them): python tools/clock_ramp.py [n] [repeats].  A -> inv(A) -> A ... in place in HBM; the trace (GDCA_SWEEP_TRACE) keeps the
LAST inverse.  Shows whether the ~25 ms clock ramp of an isolated inverse is a property of the kernel or of the load step."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
TRACE = os.environ.setdefault("GDCA_SWEEP_TRACE", "/tmp/sweep_trace.txt")
import gaussdca.jl_amd as g
from gaussdca.jl_amd import devops
from gaussdca.jl_amd._lib import DeviceBuffer
ctx = g.Context(0)
rng = np.random.default_rng(0)
B = rng.standard_normal((n, 64))
A = B @ B.T / 64 + np.diag(0.5 + rng.random(n))
dA = DeviceBuffer(ctx, 8 * n * n)
dA.upload(A)
t = []
for r in range(reps):
    t0 = time.perf_counter()
    devops.inv_cholesky_dev(ctx, dA, n)
    t.append((time.perf_counter() - t0) * 1e3)
print("wall ms per inverse (incl. the trace's own synchronisation):", " ".join("%.2f" % x for x in t))
for l in open(TRACE):
    if l.startswith("#"):
        print(l.strip()[:1600])
