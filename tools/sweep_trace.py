"""Debug aid: run one SPD inverse of size n with GDCA_SWEEP_TRACE set and print the M-list timeline of a few groups."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
TRACE = os.environ.setdefault("GDCA_SWEEP_TRACE", "/tmp/sweep_trace.txt")
import gaussdca.jl_amd as g
ctx = g.Context(0)
rng = np.random.default_rng(0)
B = rng.standard_normal((n, 64))
A = B @ B.T / 64 + np.diag(0.5 + rng.random(n))
for rep in range(2):   # second call: warm
    try:
        X = g.inv_cholesky(A, ctx=ctx)
    except g.PosDefException as e:   # timing experiments with deliberately wrong arithmetic end up here
        print("not positive definite:", e)
rows = [l.split() for l in open(TRACE) if not l.startswith("#") and not l.startswith("m ")]
for l in open(TRACE):
    if l.startswith("#"): print(l.strip())
gsel = {int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["10", "11"])}
prev_end = None
for q, e, a, b in rows:
    q, e, a, b = int(q), int(e), float(a), float(b)
    if q in gsel:
        print("group %2d item %3d  start %9.1f  end %9.1f  dur %7.1f" % (q, e, a, b, b - a))
last = max(float(r[3]) for r in rows)
print("M list span %.1f us" % last)
