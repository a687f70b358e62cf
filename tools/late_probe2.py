import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
def say(*a):
    print(*a, flush=True)
say("lib", os.environ.get("GDCA_LIB"))
import torch
import gaussdca.jl_amd as g
say("imported")
c = g.Context(0)
say("context")
c.set_options(SWEEP_TIMEOUT_MS=1500)
rng = np.random.default_rng(1)
for n in (128, 700, 2560):
    B = rng.standard_normal((n, 24)); A = (B @ B.T) / 24 + np.diag(0.5 + rng.random(n))
    t0 = time.time()
    try:
        X = g.inv_cholesky(A, ctx=c); st = "ok %.1e" % float(np.max(np.abs(A @ X - np.eye(n))))
    except Exception as e:
        st = "%s %s" % (type(e).__name__, str(e)[:60])
    say("k_sweep n=%d: %.2f s %s" % (n, time.time() - t0, st))
c2 = g.Context(0)
c2.set_options(SWEEP_TIMEOUT_MS=1500, SWEEP_DEBUG=8)
for n in (128, 700, 2560):
    B = rng.standard_normal((n, 24)); A = (B @ B.T) / 24 + np.diag(0.5 + rng.random(n))
    t0 = time.time()
    try:
        X = g.inv_cholesky(A, ctx=c2); st = "ok %.1e" % float(np.max(np.abs(A @ X - np.eye(n))))
    except Exception as e:
        st = "%s %s" % (type(e).__name__, str(e)[:60])
    say("k_sweep_merged (one member) n=%d: %.2f s %s" % (n, time.time() - t0, st))
