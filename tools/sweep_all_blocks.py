"""Every block count 1 .. NB with every group size: the persistent SPD inverse against residual probes (A X v = v), symmetry and
run-to-run bit-identity.  One child process per GDCA_GROUP (the library reads its switches once).

    python tools/sweep_all_blocks.py [NB=56] [extra env, e.g. GDCA_RAMP=0]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
import gaussdca.jl_amd as g
ctx = g.Context(0)
rng = np.random.default_rng(5)
NB = int(sys.argv[2])
bad = []
worst = 0.0
for nb in range(1, NB + 1):
    for n in sorted({128 * nb, 128 * nb - 37 if nb > 1 else 91}):
        B = rng.standard_normal((n, 24))
        A = (B @ B.T) / 24 + np.diag(0.5 + rng.random(n))
        X = g.inv_cholesky(A, ctx=ctx)
        X2 = g.inv_cholesky(A, ctx=ctx)
        V = rng.standard_normal((n, 4))
        res = float(np.max(np.abs(A @ (X @ V) - V)) / np.max(np.abs(V)))
        worst = max(worst, res)
        if not (res < 1e-9 and np.array_equal(X, X.T) and np.array_equal(X, X2)):
            bad.append([n, res, bool(np.array_equal(X, X.T)), bool(np.array_equal(X, X2))])
print(json.dumps({"bad": bad, "worst_residual": worst}))
'''
NB = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 56
extra = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
fail = 0
for gsz in (1, 2, 3, 4):
    env = dict(os.environ, GDCA_GROUP=str(gsz), **extra)
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, str(NB)], capture_output=True, text=True, env=env, timeout=3000)
    if r.returncode != 0:
        print("g=%d: child failed: %s" % (gsz, r.stderr[-500:]))
        fail += 1
        continue
    out = json.loads(r.stdout.strip().splitlines()[-1])
    print("g=%d %s: %d sizes up to %d blocks, worst residual %.2e, failures: %s" % (gsz, extra, 2 * NB, NB, out["worst_residual"], out["bad"] or "none"))
    fail += len(out["bad"])
sys.exit(1 if fail else 0)
