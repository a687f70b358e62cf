"""Inverse-stage time of the hot path versus the pivot-group size g (GDCA_GROUP, one process per g) over alignment lengths N:
the data behind the g(nblk) rule in gdca_launch_spd_inverse."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
import gaussdca.jl_amd as g
from gaussdca.jl_amd import synth
ctx = g.Context(0)
out = {}
for N in [int(x) for x in sys.argv[2].split(",")]:
    Z = np.asfortranarray(synth.synth_family(N, 3000, 21, 7 + N).T)
    best = 1e9
    for rep in range(4):
        S, st = ctx.run(Z, 21, 0.8, 0.3, 0)
        best = min(best, st["ms_inverse"])
    out[N] = best
print(json.dumps(out))
'''
Ns = os.environ.get("SWEEP_NS", "100,128,200,300,350,400,450,500,550,600,800,1000")
res = {}
combos = [(gsz, m) for gsz in (1, 2, 3, 4) for m in ((8,) if "--quick" in sys.argv else (4, 8, 12, 16))]
if os.environ.get("SWEEP_COMBOS"):   # "g:m,g:m,..."
    combos = [tuple(int(x) for x in c.split(":")) for c in os.environ["SWEEP_COMBOS"].split(",")]
for gsz, m in combos:
    env = dict(os.environ, GDCA_GROUP=str(gsz), GDCA_MCUS=str(m))
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, Ns], capture_output=True, text=True, env=env, timeout=900)
    res[(gsz, m)] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-300:]}
print("ms_inverse; columns = (GDCA_GROUP, GDCA_MCUS)")
print("N   nblk " + " ".join("g%d/m%-2d   " % k for k in res))
for N in Ns.split(","):
    nblk = -(-int(N) * 20 // 128)
    row = [res[k].get(N, float("nan")) for k in res]
    best = min(range(len(row)), key=lambda i: row[i])
    print("%4s %4d " % (N, nblk) + " ".join("%8.3f" % x for x in row) + "   best g=%d mcus=%d" % combos[best])
