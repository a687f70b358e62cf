"""One-off random campaign: whole hot path against the oracle on synthetic families with N in [60, 440) (10 .. 69 pivot blocks: every
group size the schedule rule picks below n = 9000), alternating :frob / :DI; theta, thresh, Meff must be equal, scores within 1e-6:
    python tools/campaign_midsize.py [seed] [families]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import gaussdca.jl_amd as g
from gaussdca.jl_amd import synth
from oracle import gdca_oracle as o
from gdca_testutil import score_close
ctx = g.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 99)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
t0 = time.time(); fails = 0; worst = 0.0
for k in range(n):
    N = int(rng.integers(60, 440)); M = int(rng.integers(600, 3000)); score = "frob" if k % 2 == 0 else "DI"
    pc = 0.8 if score == "frob" else 0.2
    Zo = synth.synth_family(N, M, 21, int(rng.integers(1, 2**31 - 1)))
    Z = np.asfortranarray(Zo.T)
    S, st = ctx.run(Z, 21, pc, -1.0, 1 if score == "DI" else 0)
    W, Meff, th, thr = o.compute_weights(Zo, "auto")
    S_o = o.scores_from_Z(Zo, 21, pc, "auto", score)
    ok, mr, ma = score_close(S, S_o, rtol=1e-6, atol_frac=1e-9)
    worst = max(worst, mr)
    good = ok and st["thresh"] == thr and st["Meff"] == Meff and st["theta"] == th
    if not good:
        fails += 1
        print("FAIL", N, M, score, mr, ma, st["thresh"], thr, st["Meff"], Meff)
print("%d mid-size families (N 60..440, blocks %d..%d), %d failures, worst rel dev %.2e, %.0f s" % (n, 60 * 20 // 128 + 1, 440 * 20 // 128 + 1, fails, worst, time.time() - t0))
