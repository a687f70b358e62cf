"""One-off random campaign over the body of tests/test_gpu_properties.py (GPU == oracle on small random alignments) with numpy-drawn
parameters instead of hypothesis' fixed examples:  python tools/campaign_properties.py [seed] [draws]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import gaussdca.jl_amd as g
from oracle import gdca_oracle as o
import test_gpu_properties as T
inner = T.test_fused_path_matches_oracle.hypothesis.inner_test
ctx = g.Context(0)
env = (g, o, ctx)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 12345)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
t0 = time.time(); fails = 0
for k in range(n):
    seed = int(rng.integers(0, 2**31 - 1)); M = int(rng.integers(2, 301)); N = int(rng.integers(3, 41))
    q = int(rng.choice([3, 5, 21, 24])); theta = "auto" if rng.random() < 0.5 else float(rng.random() * 0.6)
    pc = float(0.05 + 0.95 * rng.random()); score = "frob" if rng.random() < 0.5 else "DI"
    try:
        inner(env, seed, M, N, q, theta, pc, score)
    except Exception as e:  # noqa
        fails += 1
        print("FAIL", dict(seed=seed, M=M, N=N, q=q, theta=theta, pc=pc, score=score), type(e).__name__, str(e)[:200])
print("%d random draws, %d failures, %.0f s" % (n, fails, time.time() - t0))
