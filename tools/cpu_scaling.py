"""Thread scaling of the CPU port's two OpenMP loops (oracle/oracle_kernels.c) on this host, at config C's size:
    python tools/cpu_scaling.py [N] [M]
Prints seconds for the all-pairs Hamming pass and the pair tallies at 1, 8, 32, 64, 128, all threads."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussdca.jl_amd import synth
from oracle import gdca_oracle as o
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
M = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
Z = synth.synth_family(N, M, 21, 0xC500)
cores = len(os.sched_getaffinity(0))
print("host threads", cores, flush=True)
thr = int(0.35 * N)
for t in [x for x in (1, 8, 32, 64, 128, 256) if x <= cores] + ([cores] if cores not in (1, 8, 32, 64, 128, 256) else []):
    o.set_threads(t)
    Ms = M if t >= 32 else min(M, 12000)     # the low-thread points on a subset, scaled by the pair count
    t0 = time.time(); n_k = o.neighbour_counts(Z[:Ms], thr); dt = (time.time() - t0) * (M * (M - 1.0)) / (Ms * (Ms - 1.0))
    W = np.ones(M); Mf = M if t >= 32 else min(M, 4000)
    t0 = time.time(); o.compute_frequencies(Z[:Mf], 21, W[:Mf], float(Mf)); df = (time.time() - t0) * M / Mf
    print("threads %3d: hamming %.2f s  tallies %.2f s" % (t, dt, df), flush=True)
