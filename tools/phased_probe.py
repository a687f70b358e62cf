"""Per-position inverse times inside a phase-batched run (gdca_run_dev_phased): K families of config C's size, the K inverses
back to back on one stream.  python tools/phased_probe.py [K] [reps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gaussdca.jl_amd as g
from gaussdca.jl_amd import synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N, M = 500, 50000
Z = torch.from_numpy(synth.synth_family(N, M, 21, 0xC500)).cuda()
S = [torch.empty((N, N), dtype=torch.float64, device="cuda") for _ in range(K)]
cs = [g.Context(0) for _ in range(K)]
for r in range(reps):
    g.run_dev_phased(cs, [Z.data_ptr()] * K, [N] * K, [M] * K, [21] * K, 0.8, -1.0, 0, [s.data_ptr() for s in S])
    st = [c.collect() for c in cs]
    print("rep %d: k_sweep ms by position: %s" % (r, " ".join("%.2f" % s["ms_inverse_update"] for s in st)))
    print("        clock GHz by position:  %s" % " ".join("%.3f" % s["sweep_ghz"] for s in st))
