"""Shader clock of the sweep kernel group by group INSIDE the hot path (front end, inverse, scores back to back on a device-resident family,
the way bench.py's loop runs it): python tools/clock_ramp_hot_path.py [N M reps].  The trace (GDCA_SWEEP_TRACE) keeps the LAST run's inverse.
Beside tools/clock_ramp.py (inverses only, back to back) it shows what the front end's kernels in front of every sweep cost it in clock."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
M = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
TRACE = os.environ.setdefault("GDCA_SWEEP_TRACE", "/tmp/sweep_trace_hot.txt")
import torch
import gaussdca.jl_amd as g
from gaussdca.jl_amd import synth
ctx = g.Context(0)
Z = torch.from_numpy(synth.synth_family(N, M, 21, 1234)).cuda()
S = torch.empty((N, N), dtype=torch.float64, device="cuda")
out = []
for r in range(reps):
    t0 = time.perf_counter()
    st = ctx.run_dev(Z.data_ptr(), N, M, 21, 0.8, -1.0, 0, S.data_ptr())
    out.append(((time.perf_counter() - t0) * 1e3, st["ms_inverse"], st["sweep_ghz"]))
print("wall ms / ms_inverse / sweep GHz per run:", " | ".join("%.2f %.2f %.3f" % x for x in out))
for l in open(TRACE):
    if l.startswith("# per group") or l.startswith("# shader clock"):
        print(l.strip()[:1600])
