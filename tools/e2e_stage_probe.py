"""Stage times of gDCA(filename) (a run after the parse's idle time) beside those of back-to-back device-resident runs of the same family.
    python tools/e2e_stage_probe.py [C|D]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import gaussdca.jl_amd as g
from gaussdca.jl_amd import synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "C"
N, M, seed = (500, 50000, 0xC500) if cfg == "C" else (1000, 100000, 0xD1000)
Zh = synth.synth_family(N, M, 21, seed)
letters = np.frombuffer(b"?ACDEFGHIKLMNPQRSTVWY-", dtype=np.uint8)
with tempfile.NamedTemporaryFile("wb", suffix=".fasta", delete=False) as f:
    path = f.name
    for k in range(Zh.shape[0]):
        f.write(b">s%d\n" % k)
        f.write(letters[Zh[k]].tobytes())
        f.write(b"\n")
ctx = g.Context(0)
keys = ("ms_total", "ms_theta", "ms_weights", "ms_covariance", "ms_inverse", "ms_score", "sweep_ghz")
try:
    for r in range(5):
        t0 = time.perf_counter()
        R = g.gDCA(path, ctx=ctx)
        dt = time.perf_counter() - t0
        st = g.gdca.last_stats
        print("gDCA(file)   %.1f ms wall | " % (dt * 1e3) + "  ".join("%s %.3f" % (k[3:] if k.startswith("ms_") else k, st[k]) for k in keys), flush=True)
        time.sleep(0.05 * r)
    Zd = torch.from_numpy(Zh).cuda()
    S = torch.empty((N, N), dtype=torch.float64, device="cuda")
    for r in range(6):
        st = ctx.run_dev(Zd.data_ptr(), N, M, 21, 0.8, -1.0, 0, S.data_ptr())
        print("back to back              | " + "  ".join("%s %.3f" % (k[3:] if k.startswith("ms_") else k, st[k]) for k in keys), flush=True)
    for gap in (0.004, 0.02, 0.1):
        time.sleep(gap)
        st = ctx.run_dev(Zd.data_ptr(), N, M, 21, 0.8, -1.0, 0, S.data_ptr())
        print("after %5.0f ms of idle     | " % (gap * 1e3) + "  ".join("%s %.3f" % (k[3:] if k.startswith("ms_") else k, st[k]) for k in keys), flush=True)
finally:
    os.unlink(path)
