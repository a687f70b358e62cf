"""What does the "next take published late" variant of the merged loop do?  A handful of merged batches through
GDCA_LIB's build, every step reported as it happens: time per batch, status (a watchdog abort is GDCA_EHIP), deviation from numpy.
    GDCA_LIB=gaussdca.jl_amd/libgdca_late.so python tools/late_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import gaussdca.jl_amd as g

rng = np.random.default_rng(5)


def mat(n):
    B = rng.standard_normal((n, 24))
    return (B @ B.T) / 24 + np.diag(0.5 + rng.random(n))


pool = [g.Context(0) for _ in range(8)]
for trial, (K, nb) in enumerate([(1, 6), (1, 20), (2, 20), (8, 20), (4, 30), (2, 47), (8, 12), (1, 40), (3, 25)] * 2):
    ns = [128 * nb - (17 * k) % 90 for k in range(K)]
    As = [mat(n) for n in ns]
    cs = pool[:K]
    cs[0].set_options(MERGE=8, MERGE_BLOCKS=57, MERGE_GROUP=-1 if trial % 2 else 1, SWEEP_DEBUG=8 | 16, SWEEP_TIMEOUT_MS=1500)
    ds = [torch.from_numpy(A).cuda() for A in As]
    torch.cuda.synchronize()
    t0 = time.time()
    try:
        g.spd_inverse_batch_dev(cs, [d.data_ptr() for d in ds], ns)
        status = "ok"
    except Exception as e:  # noqa: BLE001
        status = "%s: %s" % (type(e).__name__, str(e)[:80])
    dt = time.time() - t0
    dev = []
    if status == "ok":
        for A, d in zip(As, ds):
            X = d.cpu().numpy()
            V = rng.standard_normal((A.shape[0], 2))
            dev.append(float(np.max(np.abs(A @ (X @ V) - V))))
    print("trial %2d: K=%d blocks=%2d group=%s: %.3f s, %s, residuals %s" % (trial, K, nb, "auto" if trial % 2 else "1", dt, status,
                                                                               " ".join("%.1e" % x for x in dev)), flush=True)
