"""Merged SPD inverses (gdca_spd_inverse_batch_dev: K matrices carried by ONE k_sweep_merged launch) against LAPACK-grade
references and against the single launch, bit for bit: every block count, ragged ends, mixed batches, K = 1 .. 8, the workspaces
poisoned with NaNs before every run (option SWEEP_DEBUG bit 4: an item that reads a buffer before its producer wrote it cannot
pass on the leftovers of an earlier identical run) and fresh contexts for part of the runs.

    python tools/stress_merged.py [--rounds 40] [--maxblocks 48] [--seed 1]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import gaussdca.jl_amd as g

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=40)
ap.add_argument("--maxblocks", type=int, default=57)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--poison", type=int, default=1)
args = ap.parse_args()
rng = np.random.default_rng(args.seed)


def mat(n):
    B = rng.standard_normal((n, 24))
    return (B @ B.T) / 24 + np.diag(0.5 + rng.random(n))


single = g.Context(0)
single.set_options(SWEEP_DEBUG=16 if args.poison else 0)
pool = [g.Context(0) for _ in range(8)]
bad = 0
runs = 0
t0 = time.time()
worst = 0.0
for rnd in range(args.rounds):
    K = int(rng.integers(1, 9))
    ns = []
    for _ in range(K):
        nb = int(rng.integers(1, args.maxblocks + 1))
        ns.append(128 * nb if rng.random() < 0.4 else max(1, 128 * nb - int(rng.integers(1, 128))))
    if rnd % 5 == 0:  # fresh contexts (new allocations) now and then
        for c in pool:
            c.close()
        pool = [g.Context(0) for _ in range(8)]
    As = [mat(n) for n in ns]
    ref = []
    for A in As:
        d = torch.from_numpy(A).cuda()
        torch.cuda.synchronize()
        info = g._lib.C.c_int32()
        single.check(single.lib.gdca_spd_inverse_dev(single.h, g._lib.C.c_void_p(d.data_ptr()), A.shape[0], g._lib.C.byref(info)))
        ref.append(d.cpu().numpy())
    # the single launch against numpy on a few probes
    for A, X in zip(As, ref):
        V = rng.standard_normal((A.shape[0], 3))
        res = float(np.max(np.abs(A @ (X @ V) - V)))
        worst = max(worst, res)
        if not res < 1e-8:
            bad += 1
            print("round %d: SINGLE launch residual %.2e at n=%d" % (rnd, res, A.shape[0]), flush=True)
    for merge, dbg, grp in ((8, 8, -1), (4, 8, -1), (2, 8, 1), (3, 0, 1), (5, 0, -1)):
        cs = pool[:K]
        cs[0].set_options(MERGE=merge, MERGE_BLOCKS=args.maxblocks, MERGE_TILES=int(rng.choice([600, 2300, 1 << 20])), MERGE_GROUP=grp,
                          SWEEP_DEBUG=dbg | (16 if args.poison else 0))
        got = []
        for rep in range(2):   # twice: a batch must give the same bits every time
            ds = [torch.from_numpy(A).cuda() for A in As]
            torch.cuda.synchronize()
            g.spd_inverse_batch_dev(cs, [d.data_ptr() for d in ds], ns)
            got.append([d.cpu().numpy() for d in ds])
            runs += 1
        for k in range(K):
            X = got[0][k]
            nb = (ns[k] + 127) // 128
            # single-block groups (MERGE_GROUP = 1) are the schedule of a launch of its own up to 44 blocks (48 until round 5's rule): bit for bit; larger
            # pivot groups sum in another order: the same inverse to rounding
            exact = grp == 1 and nb <= 44
            same = np.array_equal(X, ref[k]) if exact else bool(np.max(np.abs(X - ref[k])) <= 1e-11 * np.max(np.abs(ref[k])))
            if not (same and np.array_equal(X, got[1][k]) and np.array_equal(X, X.T)):
                bad += 1
                E = np.abs(X - ref[k])
                print("round %d merge %d group %d: member %d of %d (n=%d, %d blocks): vs single launch max %.2e (exact wanted: %s), nan %d, "
                      "rerun equal %s, symmetric %s, batch sizes %s"
                      % (rnd, merge, grp, k, K, ns[k], nb, float(np.nanmax(E)), exact, int(np.isnan(X).sum()),
                         np.array_equal(X, got[1][k]), np.array_equal(X, X.T), ns), flush=True)
print("%d merged batches (%d rounds), worst single-launch residual %.2e, mismatches: %d, %.0f s" % (runs, args.rounds, worst, bad, time.time() - t0))
sys.exit(1 if bad else 0)
