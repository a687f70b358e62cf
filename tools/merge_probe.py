"""Merged SPD inverses (gdca_run_dev_phased with option MERGE = K: one k_sweep_merged launch carries K small inverses):
bit-identity against single runs and the amortised inverse time per family.

    python tools/merge_probe.py [--sizes 128:10000 ...] [--ks 1 2 4 8] [--reps 5] [--theta 0.2] [--mcus -1]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import gaussdca.jl_amd as g
from gaussdca.jl_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", nargs="*", default=["128:10000"], help="N:M of the families of a batch (cycled up to K)")
ap.add_argument("--ks", nargs="*", type=int, default=[1, 2, 4, 8])
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--theta", type=float, default=0.2)
ap.add_argument("--mcus", type=int, default=-1, help="MERGE_MCUS: chain compute units per member")
ap.add_argument("--blocks", type=int, default=64, help="MERGE_BLOCKS")
ap.add_argument("--group", type=int, default=-1, help="MERGE_GROUP: pivot blocks per group of a merged member (-1: the rule)")
ap.add_argument("--tiles", type=int, default=1 << 20, help="MERGE_TILES (default: no limit, K members per launch)")
args = ap.parse_args()
sizes = [tuple(int(x) for x in s.split(":")) for s in args.sizes]
KMAX = max(args.ks)
fams = [sizes[k % len(sizes)] for k in range(KMAX)]
Zh = {sz: synth.synth_family(sz[0], sz[1], 21, 0xB128 + sz[0]) for sz in set(fams)}
Zd = {sz: torch.from_numpy(z).cuda() for sz, z in Zh.items()}
ctx = g.Context(0)
ref = {}
for sz in Zd:
    S = torch.empty((sz[0], sz[0]), dtype=torch.float64, device="cuda")
    st = None
    t = []
    for _ in range(4):
        st = ctx.run_dev(Zd[sz].data_ptr(), sz[0], sz[1], 21, 0.8, args.theta, 0, S.data_ptr())
        t.append(st["ms_inverse_update"])
    ref[sz] = S.cpu()
    print("single N=%d M=%d (%d blocks): k_sweep %.3f ms (best of 4 %.3f), family %.3f ms" %
          (sz[0], sz[1], st["n_pad"] // 128, float(np.mean(t[1:])), min(t), st["ms_total"]), flush=True)
cs = [g.Context(0) for _ in range(KMAX)]
cs2 = [g.Context(0) for _ in range(KMAX)]
outs = [torch.empty((f[0], f[0]), dtype=torch.float64, device="cuda") for f in fams]
outs2 = [torch.empty((f[0], f[0]), dtype=torch.float64, device="cuda") for f in fams]
ok = True
worst_dev = 0.0
for K in args.ks:
    for lead in (cs[0], cs2[0]):
        lead.set_options(MERGE=K, MERGE_BLOCKS=args.blocks, MERGE_MCUS=args.mcus, MERGE_GROUP=args.group, MERGE_TILES=args.tiles)
    f = fams[:K]

    def enqueue(cset, oset):
        g.run_dev_phased(cset[:K], [Zd[x].data_ptr() for x in f], [x[0] for x in f], [x[1] for x in f], [21] * K, 0.8, args.theta, 0,
                         [x.data_ptr() for x in oset[:K]])

    inv, wall = [], []
    for r in range(args.reps):
        for o in outs[:K]:
            o.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        enqueue(cs, outs)
        st = [c.collect() for c in cs[:K]]
        wall.append((time.perf_counter() - t0) * 1e3)
        inv.append(sum(s["ms_inverse_update"] for s in st))
        for k in range(K):
            dev = float(((outs[k].cpu() - ref[f[k]]).abs().max() / ref[f[k]].abs().max()))
            worst_dev = max(worst_dev, dev)
            if not dev < 1e-9:
                ok = False
                print("  MISMATCH K=%d member %d (N=%d): max rel dev %.3e" % (K, k, f[k][0], dev))
    # throughput form: two context sets alternate, the next batch is enqueued before the previous one is collected
    torch.cuda.synchronize()
    nb = 12
    t0 = time.perf_counter()
    pend = None
    for b in range(nb):
        cset, oset = (cs, outs) if b % 2 == 0 else (cs2, outs2)
        enqueue(cset, oset)
        if pend is not None:
            [c.collect() for c in pend[:K]]
        pend = cset
    [c.collect() for c in pend[:K]]
    dt = (time.perf_counter() - t0) * 1e3
    print("K=%d: batch %s  merged launch %.3f ms = %.3f ms per family (best %.3f); batch wall %.3f ms; pipelined %.3f ms per family "
          "(batch sizes %s)" % (K, [s["inverse_batch"] for s in st], float(np.mean(inv[1:])), float(np.mean(inv[1:])) / K, min(inv) / K,
                                float(np.mean(wall[1:])), dt / (nb * K), [x[0] for x in f]), flush=True)
print("scores equal to single runs (worst relative deviation %.2e; 0 = bit-identical): %s" % (worst_dev, ok))
sys.exit(0 if ok else 1)
