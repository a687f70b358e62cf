"""BASELINE.json configs[4] on ONE GPU's share: a batch of Pfam-like families (N in [100,600],
M in [5k,80k]) from gaussdca.jl_amd.batch, sharded over WORLD_SIZE ranks (no collective), with
--pipeline families in flight per GPU.  Prints families/s for this rank's shard.

    python tools/batch_bench.py --families 32 --pipeline 2
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import gaussdca.jl_amd as g  # noqa: E402
from gaussdca.jl_amd.batch import batch_sizes, shard_families  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--families", type=int, default=32)
    ap.add_argument("--pipeline", type=int, default=2)
    ap.add_argument("--max-m", type=int, default=80000)
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    sizes = [(n, min(m, args.max_m)) for n, m in batch_sizes(args.families)]
    mine = shard_families(sizes, world)[rank]
    fams = []
    for f in mine:
        N, M = sizes[f]
        Z = torch.from_numpy(bench.synth_family(N, M, 21, 0xE000 + f)).cuda()
        S = torch.empty((N, N), dtype=torch.float64, device="cuda")
        fams.append((f, N, M, Z, S))
    P = max(1, args.pipeline)
    ctxs = [g.Context(local) for _ in range(P)]
    for c in ctxs:
        c.set_timing(False)

    def run_all():
        busy = [False] * P
        for t, (f, N, M, Z, S) in enumerate(fams):
            c = t % P
            if busy[c]:
                ctxs[c].collect()
            ctxs[c].run_dev_async(Z.data_ptr(), N, M, 21, 0.8, -1.0, 0, S.data_ptr())
            busy[c] = True
        for c in range(P):
            if busy[c]:
                ctxs[c].collect()

    run_all()  # warm-up: workspaces grow to the largest family
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_all()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    flops = sum((N * 20.0) ** 3 for _, N, _, _, _ in fams)
    print(json.dumps({"rank": rank, "world": world, "families": len(fams), "pipeline": P, "seconds": dt,
                      "families_per_s": len(fams) / dt, "inverse_tflops_aggregate": flops / dt / 1e12,
                      "sizes_N_min_max": [min(n for _, n, _, _, _ in fams), max(n for _, n, _, _, _ in fams)],
                      "all_finite": bool(all(torch.isfinite(S).all().item() for *_, S in fams))}))


if __name__ == "__main__":
    main()
