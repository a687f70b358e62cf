"""Two sweep launches that are NOT gated against each other on one GPU: (a) two non-peer contexts of this process,
each driven by a thread of its own, (b) two processes.  Every run must give the LAPACK inverse; how long they take side by side is printed.
    python tools/side_by_side_probe.py [n] [rounds] [merged | merged_procs]            (child mode: ... child <tag>)"""
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def spd(n, seed):
    rng = np.random.default_rng(seed)
    B = rng.standard_normal((n, 40))
    return (B @ B.T) / 40 + np.diag(0.3 + rng.random(n))


def work(g, ctx, A, Xr, rounds, tag, out):
    worst, t0 = 0.0, time.time()
    for _ in range(rounds):
        X = g.inv_cholesky(A, ctx=ctx)
        worst = max(worst, float(np.max(np.abs(X - Xr)) / np.max(np.abs(Xr))))
    out[tag] = (worst, time.time() - t0)


def merged_work(g, rounds, out):
    """eight matrices of config B's size through ONE merged launch, again and again (gdca_spd_inverse_batch_dev)"""
    import torch

    pool = [g.Context(0) for _ in range(8)]
    ns = [2560 - 16 * k for k in range(8)]
    As = [spd(n, 100 + k) for k, n in enumerate(ns)]
    Xr = [np.linalg.inv(A) for A in As]
    worst, t0 = 0.0, time.time()
    try:
        for _ in range(rounds):
            ds = [torch.from_numpy(A).cuda() for A in As]
            torch.cuda.synchronize()
            g.spd_inverse_batch_dev(pool, [d.data_ptr() for d in ds], ns)
            for d, X0 in zip(ds, Xr):
                worst = max(worst, float(np.max(np.abs(d.cpu().numpy() - X0)) / np.max(np.abs(X0))))
        out["merged"] = (worst, time.time() - t0)
    except Exception as e:  # noqa: BLE001
        out["merged"] = (float("inf"), "%s: %s" % (type(e).__name__, str(e)[:100]))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 9100
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    child = len(sys.argv) > 4 and sys.argv[3] == "child"
    import gaussdca.jl_amd as g

    A = spd(n, 3 if not child else 4 + len(sys.argv[4]))
    Xr = np.linalg.inv(A)
    if child and sys.argv[4] == "merged":
        # (a process of its own that does nothing but merged launches of eight)
        import torch

        torch.zeros(1).cuda()
        out = {}
        merged_work(g, rounds, out)
        print("child merged: worst rel %s, %s" % (out["merged"][0], out["merged"][1]), flush=True)
        sys.exit(0 if out["merged"][0] <= 1e-10 else 1)
    if child:
        out = {}
        work(g, g.Context(0), A, Xr, rounds, "p", out)
        print("child %s: worst rel %.2e, %.2f s" % (sys.argv[4], out["p"][0], out["p"][1]), flush=True)
        sys.exit(0 if out["p"][0] <= 1e-10 else 1)
    if len(sys.argv) > 3 and sys.argv[3] == "merged_procs":
        # merged launches of eight in ONE process beside ungated single-family sweeps of n rows in ANOTHER: nothing orders
        # the two processes' launches, not even a common runtime
        t0 = time.time()
        ps = [subprocess.Popen([sys.executable, __file__, str(n), str(rounds), "child", "big"]),
              subprocess.Popen([sys.executable, __file__, str(n), str(8 * rounds), "child", "merged"])]
        rcs = [p.wait() for p in ps]
        print("merged launches of 8 in one process beside n=%d sweeps in another: exit codes %s, %.1f s" % (n, rcs, time.time() - t0), flush=True)
        sys.exit(0 if rcs == [0, 0] else 1)
    if len(sys.argv) > 3 and sys.argv[3] == "merged":
        # a MERGED launch (eight members) beside an ungated single-family sweep of n rows: the case the batch driver once died of
        import torch

        torch.zeros(1).cuda()   # (torch's device context is created by the main thread; the worker threads only use it)
        c1 = g.Context(0)
        out = {}
        work(g, c1, A, Xr, 1, "warm", out)
        ts = [threading.Thread(target=work, args=(g, c1, A, Xr, rounds, "big", out)), threading.Thread(target=merged_work, args=(g, 6 * rounds, out))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        print("merged launches of 8 beside n=%d: big %s, merged %s" % (n, out.get("big"), out.get("merged")), flush=True)
        ok = out["big"][0] <= 1e-10 and out["merged"][0] <= 1e-10
        sys.exit(0 if ok else 1)
    c1, c2 = g.Context(0), g.Context(0)          # NOT peers: nothing orders their inverses
    out = {}
    work(g, c1, A, Xr, 1, "warm", out)
    t0 = time.time()
    work(g, c1, A, Xr, rounds, "alone", out)
    ts = [threading.Thread(target=work, args=(g, c, A, Xr, rounds, tag, out)) for c, tag in ((c1, "t1"), (c2, "t2"))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    print("n=%d, %d inverses each: alone %.2f s; two threads side by side %.2f s / %.2f s, worst rel %.2e %.2e"
          % (n, rounds, out["alone"][1], out["t1"][1], out["t2"][1], out["t1"][0], out["t2"][0]), flush=True)
    ps = [subprocess.Popen([sys.executable, __file__, str(n), str(rounds), "child", tag]) for tag in ("a", "bb")]
    rcs = [p.wait() for p in ps]
    print("two processes: exit codes", rcs, flush=True)
    ok = out["t1"][0] <= 1e-10 and out["t2"][0] <= 1e-10 and rcs == [0, 0]
    sys.exit(0 if ok else 1)
