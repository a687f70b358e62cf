"""bench.py -- the gDCA hot path on MI355X, BASELINE.json's headline configuration.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--score frob|DI] [--N 500 --M 50000]

One "step" = one pass of the hot path (src/GaussDCA.jl:28-42 of the reference: reweighting ->
Pi/Pij -> covariance -> SPD inverse -> FN/DI -> APC) over one synthetic protein family whose
int8 alignment is already resident in HBM; the N x N score matrix is left in HBM.  With N > 1
GPUs every rank processes its own independent family per step (weak scaling, no collective on
the data path; torch.distributed is used only for the timing barrier).

Prints ONE JSON line (rank 0).  `value` = families per second over all ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

METRIC = "end-to-end gDCA sec + achieved Cholesky TFLOP/s, N=500 M=50k q=21"
PEAK_F64_MFMA_TFLOPS = 78.6  # AMD MI355X spec, FP64 matrix (not listed in MI355X_MICROARCH.md)


def synth_family(N, M, q, seed):
    """Deterministic 'Pfam-like' synthetic MSA (SplitMix64, native generator in libgdca.so),
    returned as (M, N) int8."""
    from gaussdca.jl_amd.synth import synth_family as gen

    return gen(N, M, q, seed)


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this
    same command (profiles/r01_pmc_update_traffic.json: FETCH_SIZE and WRITE_SIZE, separate passes);
    None when the file is absent or the workload differs from the profiled one."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_update_traffic.json")) as f:
            return float(json.load(f)["hbm_bytes_per_launch"])
    except Exception:  # noqa: BLE001
        return None


def end_to_end(Zh, q, score_name, pc, ctx):
    """gDCA(filename) as a user calls it: FASTA file -> ranking (native parse, upload, hot path, download,
    ranking sort).  Returns the median wall seconds of 3 calls (reported beside `value`, never as `value`)."""
    import tempfile

    import numpy as np

    import gaussdca.jl_amd as g

    letters = np.frombuffer(b"?ACDEFGHIKLMNPQRSTVWY-", dtype=np.uint8)
    with tempfile.NamedTemporaryFile("wb", suffix=".fasta", delete=False) as f:
        path = f.name
        for k in range(Zh.shape[0]):
            f.write(b">s%d\n" % k)
            f.write(letters[Zh[k]].tobytes())
            f.write(b"\n")
    try:
        times = []
        for _ in range(3):
            t = time.perf_counter()
            R = g.gDCA(path, pseudocount=pc, score=score_name, ctx=ctx)
            times.append(time.perf_counter() - t)
        assert len(R) > 0
        return float(sorted(times)[1])
    finally:
        os.unlink(path)


def cpu_baseline(N, M, q, pc, budget_s=40.0):
    """The oracle ("port": numpy + OpenMP/AVX2 C loops + OpenBLAS dpotrf/dpotri) timed on all host
    cores on a bounded sample of the same workload (about `budget_s` seconds of CPU work): the
    all-pairs Hamming pass on a sequence subsample sized from a probe (cost scaled by the pair
    count), the tallies on a sequence subsample (cost linear in M), the SPD inverse at the full n
    when a probe says it fits the budget (else at reduced n, cost scaled by n^3), pseudocount /
    covariance / FN / APC in full.  Returns an estimate of seconds per family."""
    import numpy as np

    from oracle import gdca_oracle as o

    cores = os.cpu_count() or 1
    o.set_threads(cores)
    s = q - 1
    n = N * s
    t_all = time.time()
    Zfull = synth_family(N, M, q, 0xC500)
    theta = o.compute_theta(Zfull)
    thr = o.hamming_threshold(theta, N)
    # Hamming: probe on 3000 sequences, then a sample sized for ~35 % of the budget
    Mp = min(M, 3000)
    t = time.time()
    o.neighbour_counts(Zfull[:Mp], thr)
    rate = (Mp * (Mp - 1.0) / 2) / max(1e-6, time.time() - t)          # pairs per second
    Ms = int(min(M, max(Mp, (2 * 0.35 * budget_s * rate) ** 0.5)))
    t = time.time()
    n_k = o.neighbour_counts(Zfull[:Ms], thr)
    t_ham = (time.time() - t) * (M * (M - 1.0)) / (Ms * (Ms - 1.0))
    W, Meff = o.weights_from_counts(n_k)
    # tallies: linear in M
    Mt = min(Ms, 4000)
    t = time.time()
    Pi, Pij = o.compute_frequencies(Zfull[:Mt], q, W[:Mt], float(W[:Mt].sum()))
    t_freq = (time.time() - t) * (M / Mt)
    t = time.time()
    Pi2, Pij2 = o.add_pseudocount(Pi, Pij, pc, q)
    C = o.compute_C(Pi2, Pij2)
    t_cov = time.time() - t
    # SPD inverse: probe at n/4, then full size if it fits
    npb = max(256, n // 4)
    t = time.time()
    o.spd_inverse(np.ascontiguousarray(C[:npb, :npb]))
    est_full = (time.time() - t) * (n / npb) ** 3
    ns = n if est_full < 0.4 * budget_s else max(npb, int(n * (0.4 * budget_s / est_full) ** (1 / 3)))
    Cs = np.ascontiguousarray(C[:ns, :ns])
    t = time.time()
    mJs = o.spd_inverse(Cs)
    t_inv = (time.time() - t) * (n / ns) ** 3
    mJ = mJs if ns == n else C  # FN cost does not depend on the values
    t = time.time()
    o.correct_APC(o.compute_FN(mJ, q))
    t_fn = time.time() - t
    est = t_ham + t_freq + t_cov + t_inv + t_fn
    return dict(value=1.0 / est, unit="families/s", cores=cores, kind="port",
                sec_per_family_est=est,
                sample=("oracle (numpy + OpenMP/AVX2 C + OpenBLAS potrf/potri) on %d host threads: Hamming on %d of %d "
                        "sequences (x pairs ratio), tallies on %d sequences (x M ratio), potrf+potri at n=%d of %d "
                        "(x n^3 ratio), pseudocount/covariance/FN/APC in full; stage seconds (scaled to the full "
                        "family) ham=%.2f freq=%.2f cov=%.2f inv=%.2f fn=%.2f; sample wall %.1fs"
                        % (cores, Ms, M, Mt, ns, n, t_ham, t_freq, t_cov, t_inv, t_fn, time.time() - t_all)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--score", default="frob")
    ap.add_argument("--N", type=int, default=500)
    ap.add_argument("--M", type=int, default=50000)
    ap.add_argument("--q", type=int, default=21)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="independent families in flight per GPU (one gdca context each): family f+1's "
                         "reweighting/tallies (VALU, LDS) overlap family f's SPD inverse (MFMA); 1 = strictly serial "
                         "(default: clean per-kernel timings; 2 gives +5-10 %% families/s at N=500 and +70 %% at N=128)")
    ap.add_argument("--gate", action="store_true",
                    help="with --pipeline > 1: let the SPD-inverse stages of the families in flight take turns")
    args = ap.parse_args()

    import numpy as np
    import torch

    import gaussdca.jl_amd as g

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the gDCA hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL ("nccl" on ROCm) for the timing barrier / max-over-ranks only; no collective on the data path.
        # gloo is a fallback so that a communicator problem cannot take the measurement down.
        try:
            dist.init_process_group("nccl", device_id=dev)
        except Exception:  # noqa: BLE001
            dist.init_process_group("gloo")

    N, M, q = args.N, args.M, args.q
    score = 1 if args.score == "DI" else 0
    pc = 0.2 if score == 1 else 0.8
    Zh = synth_family(N, M, q, 0xC500 + rank)                 # (M, N) == Julia's N x M column-major bytes
    Zd = torch.from_numpy(Zh).to(dev)                          # resident in HBM before the timed region
    P = max(1, args.pipeline)
    Sd = [torch.empty((N, N), dtype=torch.float64, device=dev) for _ in range(P)]  # stay in HBM
    ctxs = [g.Context(local)]
    for _ in range(P - 1):
        ctxs.append(ctxs[0].peer() if args.gate else g.Context(local))
    busy = [False] * P

    def run_steps(count, sink):
        """`count` hot-path passes, round-robin over the P contexts; a context's previous pass is
        collected (stream sync + stats) right before it is given the next one."""
        for sidx in range(count):
            c = sidx % P
            if busy[c]:
                sink.append(ctxs[c].collect())
            ctxs[c].run_dev_async(Zd.data_ptr(), N, M, q, pc, -1.0, score, Sd[c].data_ptr())
            busy[c] = True
        for c in range(P):
            if busy[c]:
                sink.append(ctxs[c].collect())
                busy[c] = False

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        for c in ctxs:
            c.synchronize()

    warm = max(args.warmup, P)  # every context is warmed (workspace allocation) before the clock starts
    run_steps(warm, [])
    barrier()
    t0 = time.perf_counter()
    stats = []
    run_steps(args.steps, stats)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        K = args.steps
        ms_step = dt / K * 1e3
        upd_ms = float(np.mean([s["ms_inverse_update"] for s in stats]))
        upd_launch = stats[-1]["update_launches"]
        upd_flops = stats[-1]["update_flops"]
        inv_ms = float(np.mean([s["ms_inverse"] for s in stats]))
        achieved = upd_flops / (upd_ms * 1e-3) / 1e12 if upd_ms > 0 else 0.0
        out = {
            "metric": METRIC,
            "value": world * K / dt,
            "unit": "families/s",
            "n_gpus": world,
            "steps": K,
            "warmup": warm,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "synthetic Pfam-like MSA N=%d M=%d q=%d, score=:%s, theta=:auto, pseudocount=%.1f "
                                   "(BASELINE.json configs[2])" % (N, M, q, args.score, pc),
                       "N": N, "M": M, "q": q, "n": N * (q - 1), "families_per_step_per_gpu": 1,
                       "families_in_flight_per_gpu": P},
            "sec_per_family": ms_step * 1e-3,
            "latency_ms_per_family": float(np.mean([s["ms_total"] for s in stats])),
            "spd_inverse_tflops": stats[-1]["inverse_flops"] / (inv_ms * 1e-3) / 1e12,
            "stage_ms": {k: float(np.mean([s[k] for s in stats])) for k in
                         ("ms_total", "ms_theta", "ms_weights", "ms_covariance", "ms_inverse", "ms_inverse_update",
                          "ms_score")},
            "theta": stats[-1]["theta"], "Meff": stats[-1]["Meff"], "thresh": stats[-1]["thresh"],
            "roofline": {
                "kernel": "k_group_update<false,true>: trailing update of the SPD inverse, f64 MFMA 128x128 tiles, three "
                          "pivots per launch at this size (K=384; a shorter last group adds one smaller launch)",
                "bound": "mfma",
                "achieved": achieved,
                "peak": PEAK_F64_MFMA_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_F64_MFMA_TFLOPS,
                "traffic": pmc_traffic(),
                "launches_per_step": upd_launch,
                "flops_per_launch": upd_flops / max(1, upd_launch),
                "avg_launch_ms": upd_ms / max(1, upd_launch),
            },
        }
        if world == 1:
            out["end_to_end_gdca_sec"] = end_to_end(Zh, q, args.score, pc, ctxs[0])
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(N, M, q, pc)
                out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
