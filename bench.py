"""bench.py -- the gDCA hot path on MI355X, BASELINE.json's headline configuration.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C|B|D|E] [--score frob|DI]

One "step" = one pass of the hot path (src/GaussDCA.jl:28-42 of the reference: reweighting ->
Pi/Pij -> covariance -> SPD inverse -> FN/DI -> APC) over one batch of synthetic input whose
int8 alignments are already resident in HBM; the N x N score matrices are left in HBM.

  --config C (default)  one family N=500, M=50k per rank per step          (BASELINE.json configs[2], the headline)
  --config B / D        one family N=128, M=10k (theta=0.2) / N=1000, M=100k per rank per step
  --config E            the 256-family Pfam-like batch (N in [100,600], M in [5k,80k]) sharded over the ranks by
                        the deterministic LPT rule of gaussdca.jl_amd/batch.py; one step = the whole batch once
                        (strong scaling: the batch is fixed, the ranks split it)

Multi-GPU: the path shards only across independent families -- one process and one gdca context per GPU, no
collective on the data path; torch.distributed (RCCL) is used for the timing barrier and the max-over-ranks
clock only.  `--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (child
`python -m torch.distributed.run`, before anything in this process touches a GPU) and exits with the child's code;
under an external launcher (WORLD_SIZE set) it is one of the ranks.

Prints ONE JSON line (rank 0).  `value` = families per second over all ranks.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

METRIC = "end-to-end gDCA sec + achieved Cholesky TFLOP/s, N=500 M=50k q=21"
PEAK_F64_MFMA_TFLOPS = 78.6  # AMD MI355X spec, FP64 matrix (not listed in MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0  # HBM3E peak bandwidth (MI355X_MICROARCH.md)
SPEC_SHADER_GHZ = 2.4  # the clock the spec peak is quoted at; the clock of the timed launches is measured by k_sweep itself
PROFILED_TRAFFIC = os.path.join(ROOT, "profiles", "r06_pmc_update_traffic.json")

CONFIGS = {  # name -> (N, M, theta, seed); None sizes = the batch
    "B": dict(N=128, M=10000, theta=0.2, seed=0xB128, ref="BASELINE.json configs[1]"),
    "C": dict(N=500, M=50000, theta=-1.0, seed=0xC500, ref="BASELINE.json configs[2]"),
    "D": dict(N=1000, M=100000, theta=-1.0, seed=0xD1000, ref="BASELINE.json configs[3]"),
    "E": dict(N=None, M=None, theta=-1.0, seed=0xE000, ref="BASELINE.json configs[4]"),
}


def synth_family(N, M, q, seed):
    """Deterministic 'Pfam-like' synthetic MSA (SplitMix64, native generator in libgdca.so),
    returned as (M, N) int8."""
    from gaussdca.jl_amd.synth import synth_family as gen

    return gen(N, M, q, seed)


def workload(cfg, args, rank, world):
    """The families this rank processes per step: list of (family_id, N, M, seed)."""
    from gaussdca.jl_amd.batch import batch_sizes, shard_families

    c = CONFIGS[cfg]
    if cfg == "E":
        sizes = batch_sizes(args.families, c["seed"])
        mine = shard_families(sizes, world)[rank]
        return [(f, sizes[f][0], sizes[f][1], c["seed"] + f) for f in mine]
    N = args.N or c["N"]
    M = args.M or c["M"]
    return [(rank, N, M, c["seed"] + rank)]  # weak scaling: every rank its own family of the same size


def pmc_traffic(N, M, score):
    """HBM bytes per launch of the dominant kernel.  NOT measured by this run (PMC counters need rocprofv3 passes of
    their own): the figure of the committed counter passes of this same command (FETCH_SIZE and WRITE_SIZE in separate
    passes, corrected with the factor measured on a copy kernel with the update kernel's load mix:
    profiles/r06_fetch_calibration.json), returned with its provenance; (None, None) unless the workload is the
    profiled one."""
    try:
        with open(PROFILED_TRAFFIC) as f:
            d = json.load(f)
        w = d.get("workload", {})
        if (w.get("N"), w.get("M"), w.get("score")) != (N, M, score):
            return None, None
        return float(d["hbm_bytes_per_launch"]), os.path.relpath(PROFILED_TRAFFIC, ROOT)
    except Exception:  # noqa: BLE001
        return None, None


def probe_reference_julia():
    """BASELINE.md 4.1 / SURVEY 8d: the preferred CPU baseline is the real reference (`julia -t N -e 'using GaussDCA'`).
    Probe for it and say what was found; the image (and the GPU box) have no Julia, so this records "absent" and the
    port is timed instead."""
    import shutil

    exe = shutil.which("julia")
    if not exe:
        return "absent (no `julia` on PATH)"
    try:
        v = subprocess.run([exe, "--version"], capture_output=True, text=True, timeout=60).stdout.strip()
        r = subprocess.run([exe, "-e", "using GaussDCA; print(1)"], capture_output=True, text=True, timeout=600)
        return "%s; using GaussDCA: %s" % (v, "ok (not timed: see cpu_baseline.kind)" if r.returncode == 0 else "fails")
    except Exception as e:  # noqa: BLE001
        return "probe failed: %s" % e


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


def cpu_baseline(N, M, q, pc, theta, score_name, seed, budget_s=60.0):
    """The oracle ("port": numpy + OpenMP/AVX2 C loops + OpenBLAS dpotrf/dpotri) timed on all host cores on the SAME
    family the GPU was timed on.  A probe (all-pairs pass on 20000 sequences, tallies on 6000, potrf+potri at n/4) estimates the cost;
    if the whole family fits the budget it is run IN FULL and every stage second below is measured, not scaled
    (config C: about 14 s on the GPU box's host).  Otherwise the two super-linear stages are run on a bounded
    sample and scaled, and the `sample` string says so.  A 1-thread run of the two OpenMP loops on a small sample
    gives the thread-scaling sanity ratios (they say how far the port is from using the host well; the port is a
    reported baseline, not a tuned CPU implementation)."""
    import numpy as np

    from oracle import gdca_oracle as o

    try:
        avail = len(os.sched_getaffinity(0))      # the cores this process may run on (affinity aware)
    except AttributeError:
        avail = os.cpu_count() or 1
    # ... capped by the cgroup CPU quota: the GPU box shows 256 hardware threads and grants "1600000 100000" in cpu.max = 16 CPUs'
    # worth of time; threads beyond that are throttled (which is why every probe below finds 16 threads fastest there)
    hw_threads = avail
    try:
        import gaussdca.jl_amd as _g

        avail = max(1, min(avail, int(_g.load().gdca_host_cpus())))
    except Exception:  # noqa: BLE001
        pass
    s = q - 1
    n = N * s
    t_all = time.time()
    Z = synth_family(N, M, q, seed)
    # How many threads to use is MEASURED, not assumed: on the GPU box's 256-thread host both OpenMP loops are fastest at 32
    # threads and 4-5x slower at 256 (tools/cpu_scaling.py, profiles/r03_cpu_scaling.log), so "all cores" would understate
    # the CPU.  Candidates: all, 128, 64, 32, 16; the fastest on a probe wins, for the OpenMP loops and for OpenBLAS
    # separately; `cores` in the JSON is what was used.
    cands = sorted({c for c in (avail, 128, 64, 32, 16, 8) if c <= avail} or {avail}, reverse=True)
    thr0 = o.hamming_threshold(0.3, N)
    Zs = Z[:min(M, 16000)]
    probe_omp = {}
    for c in cands:
        o.set_threads(c)
        o.neighbour_counts(Zs[:500], thr0)       # thread start-up outside the timed calls
        best = 1e9
        for _ in range(3):                       # best of three: a probe of some tens of milliseconds is noisy on a busy host
            t = time.time()
            o.neighbour_counts(Zs, thr0)
            best = min(best, time.time() - t)
        probe_omp[c] = best
    cores = min(probe_omp, key=probe_omp.get)
    o.set_threads(cores)
    # OpenBLAS: the thread count is chosen on the TIMED size, not on a small probe (an n/4 probe, 64x fewer flops,
    # picked 16 threads and the n = 10 000 inverse then ran at 0.65 TFLOP/s).  A first estimate on n/2 (8x fewer flops) ranks the
    # candidates; where the family is measured in full, potrf + potri then runs at full size on the best candidates and the stage
    # time is the minimum over them (`blas_inv_sec_at_n` in thread_choice says which counts were timed).
    probe_blas = {}
    blas_threads = avail
    threadpool_limits = None
    try:
        from threadpoolctl import threadpool_limits

        rng0 = np.random.default_rng(1)
        B0 = rng0.standard_normal((max(512, n // 2), 64))
        C0 = B0 @ B0.T / 64 + np.eye(B0.shape[0])
        for c in cands:
            with threadpool_limits(limits=c, user_api="blas"):
                o.spd_inverse(np.eye(64))
                t = time.time()
                o.spd_inverse(C0)
                probe_blas[c] = time.time() - t
        del B0, C0
        blas_threads = min(probe_blas, key=probe_blas.get)
        blas_limit = threadpool_limits(limits=blas_threads, user_api="blas")
    except Exception:  # noqa: BLE001
        blas_limit = None
    blas_full = {}
    th = o.compute_theta(Z) if theta < 0 else float(theta)
    thr = o.hamming_threshold(th, N)

    # probes, sized so that thread start-up and the n x n memsets do not dominate them
    Mp = min(M, 20000)
    o.neighbour_counts(Z[:min(M, 500)], thr)      # thread pool warm-up
    t = time.time()
    o.neighbour_counts(Z[:Mp], thr)
    ham_rate = (Mp * (Mp - 1.0) / 2) / max(1e-6, time.time() - t)  # pairs per second, all threads
    est_ham = (M * (M - 1.0) / 2) / ham_rate
    rng = np.random.default_rng(0)
    npb = max(256, n // 2)
    if probe_blas:
        est_inv = probe_blas[blas_threads] * (n / max(512, n // 2)) ** 3
    else:
        Bp = rng.standard_normal((npb, 64))
        Cp = Bp @ Bp.T / 64 + np.eye(npb)
        o.spd_inverse(np.eye(64))                      # LAPACK import / thread start-up outside the probe
        t = time.time()
        o.spd_inverse(Cp)
        est_inv = (time.time() - t) * (n / npb) ** 3
    Mf = min(M, 6000)
    t = time.time()
    o.compute_frequencies(Z[:Mf], q, np.ones(Mf), float(Mf))
    t_f1 = time.time() - t
    t = time.time()
    o.compute_frequencies(Z[:Mf // 3], q, np.ones(Mf // 3), float(Mf // 3))
    t_f0 = time.time() - t                         # the same call on a third of the sequences: separates the fixed
    freq_all = (t_f1 - t_f0) / (Mf - Mf // 3)      # n x n memset / mirror cost from the per-sequence cost
    if freq_all <= 0:
        freq_all = t_f1 / Mf
    # thread-scaling sanity: the same two loops on one thread
    Ms = min(M, 4000)
    o.set_threads(1)
    t = time.time()
    o.neighbour_counts(Z[:Ms], thr)
    ham_rate_1 = (Ms * (Ms - 1.0) / 2) / max(1e-6, time.time() - t)
    Mg = min(M, 1500)
    t = time.time()
    o.compute_frequencies(Z[:Mg], q, np.ones(Mg), float(Mg))
    t_g1 = time.time() - t
    t = time.time()
    o.compute_frequencies(Z[:Mg // 3], q, np.ones(Mg // 3), float(Mg // 3))
    freq_1 = (t_g1 - (time.time() - t)) / (Mg - Mg // 3)
    if freq_1 <= 0:
        freq_1 = t_g1 / Mg
    o.set_threads(cores)

    full = est_ham + est_inv + freq_all * M * 1.2 < budget_s
    stage = {}
    if full:
        t = time.time()
        n_k = o.neighbour_counts(Z, thr)
        stage["ham"] = time.time() - t
        W, Meff = o.weights_from_counts(n_k)
        t = time.time()
        Pi, Pij = o.compute_frequencies(Z, q, W, Meff)
        stage["freq"] = time.time() - t
        t = time.time()
        Pi2, Pij2 = o.add_pseudocount(Pi, Pij, pc, q)
        C = o.compute_C(Pi2, Pij2)
        stage["cov"] = time.time() - t
        del Pi, Pij, Pij2
        # potrf + potri on the family's own covariance with the best candidate thread counts of the n/2 estimate: the stage time
        # is the fastest of them
        if threadpool_limits is not None and probe_blas:
            # (the two best of the n/2 ranking always -- on the GPU box it put 8 threads ahead of 16 and the full size had it the
            # other way round, 2.03 s against 1.53 s -- and a third if it was within 1.3x)
            ranked = sorted(probe_blas, key=probe_blas.get)
            tryc = ranked[:2] + [c for c in ranked[2:3] if probe_blas[c] <= 1.3 * probe_blas[ranked[0]]]
            mJ = None
            for c in tryc:
                with threadpool_limits(limits=c, user_api="blas"):
                    t = time.time()
                    mJ = o.spd_inverse(C)
                    blas_full[c] = time.time() - t
            blas_threads = min(blas_full, key=blas_full.get)
            stage["inv"] = blas_full[blas_threads]
        else:
            t = time.time()
            mJ = o.spd_inverse(C)
            stage["inv"] = time.time() - t
        t = time.time()
        S = o.compute_DI_gauss(mJ, C, q) if score_name == "DI" else o.compute_FN(mJ, q)
        o.correct_APC(S)
        stage["score"] = time.time() - t
        how = "the whole family, every stage measured in full"
        # the all-thread Hamming rate of the sanity ratio comes from the full-size stage (a probe is start-up-bound on 256 threads)
        ham_rate = (M * (M - 1.0) / 2) / max(1e-6, stage["ham"])
    else:
        f_h = min(1.0, (0.4 * budget_s / est_ham) ** 0.5)
        Mh = max(Mp, int(M * f_h))
        t = time.time()
        n_k = o.neighbour_counts(Z[:Mh], thr)
        stage["ham"] = (time.time() - t) * (M * (M - 1.0)) / (Mh * (Mh - 1.0))
        W, Meff = o.weights_from_counts(n_k)
        Mt = min(Mh, max(4000, int(0.15 * budget_s / max(freq_all, 1e-9))))
        t = time.time()
        Pi, Pij = o.compute_frequencies(Z[:Mt], q, W[:Mt], float(W[:Mt].sum()))
        stage["freq"] = (time.time() - t) * (M / Mt)
        t = time.time()
        Pi2, Pij2 = o.add_pseudocount(Pi, Pij, pc, q)
        C = o.compute_C(Pi2, Pij2)
        stage["cov"] = time.time() - t
        ns = n if est_inv < 0.4 * budget_s else max(npb, int(n * (0.4 * budget_s / est_inv) ** (1 / 3)))
        t = time.time()
        mJs = o.spd_inverse(np.ascontiguousarray(C[:ns, :ns]))
        stage["inv"] = (time.time() - t) * (n / ns) ** 3
        mJ = mJs if ns == n else C
        t = time.time()
        o.correct_APC(o.compute_FN(mJ, q))
        stage["score"] = time.time() - t
        how = ("bounded sample, the two super-linear stages SCALED: Hamming on %d of %d sequences (x pair ratio), "
               "tallies on %d (x M ratio), potrf+potri at n=%d of %d (x n^3 ratio)" % (Mh, M, Mt, ns, n))
    sec = sum(stage.values())
    if blas_limit is not None:
        blas_limit.restore_original_limits()
    inv_flops = (n ** 3 / 3.0 + n * n / 2.0 + n / 6.0) + (2.0 * n ** 3 / 3.0 + n * n / 2.0 + 5.0 * n / 6.0)
    return dict(value=1.0 / sec, unit="families/s", cores=max(cores, blas_threads),
                threads={"omp": cores, "blas": blas_threads}, host_threads_available=avail, host_hardware_threads=hw_threads,
                thread_choice={"openmp_probe_sec": {str(k): round(v, 4) for k, v in probe_omp.items()},
                               "blas_probe_sec_at_half_n": {str(k): round(v, 4) for k, v in probe_blas.items()},
                               "blas_inv_sec_at_n": {str(k): round(v, 4) for k, v in blas_full.items()}},
                kind="port", measured_in_full=bool(full),
                sec_per_family=sec, stage_sec={k: round(v, 4) for k, v in stage.items()},
                inv_tflops=inv_flops / max(stage["inv"], 1e-9) / 1e12,
                thread_scaling={"hamming_pairs_per_s_1_thread": ham_rate_1, "hamming_pairs_per_s_all": ham_rate,
                                "hamming_speedup": ham_rate / ham_rate_1,
                                "tally_s_per_seq_1_thread": freq_1, "tally_s_per_seq_all": freq_all,
                                "tally_speedup": freq_1 / max(freq_all, 1e-12)},
                sample=("oracle (numpy + OpenMP/AVX2 C + OpenBLAS potrf/potri; NOT the Julia reference, which cannot run "
                        "here) on %d OpenMP / %d OpenBLAS threads (the fastest of the probed counts; %d available), "
                        "N=%d M=%d q=%d score=%s: %s; wall incl. probes %.1fs"
                        % (cores, blas_threads, avail, N, M, q, score_name, how, time.time() - t_all)))


def launch_ranks(n):
    """`--gpus n` without an external launcher: start the n ranks as a child torch.distributed.run and hand back
    its exit code.  Called before this process has imported torch or touched a GPU (a process that has initialised
    the GPU must never replace itself or fork workers)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["GDCA_BENCH_CHILD"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 40 for B / C, 10 for D, 2 for E: about a "
                    "second or more of GPU time, long enough not to be a clock-ramp artefact)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default: 5, 2, 1)")
    ap.add_argument("--config", default="C", choices=sorted(CONFIGS))
    ap.add_argument("--score", default="frob")
    ap.add_argument("--N", type=int, default=0, help="override the config's N (configs B, C, D)")
    ap.add_argument("--M", type=int, default=0)
    ap.add_argument("--q", type=int, default=21)
    ap.add_argument("--families", type=int, default=256, help="--config E: families in the batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default run (config C, :frob): skip the lines of :DI, B, D and the E prefix (`other_configs`)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: every rank derives its shard, the ranks meet at the (gloo) barrier and rank 0 prints "
                         "the JSON line with the shards instead of timings (used by the CPU tests of the N > 1 path)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="independent families in flight per GPU (one gdca context each): family f+1's "
                         "reweighting/tallies (VALU, LDS) overlap family f's SPD inverse (MFMA); 1 = strictly serial")
    ap.add_argument("--gate", action="store_true",
                    help="with --pipeline > 1: let the SPD-inverse stages of the families in flight take turns")
    ap.add_argument("--phased-one-set", action="store_true",
                    help="with --phased: one set of P contexts instead of two alternating ones (no batch is enqueued while another runs)")
    ap.add_argument("--dump-families", default="",
                    help="write the timed runs' per-family stage times (N, M, ms_*) to this JSON file (tools/fit_batch_model.py, "
                         "tests/test_batch_sharding.py: the LPT cost model against measured times)")
    ap.add_argument("--phased", action="store_true",
                    help="with --pipeline K: batch the K families in flight BY PHASE on one stream (gdca_run_dev_phased): K front "
                         "ends, then K SPD inverses back to back, then K score stages -- the matrix pipes see one load step per "
                         "K families instead of one per family (throughput form; --pipeline 1 stays the latency headline)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = {"B": 40, "C": 40, "D": 10, "E": 2}[args.config]
    if args.warmup is None:
        args.warmup = {"B": 5, "C": 5, "D": 2, "E": 1}[args.config]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    q = args.q
    fams = workload(args.config, args, rank, world)

    import torch

    dist = None
    comm_fallback = None
    if args.dry_run:
        if world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
            gathered = [None] * world
            dist.all_gather_object(gathered, [f[0] for f in fams])
            tt = torch.tensor([1.0 + rank], dtype=torch.float64)
            dist.barrier()
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tmax = float(tt.item())
            dist.destroy_process_group()
        else:
            gathered, tmax = [[f[0] for f in fams]], 1.0
        if rank == 0:
            print(json.dumps({"metric": METRIC, "dry_run": True, "n_gpus": world, "config": {"workload": args.config},
                              "shards": gathered, "max_over_ranks": tmax,
                              "comm": {"backend": "gloo" if world > 1 else None, "world_size": world,
                                       "data_path_collectives": 0},
                              "scaling": "strong" if args.config == "E" else "weak"}))
        return

    import gaussdca.jl_amd as g

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the gDCA hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL ("nccl" on ROCm) for the timing barrier / max-over-ranks only; no collective on the data path.
        # GDCA_BENCH_BACKEND=gloo selects gloo explicitly.  If the RCCL communicator cannot be created the measurement
        # still runs over gloo -- loudly: a warning on stderr and `comm.fallback_reason` in the JSON line, which always
        # names the backend and the world size it really ran with.
        backend = os.environ.get("GDCA_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=dev)
            except Exception as e:  # noqa: BLE001
                comm_fallback = "%s: %s" % (type(e).__name__, str(e)[:300])
                print("bench.py: RCCL init failed (%s); timing barrier over gloo instead" % comm_fallback, file=sys.stderr)
                dist.init_process_group("gloo")
        else:
            dist.init_process_group(backend)

    meas = measure(args, args.config, args.score, args.steps, args.warmup, rank, world, local, dev, dist)
    out = summarize(args, meas, world, dist, comm_fallback)  # (every rank: it holds the reductions)
    if args.dump_families and rank == 0:
        keys = ("N", "M", "n", "ms_total", "ms_theta", "ms_weights", "ms_covariance", "ms_inverse", "ms_score", "inverse_batch")
        with open(args.dump_families, "w") as f:
            json.dump({"workload": out["config"]["workload"], "schedule": out["config"]["schedule"],
                       "families": [{k: s[k] for k in keys} for s in meas["stats"]]}, f, indent=0)
    if rank != 0:
        release(meas)
    else:
        cfg = CONFIGS[args.config]
        N0, M0 = meas["fams"][0][1], meas["fams"][0][2]
        if world == 1 and args.config != "E":
            out["end_to_end_gdca_sec"] = end_to_end(meas["Zh"][0], q, args.score, meas["pc"], meas["ctxs"][0])
        release(meas)
        # every other single-GPU configuration of BASELINE.json in the same run: :DI at the headline size, B
        # (alone and eight at a time through the merged sweep), D and a 32-family prefix of the batch E
        if world == 1 and args.config == "C" and args.score == "frob" and not args.no_other_configs and not (args.N or args.M):
            others = {}
            plan = (("C_DI", "C", "DI", 20, 3, 1, False, 256), ("B", "B", "frob", 40, 5, 1, False, 256),
                    ("B_merged8", "B", "frob", 80, 8, 8, True, 256), ("D", "D", "frob", 5, 1, 1, False, 256),
                    ("E32", "E", "frob", 2, 1, 1, False, 32), ("E32_p2", "E", "frob", 2, 1, 2, False, 32),
                    ("E32_phased8", "E", "frob", 2, 1, 8, True, 32),
                    # the WHOLE batch of BASELINE.json configs[4] on this one GPU, best schedule (phase batches of sixteen, the small
                    # inverses merged, the batch's kernels as batched grids): one warm-up pass and one timed pass, ~7 s
                    ("E256", "E", "frob", 1, 1, 16, True, 256))
            for key, cname, sc, st_, wu, P_, ph_, nf in plan:
                try:
                    a2 = argparse.Namespace(**vars(args))
                    a2.pipeline, a2.phased, a2.gate, a2.families, a2.N, a2.M = P_, ph_, False, nf, 0, 0
                    m2 = measure(a2, cname, sc, st_, wu, 0, 1, local, dev, None)
                    o2 = summarize(a2, m2, 1, None, None)
                    if cname not in ("E",):
                        o2["end_to_end_gdca_sec"] = end_to_end(m2["Zh"][0], q, sc, m2["pc"], m2["ctxs"][0]) if key in ("C_DI", "D") else None
                    release(m2)
                    others[key] = {k: o2[k] for k in ("value", "unit", "steps", "warmup", "ms_per_step", "sec_per_family", "stage_ms",
                                                      "spd_inverse_tflops", "end_to_end_gdca_sec") if k in o2}
                    others[key]["workload"] = o2["config"]["workload"]
                    others[key]["schedule"] = o2["config"]["schedule"]
                    others[key]["roofline"] = {k: o2["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "avg_launch_ms",
                                                                             "launches_per_step", "measured_shader_ghz",
                                                                             "frac_of_attainable_at_measured_clock")}
                    others[key]["roofline_hbm"] = o2.get("roofline_hbm", [])
                except Exception as e:  # noqa: BLE001
                    others[key] = {"error": "%s: %s" % (type(e).__name__, e)}
            out["other_configs"] = others
        if world == 1 and not args.no_cpu_baseline and args.config != "E":
            out["reference_julia"] = probe_reference_julia()
            try:
                out["cpu_baseline"] = cpu_baseline(N0, M0, q, meas["pc"], cfg["theta"], args.score, meas["fams"][0][3])
                out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def release(meas):
    """Free a measurement's contexts and device buffers before the next configuration is set up."""
    for c in meas["ctxs"] + meas["ctxs2"]:
        c.close()
    meas["ctxs"], meas["ctxs2"] = [], []
    meas["Zd"] = meas["Sd"] = meas["Sd2"] = None
    import torch

    torch.cuda.empty_cache()


def measure(args, config, score_name, steps, warmup, rank, world, local, dev, dist):
    """W untimed and exactly K timed steps of `config` on this rank's GPU, bracketed by barrier + synchronize on both sides."""
    import numpy as np  # noqa: F401
    import torch

    import gaussdca.jl_amd as g

    q = args.q
    score = 1 if score_name == "DI" else 0
    pc = 0.2 if score == 1 else 0.8
    cfg = CONFIGS[config]
    fams = workload(config, args, rank, world)
    P = max(1, args.pipeline)
    phased = bool(args.phased and P > 1)
    if config == "E" and phased:
        # batches of P neighbours in this order: families of similar covariance size side by side, so that the small ones of
        # a batch share merged sweep launches (any processing order is as good as another: the families are independent)
        fams = sorted(fams, key=lambda f: (-f[1], f[0]))
    # synthetic families, resident in HBM before the timed region ((M, N) int8 == Julia's N x M column-major bytes)
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        Zh = list(ex.map(lambda f: synth_family(f[1], f[2], q, f[3]), fams))
    Zd = [torch.from_numpy(z).to(dev) for z in Zh]
    if config == "E":
        Zh = Zh[:1]
    nmax = max(f[1] for f in fams)
    Sd = [torch.empty((nmax, nmax), dtype=torch.float64, device=dev) for _ in range(P)]  # stay in HBM
    ctxs = [g.Context(local)]
    # (phase batches: every context of both alternating sets is a peer of ONE pipeline -- its batches are then one in-order sequence
    # on the pipeline's stream: no hand-over between two streams that may share a hardware queue, 400 us at config B)
    for _ in range(P - 1):
        ctxs.append(ctxs[0].peer() if (args.gate or phased) else g.Context(local))
    busy = [False] * P
    ctxs2, Sd2 = [], []
    if phased:
        ctxs2 = [ctxs[0].peer() for _ in range(P)]
        Sd2 = [torch.empty((nmax, nmax), dtype=torch.float64, device=dev) for _ in range(P)]

    def run_steps_phased(count, sink):
        """`count` steps, the rank's families taken P at a time and batched by phase (gdca_run_dev_phased).  Two sets of P
        contexts alternate: the next batch is enqueued (on the other set's stream) before the previous one is collected, so
        the GPU never waits for the host to enqueue a batch."""
        work = [fi for _ in range(count) for fi in range(len(fams))]
        pending = None
        for b, a in enumerate(range(0, len(work), P)):
            grp = work[a:a + P]
            alt = b % 2 == 1 and not args.phased_one_set
            cset = (ctxs2 if alt else ctxs)[:len(grp)]
            sset = (Sd2 if alt else Sd)[:len(grp)]
            if args.phased_one_set and pending is not None:   # one set of contexts: the batch before must be collected first
                sink.extend(c.collect() for c in pending)
                pending = None
            g.run_dev_phased(cset, [Zd[fi].data_ptr() for fi in grp], [fams[fi][1] for fi in grp],
                             [fams[fi][2] for fi in grp], [q] * len(grp), pc, cfg["theta"], score,
                             [x.data_ptr() for x in sset])
            if pending is not None:
                sink.extend(c.collect() for c in pending)
            pending = cset
        if pending is not None:
            sink.extend(c.collect() for c in pending)

    def run_steps(count, sink):
        """`count` steps; inside a step the rank's families go round-robin over the P contexts; a context's previous
        pass is collected (stream sync + stats) right before it is given the next one."""
        if phased:
            return run_steps_phased(count, sink)
        t = 0
        for _ in range(count):
            for fi, (_, N, M, _) in enumerate(fams):
                c = t % P
                t += 1
                if busy[c]:
                    sink.append(ctxs[c].collect())
                ctxs[c].run_dev_async(Zd[fi].data_ptr(), N, M, q, pc, cfg["theta"], score, Sd[c].data_ptr())
                busy[c] = True
        for c in range(P):
            if busy[c]:
                sink.append(ctxs[c].collect())
                busy[c] = False

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        for c in ctxs + ctxs2:
            c.synchronize()

    # every context is warmed (workspace allocation: hipMalloc synchronises the device) before the clock starts
    warm = max(warmup, 1, -(-(2 * P if phased else P) // max(1, len(fams))))
    run_steps(warm, [])
    barrier()
    t0 = time.perf_counter()
    stats = []
    run_steps(steps, stats)
    barrier()
    dt = time.perf_counter() - t0
    return dict(config=config, score_name=score_name, pc=pc, fams=fams, Zh=Zh, Zd=Zd, Sd=Sd, Sd2=Sd2, ctxs=ctxs, ctxs2=ctxs2,
                stats=stats, dt=dt, steps=steps, warm=warm, P=P, phased=phased)


def pmc_kernel_traffic(N, M, score, prefix):
    """HBM bytes per launch of the kernel whose name starts with `prefix`, from the committed counter passes (see pmc_traffic):
    None unless the workload is the profiled one and the file holds that kernel."""
    try:
        with open(PROFILED_TRAFFIC) as f:
            d = json.load(f)
        w = d.get("workload", {})
        if (w.get("N"), w.get("M"), w.get("score")) != (N, M, score):
            return None
        hit = [v["hbm_bytes_per_launch"] for k, v in d.get("per_kernel", {}).items() if k.startswith(prefix)]
        return float(max(hit)) if hit else None
    except Exception:  # noqa: BLE001
        return None


def hbm_records(stats, q, traffic_of=None):
    """`roofline` objects (bound "hbm") of the HBM-bound kernels of the timed runs: FN and the covariance write.
    traffic_of(prefix): HBM bytes per launch of that kernel from the committed counter passes, or None."""
    import numpy as np

    out = []
    traffic_of = traffic_of or (lambda prefix: None)
    s_dim = q - 1
    fn = [(8.0 * st["n"] * (st["n"] - s_dim) / 2.0, st["ms_fn"]) for st in stats if st.get("ms_fn", 0.0) > 0.0]
    if fn:
        by, ms = float(np.sum([b for b, _ in fn])), float(np.sum([m for _, m in fn]))
        ach = by / (ms * 1e-3) / 1e9
        out.append({"kernel": "k_fn (compute_FN: one pass over the lower block triangle of the inverse)", "bound": "hbm", "achieved": ach,
                    "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "bytes_per_launch": by / len(fn),
                    "avg_launch_ms": ms / len(fn), "traffic": traffic_of("k_fn")})
    pt = [(8.0 * st["n"] * st["n"], st["ms_pair_tally"]) for st in stats if st.get("ms_pair_tally", 0.0) > 0.0]
    if pt:
        by, ms = float(np.sum([b for b, _ in pt])), float(np.sum([m for _, m in pt]))
        ach = by / (ms * 1e-3) / 1e9
        out.append({"kernel": "k_pair_tally (pair tallies + pseudocount + covariance epilogue: C written once, 8 n^2 bytes; the kernel is "
                              "bound by its LDS atomics)", "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": ach / PEAK_HBM_GBS, "bytes_per_launch": by / len(pt), "avg_launch_ms": ms / len(pt), "traffic": traffic_of("k_pair_tally")})
    return out


def summarize(args, meas, world, dist, comm_fallback):
    """Rank 0's JSON object of one measurement (all ranks take part in the reductions)."""
    import numpy as np
    import torch

    q = args.q
    cfg = CONFIGS[meas["config"]]
    config, score_name, pc = meas["config"], meas["score_name"], meas["pc"]
    fams, stats, dt, P = meas["fams"], meas["stats"], meas["dt"], meas["P"]
    nfam_local = len(fams) * meas["steps"]
    nfam = nfam_local
    flops_local = float(sum(s["inverse_flops"] for s in stats))
    flops = flops_local
    if dist is not None:
        dev = torch.device("cuda", torch.cuda.current_device())
        on = dev if dist.get_backend() == "nccl" else "cpu"
        tt = torch.tensor([dt], dtype=torch.float64, device=on)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        cnt = torch.tensor([float(nfam_local), flops_local], dtype=torch.float64, device=on)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        nfam, flops = int(round(float(cnt[0].item()))), float(cnt[1].item())
    K = meas["steps"]
    ms_step = dt / K * 1e3
    upd_ms = float(np.sum([s["ms_inverse_update"] for s in stats]))
    upd_launch = int(np.sum([s["update_launches"] for s in stats]))
    upd_flops = float(np.sum([s["update_flops"] for s in stats]))
    inv_ms = float(np.sum([s["ms_inverse"] for s in stats]))
    alg_flops = float(np.sum([s["inverse_flops"] for s in stats]))  # F = n^3 + n^2 + n per family
    achieved = alg_flops / (upd_ms * 1e-3) / 1e12 if upd_ms > 0 else 0.0
    N0, M0 = fams[0][1], fams[0][2]
    ghz = float(np.mean([s["sweep_ghz"] for s in stats]))
    traffic, traffic_src = pmc_traffic(N0, M0, score_name) if config != "E" else (None, None)
    merged = sorted({int(s["inverse_batch"]) for s in stats})
    if config == "E":
        wl = ("batch of %d synthetic Pfam-like families, N in [100,600], M in [5k,80k], q=%d, score=:%s, "
              "theta=:auto, pseudocount=%.1f, LPT-sharded over %d rank(s) (%s)"
              % (args.families, q, score_name, pc, world, cfg["ref"]))
    else:
        wl = ("synthetic Pfam-like MSA N=%d M=%d q=%d, score=:%s, theta=%s, pseudocount=%.1f (%s)"
              % (N0, M0, q, score_name, ":auto" if cfg["theta"] < 0 else repr(cfg["theta"]), pc, cfg["ref"]))
    kernel = ("k_sweep (the whole SPD inverse as ONE persistent launch: block symmetric sweep, f64 MFMA 128x128 "
              "tiles, pivot chain on elected CUs; duration from HIP events on its stream)")
    if merged != [1]:
        kernel = ("k_sweep / k_sweep_merged (one persistent launch per SPD inverse, or per group of up to %d small inverses sharing a "
                  "launch: block symmetric sweep, f64 MFMA 128x128 tiles; duration from HIP events on its stream)" % max(merged))
    out = {
        "metric": METRIC,
        "value": nfam / dt,
        "unit": "families/s",
        "n_gpus": world,
        "steps": K,
        "warmup": meas["warm"],
        "ms_per_step": ms_step,
        "higher_is_better": True,
        "scaling": "strong" if config == "E" else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": wl, "name": config, "q": q,
                   "families_per_step": nfam // K, "families_per_step_rank0": len(fams),
                   "families_in_flight_per_gpu": P,
                   "inverses_per_sweep_launch": merged,
                   "schedule": ("phase-batched: %d front ends, the %d inverses (the small ones merged into shared launches), %d score stages"
                                % (P, P, P)) if meas["phased"] else ("one family after the other" if P == 1 else "independent streams")},
        "sec_per_family": dt / (nfam / world) if config != "E" else dt / nfam,
        "aggregate_inverse_tflops": flops / dt / 1e12,
        "latency_ms_per_family": float(np.mean([s["ms_total"] for s in stats])),
        "spd_inverse_tflops": float(np.sum([s["inverse_flops"] for s in stats])) / (inv_ms * 1e-3) / 1e12,
        "stage_ms": {k: float(np.mean([s[k] for s in stats])) for k in
                     ("ms_total", "ms_theta", "ms_weights", "ms_covariance", "ms_inverse", "ms_inverse_update",
                      "ms_score")},
        # inverses of the timed region that were run AGAIN because the sweep's watchdog had ended their launch (normally 0: DESIGN.md 3.1a)
        "sweep_retries": int(np.sum([s.get("sweep_retries", 0) for s in stats])),
        "roofline": {
            "kernel": kernel,
            "bound": "mfma",
            "achieved": achieved,
            "peak": PEAK_F64_MFMA_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / PEAK_F64_MFMA_TFLOPS,
            "traffic": traffic,
            "traffic_source": traffic_src,  # a committed rocprofv3 counter pass of this command, not this run
            "launches_per_step": upd_launch / K,
            "flops_per_launch": alg_flops / max(1, upd_launch),
            "mfma_flops_issued_per_launch": upd_flops / max(1, upd_launch),  # incl. the padding to 128-blocks
            "avg_launch_ms": upd_ms / max(1, upd_launch),
            # shader clock of the timed launches, measured by the kernel (s_memtime / 100 MHz wall clock of one
            # workgroup, gdca_stats.sweep_ghz), and what the matrix pipes could deliver at that clock
            "measured_shader_ghz": ghz,
            "attainable_at_measured_clock": PEAK_F64_MFMA_TFLOPS * ghz / SPEC_SHADER_GHZ,
            "frac_of_attainable_at_measured_clock": (achieved / (PEAK_F64_MFMA_TFLOPS * ghz / SPEC_SHADER_GHZ)) if ghz > 0 else None,
        },
        # the HBM-bound stages: algorithmic bytes / the kernel's own duration (HIP events around that kernel on its
        # stream, gdca_stats.ms_fn / ms_pair_tally) against the 8 TB/s HBM3E peak.  FN: one pass over the lower block triangle of
        # the inverse, 8 n (n - s) / 2 bytes.  The covariance build's compulsory write (8 n^2 bytes) is listed against the pair-tally
        # kernel that contains it -- a kernel bound by its LDS atomics, not by that write: the fraction says how far from HBM it is.
        "roofline_hbm": hbm_records(stats, q, (lambda pre: pmc_kernel_traffic(N0, M0, score_name, pre)) if config != "E" else None),
        "comm": {"backend": (dist.get_backend() if dist is not None else None), "world_size": world,
                 "data_path_collectives": 0, "fallback_reason": comm_fallback},
    }
    if config != "E":
        out["config"].update({"N": N0, "M": M0, "n": N0 * (q - 1)})
        out.update({"theta": stats[-1]["theta"], "Meff": stats[-1]["Meff"], "thresh": stats[-1]["thresh"]})
    return out


if __name__ == "__main__":
    main()
