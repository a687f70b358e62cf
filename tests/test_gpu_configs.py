"""GPU parity tests at BASELINE.json's own configurations (run with `-m gpu` on the MI355X box), all through the
C-ABI (libgdca.so via ctypes) against the CPU oracle:

  B  N=128,  M=10k,  theta=0.2, :frob              full oracle comparison
  C  N=500,  M=50k,  theta=:auto, :frob AND :DI    full oracle comparison at the headline size (one oracle pass)
  D  N=1000, M=100k, :frob                         full oracle comparison (LAPACK inverse at n = 20 000) + the properties of the `_dev` chain
  E  16 families of the 256-family batch through `gdca_cli --batch`, byte-equal to the single-family path; four of them
     (smallest, largest, two mid-size) against the oracle

Bars (north_star): Hamming counts / thresholds / identity sums / ranking indices bit-exact, FN / DI scores within
1e-6 relative.  Tolerances are written at each assert.  Seeds are SURVEY.md 8d's (0xB128, 0xC500, 0xD1000, 0xE000+f).
"""
import os
import subprocess

import math
import numpy as np
import pytest

from gdca_testutil import score_close

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "gaussdca.jl_amd", "gdca_cli")


@pytest.fixture(scope="module")
def g():
    import gaussdca.jl_amd as g

    assert os.path.exists(g._lib.LIB_PATH), "libgdca.so missing: the GPU tests never fall back to the CPU"
    assert g.load().gdca_device_count() > 0, "no HIP device"
    return g


@pytest.fixture(scope="module")
def ctx(g):
    c = g.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def o():
    from oracle import gdca_oracle as o

    o.set_threads(min(os.cpu_count() or 1, 32))   # the OpenMP loops peak at 32 threads on the GPU box (tools/cpu_scaling.py)
    return o


def _synth(N, M, seed):
    from gaussdca.jl_amd import synth

    return synth.synth_family(N, M, 21, seed)


def _top_order_equal(g, o, S, S_o, top=200):
    """The first `top` pairs in the oracle's order, and the WHOLE ranking as an ordering of the oracle's scores: the same set of
    pairs, and walking down the device's ranking the oracle's score never rises by more than the 1e-6 bar allows -- i.e. the
    two rankings can differ only by swaps among pairs whose scores the bar does not separate (VERDICT r03 weak spot 1b)."""
    R, R_o = g.compute_ranking(S, 5), o.compute_ranking(S_o, 5)
    if [t[:2] for t in R[:top]] != [t[:2] for t in R_o[:top]]:
        return False
    ii, jj = np.asarray(R.i, dtype=np.int64) - 1, np.asarray(R.j, dtype=np.int64) - 1
    if sorted(zip(ii.tolist(), jj.tolist())) != sorted((t[0] - 1, t[1] - 1) for t in R_o):
        return False
    along = S_o[jj, ii]                                   # the oracle's scores in the device's order
    rises = along[1:] - along[:-1]
    return bool(rises.max() <= 2e-6 * float(np.abs(S_o).max()))


# ---- B -------------------------------------------------------------------------------------------------------------
def test_config_B_full_oracle_comparison(g, ctx, o):
    """BASELINE.json configs[1]: N=128, M=10k, q=21, :frob, theta=0.2 (fixed), pseudocount 0.8; 2560 x 2560 covariance
    (20 pivot blocks: the single-pivot look-ahead schedule)."""
    N, M, q, theta, pc = 128, 10000, 21, 0.2, 0.8
    Zo = _synth(N, M, 0xB128)
    Z = np.asfortranarray(Zo.T)
    W_o, Meff_o, th_o, thr_o = o.compute_weights(Zo, theta)
    assert thr_o == 25
    assert np.array_equal(g.neighbour_counts(Z, thr_o, ctx=ctx), o.neighbour_counts(Zo, thr_o))   # bit-exact
    S, st = ctx.run(Z, q, pc, theta, 0)
    assert st["theta"] == theta and st["thresh"] == thr_o and st["Meff"] == Meff_o and st["info"] == 0
    S_o = o.scores_from_Z(Zo, q, pc, theta, "frob")
    ok, max_rel, max_abs = score_close(S, S_o, rtol=1e-6, atol_frac=1e-9)    # 1e-6 relative
    assert ok, (max_rel, max_abs)
    assert np.array_equal(S, S.T)
    R, R_o = g.compute_ranking(S, 5), o.compute_ranking(S_o, 5)
    assert [t[:2] for t in R] == [t[:2] for t in R_o] or _top_order_equal(g, o, S, S_o, 500)
    # the statement-by-statement device-resident chain gives the same bits as the fused call
    from gaussdca.jl_amd import devops

    S2, info = devops.scores_stepwise(Z, q, pc, theta, "frob", ctx=ctx)
    assert np.array_equal(S2, S) and info["Meff"] == Meff_o and info["thresh"] == thr_o


# ---- C: the headline configuration, compared with the oracle at its own size -----------------------------------------
@pytest.fixture(scope="module")
def headline(o):
    """One oracle pass over the headline family shared by the :frob and :DI comparisons (they differ only from the
    pseudocount on): theta, threshold, neighbour counts, W, Meff, Pi_true, Pij_true."""
    N, M, q = 500, 50000, 21
    Zo = _synth(N, M, 0xC500)
    theta = o.compute_theta(Zo)
    thr = o.hamming_threshold(theta, N)
    n_k = o.neighbour_counts(Zo, thr)
    W, Meff = o.weights_from_counts(n_k)
    Pi_t, Pij_t = o.compute_frequencies(Zo, q, W, Meff)
    return dict(N=N, M=M, q=q, Zo=Zo, theta=theta, thr=thr, n_k=n_k, W=W, Meff=Meff, Pi_t=Pi_t, Pij_t=Pij_t,
                pair_sum=o.pair_identity_sum(Zo))


def test_config_C_integers_bit_exact_at_full_size(g, ctx, headline):
    h = headline
    Z = np.asfortranarray(h["Zo"].T)
    assert g.pair_identity_sum(Z, ctx=ctx) == h["pair_sum"]
    assert np.array_equal(g.neighbour_counts(Z, h["thr"], ctx=ctx), h["n_k"])        # all 50 000 counts, bit-exact
    W, Meff, th, thr = g.compute_weights(Z, h["q"], "auto", ctx=ctx, return_theta=True)
    assert th == h["theta"] and thr == h["thr"] and Meff == h["Meff"] and np.array_equal(W, h["W"])


@pytest.mark.parametrize("score,pc", [("frob", 0.8), ("DI", 0.2)])
def test_config_C_scores_match_oracle_at_full_size(g, ctx, o, headline, score, pc):
    """BASELINE.json configs[2], both scores: N=500, M=50k, theta=:auto; 10 000 x 10 000 covariance, 79 pivot
    blocks (groups of three pivots, K=384 trailing updates): S within 1e-6 relative of the oracle's (LAPACK
    potrf+potri inverse), top-200 contact order identical."""
    h = headline
    N, q = h["N"], h["q"]
    Z = np.asfortranarray(h["Zo"].T)
    S, st = ctx.run(Z, q, pc, -1.0, 1 if score == "DI" else 0)
    assert st["theta"] == h["theta"] and st["thresh"] == h["thr"] and st["Meff"] == h["Meff"] and st["info"] == 0
    assert st["pair_identity_sum"] == h["pair_sum"]
    Pi, Pij = o.add_pseudocount(h["Pi_t"], h["Pij_t"], pc, q)
    C = o.compute_C(Pi, Pij)
    del Pij
    mJ = o.spd_inverse(C)
    S_o = o.compute_DI_gauss(mJ, C, q) if score == "DI" else o.compute_FN(mJ, q)
    S_o = o.correct_APC(S_o)
    ok, max_rel, max_abs = score_close(S, S_o, rtol=1e-6, atol_frac=1e-9)    # 1e-6 relative
    assert ok, (score, max_rel, max_abs)
    assert _top_order_equal(g, o, S, S_o, 200)
    assert np.array_equal(S, S.T)


# ---- D -------------------------------------------------------------------------------------------------------------
def test_config_D_properties_at_full_size(g, ctx, o):
    """BASELINE.json configs[3]: N=1000, M=100k, q=21, :frob, theta=:auto: n = 20 000 (157 pivot blocks: groups of four
    pivots).  A full CPU potrf+potri at this size is out of a test's budget, so the device results are checked
    through the arrays of the statement-by-statement (`_dev`) pipeline, which must give the same bits as the fused
    call:
      - theta / threshold from the closed form, 32 sampled neighbour counts by brute force (bit-exact)
      - 24 sampled 20 x 20 blocks of the device covariance against the oracle's formula for those site pairs
      - C X v == v on random probes with the device's own C and X = mJ
      - the oracle's compute_FN + correct_APC applied to the DEVICE mJ against the device scores."""
    from gaussdca.jl_amd import devops

    N, M, q, pc = 1000, 100000, 21, 0.8
    s, n = q - 1, N * (q - 1)
    Zo = _synth(N, M, 0xD1000)
    Z = np.asfortranarray(Zo.T)
    S, st = ctx.run(Z, q, pc, -1.0, 0)
    assert st["info"] == 0 and np.isfinite(S).all() and np.array_equal(S, S.T) and st["n"] == n
    tot = o.pair_identity_sum(Zo)
    theta = min(0.5, 0.38 * 0.32 / (tot / (N * (0.5 * M * (M - 1)))))
    assert st["pair_identity_sum"] == tot and st["theta"] == theta and st["thresh"] == int(np.floor(theta * N))
    thr = st["thresh"]

    n_gpu = g.neighbour_counts(Z, thr, ctx=ctx)
    rng = np.random.default_rng(0xD)
    for k in rng.choice(M, size=32, replace=False):
        d = np.count_nonzero(Zo != Zo[k], axis=1)
        assert n_gpu[k] == int(np.count_nonzero(d < thr))                   # includes k itself (d = 0): bit-exact
    W = 1.0 / n_gpu
    Meff = math.fsum(W.tolist())   # the exact sum, rounded once
    assert st["Meff"] == Meff

    # the device-resident statement chain, keeping C and mJ
    dZ = g.DeviceBuffer.from_array(ctx, Z)
    dW, Meff_d, th_d, thr_d = devops.compute_weights_dev(ctx, dZ, N, M, "auto")
    assert (Meff_d, th_d, thr_d) == (Meff, theta, thr)
    dPi, dPij = devops.compute_weighted_frequencies_dev(ctx, dZ, N, M, q, dW, Meff_d)
    Pi_true = dPi.download((n,))
    devops.add_pseudocount_dev(ctx, dPi, dPij, N, q, pc)
    Pi_pc = dPi.download((n,))
    devops.compute_C_dev(ctx, dPi, dPij, n, dC=dPij)
    C_dev = dPij.download((n, n))
    assert np.array_equal(C_dev, C_dev.T)

    # sampled covariance blocks against the oracle's formula (rules 5-7 of SURVEY 4.3) for those site pairs
    Pi_o = np.zeros(n)
    for i in rng.choice(N, size=8, replace=False):
        a = Zo[:, i].astype(np.int64)
        m = a < q
        Pi_o[i * s:(i + 1) * s] = np.bincount(a[m] - 1, weights=W[m], minlength=s) / Meff
        assert np.max(np.abs(Pi_true[i * s:(i + 1) * s] - Pi_o[i * s:(i + 1) * s])) <= 1e-12
    assert np.max(np.abs(Pi_pc - ((1.0 - pc) * Pi_true + pc / q))) == 0.0     # elementwise, contraction off: exact
    cmax = np.max(np.abs(C_dev))
    pairs = [(int(a), int(b)) for a, b in rng.integers(0, N, size=(20, 2))] + [(int(i), int(i)) for i in
                                                                                 rng.integers(0, N, size=4)]
    for i, j in pairs:
        a, b = Zo[:, i].astype(np.int64), Zo[:, j].astype(np.int64)
        m = (a < q) & (b < q)
        blk = np.bincount((a[m] - 1) * s + (b[m] - 1), weights=W[m], minlength=s * s).reshape(s, s) / Meff
        if i == j:
            blk = (1.0 - pc) * blk + (pc / q) * np.eye(s)
        else:
            blk = (1.0 - pc) * blk + pc / q / q
        blk = blk - np.outer(Pi_pc[i * s:(i + 1) * s], Pi_pc[j * s:(j + 1) * s])
        got = C_dev[i * s:(i + 1) * s, j * s:(j + 1) * s]
        assert np.max(np.abs(got - blk)) <= 1e-12 * cmax, (i, j)            # fixed-point tallies vs f64 sums

    devops.inv_cholesky_dev(ctx, dPij, n)
    mJ_dev = dPij.download((n, n))
    assert np.array_equal(mJ_dev, mJ_dev.T)
    V = rng.standard_normal((n, 6))
    Rv = C_dev @ (mJ_dev @ V) - V
    assert np.max(np.abs(Rv)) <= 1e-8 * np.max(np.abs(V))                    # cond(C) ~ 1e4
    del C_dev

    dS = devops.compute_FN_dev(ctx, dPij, N, q)
    devops.correct_APC_dev(ctx, dS, N)
    S_step = dS.download((N, N))
    assert np.array_equal(S_step, S)                                         # chain == fused call, bit for bit
    S_o = o.correct_APC(o.compute_FN(mJ_dev, q))                             # oracle scores from the device's mJ
    ok, max_rel, max_abs = score_close(S, S_o, rtol=1e-9, atol_frac=1e-12)
    assert ok, (max_rel, max_abs)
    for b in (dZ, dW, dPi, dPij, dS):
        b.free()
    R = g.compute_ranking(S, 5)
    assert len(R) == (N - 5) * (N - 4) // 2 and all(R[t][2] >= R[t + 1][2] for t in range(len(R) - 1))


def test_config_D_scores_match_oracle_at_full_size(g, ctx, o):
    """BASELINE.json configs[3] against an INDEPENDENT inverse: the oracle's whole pipeline at N=1000, M=100k (all-pairs
    Hamming, tallies, pseudocount, covariance, LAPACK dpotrf+dpotri at n = 20 000, FN, APC; about a minute on the GPU
    box's host cores) against the fused device run -- 157 pivot blocks in groups of four, the largest schedule the
    sweep kernel runs in the benchmarks.  All 100 000 neighbour counts bit-exact, theta / threshold / Meff equal,
    scores within 1e-6 relative, top-200 contact order identical."""
    N, M, q, pc = 1000, 100000, 21, 0.8
    Zo = _synth(N, M, 0xD1000)
    Z = np.asfortranarray(Zo.T)
    S, st = ctx.run(Z, q, pc, -1.0, 0)
    assert st["info"] == 0 and st["n"] == N * (q - 1)
    theta = o.compute_theta(Zo)
    thr = o.hamming_threshold(theta, N)
    assert st["theta"] == theta and st["thresh"] == thr
    n_k = o.neighbour_counts(Zo, thr)
    assert np.array_equal(g.neighbour_counts(Z, thr, ctx=ctx), n_k)          # all 100 000 counts, bit-exact
    del Z
    W, Meff = o.weights_from_counts(n_k)
    assert st["Meff"] == Meff
    Pi_t, Pij_t = o.compute_frequencies(Zo, q, W, Meff)
    Pi, Pij = o.add_pseudocount(Pi_t, Pij_t, pc, q)
    del Pij_t
    C = o.compute_C(Pi, Pij)
    del Pij
    mJ = o.spd_inverse(C)
    del C
    S_o = o.correct_APC(o.compute_FN(mJ, q))
    del mJ
    ok, max_rel, max_abs = score_close(S, S_o, rtol=1e-6, atol_frac=1e-9)    # 1e-6 relative
    assert ok, (max_rel, max_abs)
    assert _top_order_equal(g, o, S, S_o, 200)
    assert np.array_equal(S, S.T)


# ---- E -------------------------------------------------------------------------------------------------------------
def test_config_E_subset_through_the_batch_driver(g, ctx, tmp_path):
    """BASELINE.json configs[4]: 16 families of the 256-family batch (every 16th, at their real sizes N in [134, 507],
    M in [8k, 78k]) written as FASTA files and run through `gdca_cli --batch` (parser threads -> queue -> one worker
    per GPU context -> writer threads).  Every ranking file must be byte-identical to the single-family path
    (gDCA(file) + printrank through the Python mirror of the same C-ABI)."""
    from gaussdca.jl_amd import synth
    from gaussdca.jl_amd.batch import batch_sizes

    sizes = batch_sizes(256)
    fams = list(range(0, 256, 16))
    indir, outdir = tmp_path / "in", tmp_path / "out"
    indir.mkdir()
    for f in fams:
        N, M = sizes[f]
        synth.write_fasta(str(indir / ("fam%03d.fasta" % f)), synth.synth_family(N, M, 21, 0xE000 + f))
    # (every family through a launch of its own -- the driver's default -- which is what the single-family path below runs; merged
    # batches of small families are compared in test_batch_driver_merges_small_families)
    r = subprocess.run([CLI, "--batch", str(indir), "--out", str(outdir), "--parsers", "4", "--merge", "1"], capture_output=True,
                       text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "16 families" in r.stderr and "(0 failed)" in r.stderr
    want = tmp_path / "want.txt"
    for f in fams:
        N, M = sizes[f]
        R = g.gDCA(str(indir / ("fam%03d.fasta" % f)), ctx=ctx)
        st = g.gdca.last_stats
        assert (st["N"], st["M"]) == (N, M) and st["info"] == 0
        g.printrank(str(want), R)
        got = (outdir / ("fam%03d.rank.txt" % f)).read_bytes()
        assert got == want.read_bytes(), f


@pytest.mark.slow
def test_config_E_whole_batch_through_the_batch_driver(g, ctx, tmp_path):
    """BASELINE.json configs[4] WHOLE, on one GPU: all 256 families (N in [100, 600], M in [5k, 80k], 4 GB of FASTA written by the
    library's own generator) through `gdca_cli --batch` -- every family must come out (`(0 failed)`, 256 ranking files with the
    length compute_ranking gives for its N), and a sample spread over the sizes must be byte-identical to the single-family path.
    (The batch of the bench line: `bench.py --config E` runs the same 256 families device-resident.)"""
    from gaussdca.jl_amd.batch import batch_sizes

    sizes = batch_sizes(256)
    indir, outdir = tmp_path / "in", tmp_path / "out"
    indir.mkdir()
    for f, (N, M) in enumerate(sizes):
        subprocess.run([CLI, "--synth", str(N), str(M), str(0xE000 + f), str(indir / ("fam%03d.fasta" % f))], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run([CLI, "--batch", str(indir), "--out", str(outdir), "--merge", "1"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "256 families" in r.stderr and "(0 failed)" in r.stderr, r.stderr[-2000:]
    for f, (N, M) in enumerate(sizes):
        with open(outdir / ("fam%03d.rank.txt" % f), "rb") as fh:
            assert sum(1 for _ in fh) == (N - 5) * (N - 4) // 2, f       # compute_ranking with min_separation = 5
    by_size = sorted(range(256), key=lambda f: sizes[f][0] * 100000 + sizes[f][1])
    want = tmp_path / "want.txt"
    for f in [by_size[k] for k in (0, 37, 90, 128, 171, 214, 255)] + [7]:
        N, M = sizes[f]
        R = g.gDCA(str(indir / ("fam%03d.fasta" % f)), ctx=ctx)
        st = g.gdca.last_stats
        assert (st["N"], st["M"]) == (N, M) and st["info"] == 0
        g.printrank(str(want), R)
        assert (outdir / ("fam%03d.rank.txt" % f)).read_bytes() == want.read_bytes(), f


def _read_rank(path):
    ii, jj, ss = [], [], []
    with open(path) as fh:
        for line in fh:
            a, b, c = line.split()
            ii.append(int(a))
            jj.append(int(b))
            ss.append(float(c))
    return np.array(ii), np.array(jj), np.array(ss)


def test_batch_driver_merges_small_families(g, ctx, tmp_path):
    """`gdca_cli --batch --merge 8 --merge-blocks 57` sends families with a covariance of at most 57 blocks through
    gdca_run_ranked_phased_async, up to eight at a time, their SPD inverses sharing launches of the sweep kernel.  20 small families (1 to 44 blocks, ragged sizes) and two big
    ones in one directory: (a) with GDCA_MERGE_GROUP=1 (single-block groups: the schedule a launch of its own runs up to 44 blocks)
    every ranking file is byte-identical to the unmerged driver's (`--merge 1`); (b) with the default grouping the same pairs come
    out with scores equal to the file's seven digits or one unit of the last one."""
    from gaussdca.jl_amd import synth

    rng = np.random.default_rng(8)
    indir = tmp_path / "in"
    indir.mkdir()
    sizes = [(min(int(n), 281), int(m)) for n, m in zip(rng.integers(6, 288, size=20), rng.integers(300, 4000, size=20))] + [(420, 3000), (380, 2500)]
    for f, (N, M) in enumerate(sizes):
        synth.write_fasta(str(indir / ("fam%03d.fasta" % f)), synth.synth_family(N, M, 21, 0xABC0 + f))
    outs = {}
    wide = ["--merge", "8", "--merge-blocks", "57"]
    # (round 6: by DEFAULT -- no --merge on the command line, a directory of 128 files or more [here: GDCA_CLI_MERGE_MIN_FILES] -- families of up
    # to 24 blocks go eight to a batch with the pivot groups of single launches: byte-identical files; `--merge K` selects the library's own
    # grouping of merged members)
    for name, args, env in (("unmerged", ["--merge", "1"], {}), ("merged_g1", wide, {"GDCA_MERGE_GROUP": "1"}), ("merged", wide, {}),
                            ("default", [], {"GDCA_CLI_MERGE_MIN_FILES": "1"}), ("default_wide", ["--merge-blocks", "57"], {"GDCA_CLI_MERGE_MIN_FILES": "1"})):
        out = tmp_path / name
        r = subprocess.run([CLI, "--batch", str(indir), "--out", str(out), "--parsers", "4", *args], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, **env))
        assert r.returncode == 0 and "22 families" in r.stderr and "(0 failed)" in r.stderr, r.stderr[-3000:]
        outs[name] = out
    for f in range(len(sizes)):
        fn = "fam%03d.rank.txt" % f
        ref = (outs["unmerged"] / fn).read_bytes()
        assert (outs["merged_g1"] / fn).read_bytes() == ref, (f, sizes[f])
        assert (outs["default"] / fn).read_bytes() == ref and (outs["default_wide"] / fn).read_bytes() == ref, (f, sizes[f])
        i0, j0, s0 = _read_rank(outs["unmerged"] / fn)
        i1, j1, s1 = _read_rank(outs["merged"] / fn)
        assert len(i0) == len(i1) and sorted(zip(i0.tolist(), j0.tolist())) == sorted(zip(i1.tolist(), j1.tolist()))
        a = {(x, y): v for x, y, v in zip(i0.tolist(), j0.tolist(), s0.tolist())}
        dev = max(abs(a[(x, y)] - v) / max(abs(v), 1e-30) for x, y, v in zip(i1.tolist(), j1.tolist(), s1.tolist()))
        assert dev <= 2e-6, (f, sizes[f], dev)          # (%e keeps seven digits: one unit of the last is 1e-6 relative)


def test_batch_driver_survives_hardware_queues_made_beside_running_sweeps(g, ctx, tmp_path):
    """Round 6's standstill, kept as a test (tools/rounds/r06/gpu_r6q.sh, gpu_r6s.sh): with GDCA_CLI_LAZY_SETS=1 the driver makes the
    contexts of its phase batches when the first small family shows up -- their streams' hardware queues come to life beside the slots'
    running sweeps, the driver maps them by taking every queue off the device and back, and now and then (1 to 8 runs of 100) a sweep does
    not get all its workgroups back.  The sweep notices (a hole between two polls, a wave on another compute unit), gives up early, and the
    collect runs the inverse again: no run may fail, every ranking file must be the undisturbed driver's, byte for byte."""
    from gaussdca.jl_amd import synth

    rng = np.random.default_rng(8)
    indir = tmp_path / "in"
    indir.mkdir()
    sizes = [(min(int(n), 281), int(m)) for n, m in zip(rng.integers(6, 288, size=20), rng.integers(300, 4000, size=20))] + [(420, 3000), (380, 2500)]
    for f, (N, M) in enumerate(sizes):
        synth.write_fasta(str(indir / ("fam%03d.fasta" % f)), synth.synth_family(N, M, 21, 0xABC0 + f))
    wide = ["--merge", "8", "--merge-blocks", "57"]
    ref = tmp_path / "ref"
    r = subprocess.run([CLI, "--batch", str(indir), "--out", str(ref), "--parsers", "4", *wide], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, GDCA_MERGE_GROUP="1"))
    assert r.returncode == 0 and "(0 failed)" in r.stderr, r.stderr[-3000:]
    again = 0
    for rep in range(30):
        out = tmp_path / ("lazy%02d" % rep)
        r = subprocess.run([CLI, "--batch", str(indir), "--out", str(out), "--parsers", "4", *wide], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, GDCA_MERGE_GROUP="1", GDCA_CLI_LAZY_SETS="1"))
        assert r.returncode == 0 and "22 families" in r.stderr and "(0 failed)" in r.stderr, (rep, r.stderr[-3000:])
        again += r.stderr.count("run again")
        for f in range(len(sizes)):
            fn = "fam%03d.rank.txt" % f
            assert (out / fn).read_bytes() == (ref / fn).read_bytes(), (rep, f, sizes[f])
    print("inverses run again in 30 runs:", again)


@pytest.mark.parametrize("which", ["smallest", "largest", "mid_a", "mid_b"])
def test_config_E_families_match_oracle(g, ctx, o, which):
    """Four of the 16 sampled batch families at their real sizes against the oracle's whole pipeline (the batch test
    above compares the batch driver with the single-family path -- HIP against HIP; this one is the parity check at
    config E's sizes): the family with the smallest and the largest covariance of the sample and two in between."""
    from gaussdca.jl_amd.batch import batch_sizes

    sizes = batch_sizes(256)
    fams = sorted(range(0, 256, 16), key=lambda f: (sizes[f][0], sizes[f][1]))
    f = {"smallest": fams[0], "largest": fams[-1], "mid_a": fams[5], "mid_b": fams[10]}[which]
    N, M = sizes[f]
    q, pc = 21, 0.8
    Zo = _synth(N, M, 0xE000 + f)
    Z = np.asfortranarray(Zo.T)
    S, st = ctx.run(Z, q, pc, -1.0, 0)
    theta = o.compute_theta(Zo)
    thr = o.hamming_threshold(theta, N)
    n_k = o.neighbour_counts(Zo, thr)
    assert np.array_equal(g.neighbour_counts(Z, thr, ctx=ctx), n_k)          # bit-exact
    W, Meff = o.weights_from_counts(n_k)
    assert st["theta"] == theta and st["thresh"] == thr and st["Meff"] == Meff and st["info"] == 0
    S_o = o.scores_from_Z(Zo, q, pc, "auto", "frob")
    ok, max_rel, max_abs = score_close(S, S_o, rtol=1e-6, atol_frac=1e-9)    # 1e-6 relative
    assert ok, (f, N, M, max_rel, max_abs)
    assert _top_order_equal(g, o, S, S_o, 200)


# ---- boundary behaviour added with the device-resident operators ----------------------------------------------------
def test_out_of_range_symbols_and_weights_are_rejected(g, ctx):
    """A byte of Z outside 1..q (q < max(Z), 0, negative) gives ArgumentError (GDCA_EINVAL) from every entry that
    consumes Z instead of a silently inconsistent covariance; caller-given weights must lie in [0, 1]."""
    rng = np.random.default_rng(3)
    Z = np.asfortranarray(rng.integers(1, 22, size=(40, 300)).astype(np.int8))
    S, st = ctx.run(Z, 21, 0.8, 0.3, 0)                    # fine
    with pytest.raises(g.ArgumentError):
        ctx.run(Z, 20, 0.8, 0.3, 0)                        # q smaller than the largest symbol
    for bad in (0, -3, 33):
        Zb = Z.copy(order="F")
        Zb[7, 123] = bad
        with pytest.raises(g.ArgumentError):
            ctx.run(Zb, 21, 0.8, 0.3, 0)
        with pytest.raises(g.ArgumentError):
            g.compute_weights(Zb, 21, 0.3, ctx=ctx)
        with pytest.raises(g.ArgumentError):
            g.compute_weighted_frequencies(Zb, np.full(300, 0.5), 150.0, ctx=ctx)
    Zo = np.asfortranarray(rng.integers(1, 22, size=(37, 300)).astype(np.int8))   # N not a multiple of 4: slow pack path
    Zo[36, 299] = 0
    with pytest.raises(g.ArgumentError):
        g.neighbour_counts(Zo, 5, ctx=ctx)
    for w in (1.5, -0.1, np.nan):
        W = np.full(300, 0.5)
        W[17] = w
        with pytest.raises(g.ArgumentError):
            g.compute_weighted_frequencies(Z, W, 150.0, ctx=ctx)
    S2, _ = ctx.run(Z, 21, 0.8, 0.3, 0)                    # the context is still usable and deterministic
    assert np.array_equal(S, S2)


def test_second_async_run_before_collect_is_an_error(g, ctx):
    import torch

    Zo = _synth(40, 300, 5)
    Zd = torch.from_numpy(Zo).cuda()
    S = torch.empty((40, 40), dtype=torch.float64, device="cuda")
    c = g.Context(0)
    c.run_dev_async(Zd.data_ptr(), 40, 300, 21, 0.8, -1.0, 0, S.data_ptr())
    with pytest.raises(g.ArgumentError):
        c.run_dev_async(Zd.data_ptr(), 40, 300, 21, 0.8, -1.0, 0, S.data_ptr())   # one run outstanding per ctx
    st = c.collect()
    assert st["info"] == 0
    c.run_dev_async(Zd.data_ptr(), 40, 300, 21, 0.0, -1.0, 0, S.data_ptr())       # pc = 0: singular covariance
    with pytest.raises(g.PosDefException):
        c.collect()                                                                 # the status reaches the caller
    c.close()


@pytest.mark.parametrize("score", ["frob", "DI"])
def test_device_resident_chain_equals_fused_run(g, ctx, o, score):
    """The reference's six statements (src/GaussDCA.jl:28-42) through the `_dev` operators, arrays in gdca_dbuf
    buffers, give the same bits as gdca_run and match the oracle."""
    from gaussdca.jl_amd import devops

    pc = 0.2 if score == "DI" else 0.8
    for (N, M, q) in ((75, 900, 21), (33, 400, 5), (130, 2000, 21)):
        Zo = _synth(N, M, 77 + N) if q == 21 else np.ascontiguousarray(
            np.random.default_rng(N).integers(1, q + 1, size=(M, N)).astype(np.int8))
        Z = np.asfortranarray(Zo.T)
        S, st = ctx.run(Z, q, pc, -1.0, 1 if score == "DI" else 0)
        S2, info = devops.scores_stepwise(Z, q, pc, "auto", score, ctx=ctx)
        assert np.array_equal(S, S2), (N, M, q)
        assert info["Meff"] == st["Meff"] and info["thresh"] == st["thresh"] and info["theta"] == st["theta"]
        S_o = o.scores_from_Z(Zo, q, pc, "auto", score)
        ok, max_rel, _ = score_close(S2, S_o, rtol=1e-6, atol_frac=1e-9)
        assert ok, max_rel
