"""GPU parity tests (run on the MI355X box with `-m gpu`): the HIP path, called through the C-ABI
(libgdca.so via ctypes), against the CPU oracle and the reference's golden vectors.

Bars (BASELINE.json north_star): Hamming counts, thresholds, pair-identity sums and ranking
indices bit-exact; FN / DI scores within 1e-6 relative (tolerances are written at each assert).
"""
import json
import os

import math
import numpy as np
import pytest

from gdca_testutil import CASES, compare_with_golden, random_msa, score_close

pytestmark = pytest.mark.gpu


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

    return o


# ---- the reference's own test-suite, through the C-ABI (test/runtests.jl:52-86) -------------------
@pytest.mark.parametrize("golden", list(CASES))
def test_reference_goldens_through_cabi(g, ctx, golden, refdata):
    c = CASES[golden]
    R = g.gDCA(os.path.join(refdata, c["fasta"]), ctx=ctx, **c["kw"])
    rep = compare_with_golden(R, os.path.join(refdata, golden))
    assert rep["keys_equal"]                      # pair enumeration: bit-exact
    assert rep["max_rel"] <= 1e-6, rep            # scores: 1e-6 relative (7-digit prints)
    # 7th-digit boundary flips: a score within ~1e-10 relative of a print-rounding boundary may print
    # differently (expected count ~ rows * 2 * 1e-10 / 1e-6); they are not parity failures (the hard
    # bar is max_rel above) but must stay isolated: <= 0.05 % of the rows
    assert rep["string_mismatches"] <= max(3, len(R) // 2000), rep
    assert rep["order_equal_mod_ties"], rep       # ranking order identical (tie rule of SURVEY 4.3)
    if not golden.startswith("large"):
        assert rep["order_equal"], rep
    st = g.gdca.last_stats
    with open(os.path.join(os.path.dirname(refdata), "intermediates.json")) as f:
        want = json.load(f)[golden]
    assert st["thresh"] == want["thresh"] and st["N"] == want["N"] and st["M"] == want["M"]
    assert st["theta"] == want["theta"]           # same f64 expression as the oracle: bit-exact
    assert st["Meff"] == want["Meff"]             # exactly rounded sum of the 1/n_k: bit-exact
    if c["kw"].get("theta", "auto") == "auto":
        assert st["pair_identity_sum"] == want["pair_identity_sum"]


def test_second_hamming_implementation_against_the_same_goldens(g, ctx, o, refdata):
    """test3 of the reference (test/runtests.jl:78-86) runs a golden again with ENV["DCAUTILS_FORCE_FALLBACK"], i.e. through
    DCAUtils' second, independent Hamming implementation.  The analogue here: GDCA_FORCE_FALLBACK switches the reweighting to a
    plain byte-compare kernel that shares nothing with the bit-sliced one (an option of the context: gdca_ctx_set_option, or the
    environment variable when the context is created).  Both must reproduce the golden (the reference's case and the large one)
    and give identical neighbour counts, bit for bit, on random alignments with awkward sizes."""
    ctx.set_option("GDCA_FORCE_FALLBACK", "true")
    for golden in ("small.DIRout.txt", "large.DIRout.txt"):
        c = CASES[golden]
        R = g.gDCA(os.path.join(refdata, c["fasta"]), ctx=ctx, **c["kw"])
        rep = compare_with_golden(R, os.path.join(refdata, golden))
        # (the large golden holds scores that are equal at the 7 digits it prints: order modulo those ties, as in the main test)
        assert rep["keys_equal"] and rep["order_equal_mod_ties"] and rep["max_rel"] <= 1e-6, (golden, rep)
    rng = np.random.default_rng(17)
    for (M, N, q, thr) in ((301, 37, 21, 12), (1000, 130, 21, 45), (129, 33, 5, 20), (64, 7, 3, 4)):
        Zo = random_msa(rng, M, N, q)
        Z = np.asfortranarray(Zo.T)
        ctx.set_option("GDCA_FORCE_FALLBACK", "true")
        n_fb = g.neighbour_counts(Z, thr, ctx=ctx)
        ctx.set_option("GDCA_FORCE_FALLBACK", "0")
        n_bs = g.neighbour_counts(Z, thr, ctx=ctx)
        assert np.array_equal(n_fb, n_bs) and np.array_equal(n_fb, o.neighbour_counts(Zo, thr)), (M, N, q)


def test_hamming_lower_bound_form_counts_exactly(g, ctx, o):
    """The reweighting kernel has three forms: exact distances on all five bit planes (csrc/k_hamming.hip), a three-plane
    lower bound followed by exact refinement of the few pairs the bound does not rule out, and (round 6, csrc/k_hamming_fp4.hip)
    a bit-count lower bound computed as a Gram matrix on the fp4 matrix pipe, refined the same way; a sample of tiles decides per family
    (dense families keep the exact form).  All four settings -- forced exact, forced bound, forced mfma, automatic -- must give the same
    neighbour counts, bit for bit, as the oracle: on unrelated random sequences, on a clustered synthetic family (the
    benchmark's generator), on a DENSE family in which every pair is a neighbour (every pair a candidate: the worst case of the
    bound), on awkward sizes and at thresholds around the bound's blind spot."""
    from gaussdca.jl_amd import synth

    rng = np.random.default_rng(23)
    root = rng.integers(1, 21, size=300).astype(np.int8)
    dense = np.tile(root, (700, 1))
    flip = rng.random(dense.shape) < 0.05
    dense[flip] = rng.integers(1, 22, size=int(flip.sum())).astype(np.int8)
    cases = [("random", random_msa(rng, 1500, 90, 21), 40), ("random_q5", random_msa(rng, 400, 33, 5), 25),
             ("clustered", synth.synth_family(200, 9000, 21, 0xC0DE), 70), ("clustered_low_thr", synth.synth_family(64, 3000, 21, 7), 3),
             ("dense", dense, 100), ("tiny", random_msa(rng, 70, 9, 21), 5)]
    for name, Zo, thr in cases:
        Z = np.asfortranarray(Zo.T)
        want = o.neighbour_counts(Zo, thr)
        for mode in ("full", "bound", "mfma", "auto"):
            ctx.set_option("HAMMING_MODE", mode)
            assert np.array_equal(g.neighbour_counts(Z, thr, ctx=ctx), want), (name, mode)
    # sizes around the fp4 form's 256 x 256 tiles and 8-entry chunks (N = 32 k +- 1, M = 256 k +- 1), thresholds from 1 to N / 2
    for M, N, thr in ((255, 31, 1), (256, 32, 10), (257, 33, 16), (511, 96, 48), (513, 161, 40), (1025, 257, 100), (700, 353, 176)):
        Zo = synth.synth_family(N, M, 21, 0xF4 + M)
        want = o.neighbour_counts(Zo, thr)
        for mode in ("mfma", "bound", "auto"):
            ctx.set_option("HAMMING_MODE", mode)
            assert np.array_equal(g.neighbour_counts(np.asfortranarray(Zo.T), thr, ctx=ctx), want), (M, N, thr, mode)
    ctx.set_option("HAMMING_MODE", "auto")


def test_gdca_matches_oracle_ranking_order(g, ctx, o, refdata):
    """Ranking indices identical to the oracle's on test/data (no ties on `small`)."""
    f = os.path.join(refdata, "small.fasta.gz")
    for kw in ({}, dict(pseudocount=0.2, score="DI", remove_dups=True), dict(theta=0.3, min_separation=1)):
        R = g.gDCA(f, ctx=ctx, **kw)
        Ro = o.gDCA(f, **kw)
        assert [(i, j) for i, j, _ in R] == [(i, j) for i, j, _ in Ro]


# ---- operator level, seeded random alignments incl. ragged sizes ------------------------------------
SHAPES = [  # (M, N, q, theta, pc)
    (2, 1, 21, "auto", 0.8),
    (3, 7, 21, 0.5, 0.8),
    (65, 31, 21, "auto", 0.5),
    (129, 33, 21, "auto", 0.2),
    (300, 64, 21, 0.25, 0.8),
    (257, 40, 5, "auto", 0.3),
    (200, 20, 31, 0.3, 0.6),
    (150, 36, 22, "auto", 0.7),   # s = 21: first size on the 16-column tally variant
    (1100, 45, 20, "auto", 0.4),  # no gap state present: q = 20, s = 19; more than one 1024-sequence pass
    (1000, 97, 21, "auto", 0.8),
]


@pytest.mark.parametrize("M,N,q,theta,pc", SHAPES)
def test_operator_parity(g, ctx, o, M, N, q, theta, pc):
    rng = np.random.default_rng(1000 * M + N)
    Zo = random_msa(rng, M, N, q)
    Zo[0, 0] = q  # make sure maximum(Z) == q
    Z = np.asfortranarray(Zo.T)

    # integers: bit-exact
    assert g.pair_identity_sum(Z, ctx=ctx) == o.pair_identity_sum(Zo) == o.pair_identity_sum_allpairs(Zo)
    W_o, Meff_o, th_o, thr_o = o.compute_weights(Zo, theta)
    for thr in sorted({0, 1, thr_o, N // 2, N + 1}):
        assert np.array_equal(g.neighbour_counts(Z, thr, ctx=ctx), o.neighbour_counts(Zo, thr)), thr
    # theta, W: same f64 expressions; Meff: the exact sum rounded once on both sides -> bit-exact
    W, Meff, th, thr = g.compute_weights(Z, q, theta, ctx=ctx, return_theta=True)
    assert th == th_o and thr == thr_o
    assert np.array_equal(W, W_o) and Meff == Meff_o

    # frequencies: 64-bit fixed-point tallies vs sequential f64 sums: 1e-12 relative to max
    Pi_o, Pij_o = o.compute_frequencies(Zo, q, W_o, Meff_o)
    Pi, Pij = g.compute_weighted_frequencies(Z, W_o, Meff_o, ctx=ctx)
    assert np.max(np.abs(Pi - Pi_o)) <= 1e-12 * max(1e-300, np.max(np.abs(Pi_o)))
    assert np.max(np.abs(Pij - Pij_o)) <= 1e-12 * max(1e-300, np.max(np.abs(Pij_o)))
    assert np.array_equal(Pij, Pij.T)
    Pi4, Pij4, Meff4, W4 = g.compute_weighted_frequencies(Z, q, theta, ctx=ctx)
    assert np.array_equal(Pi4, Pi) and np.array_equal(Pij4, Pij) and Meff4 == Meff_o and np.array_equal(W4, W_o)

    # add_pseudocount / compute_C: elementwise f64 with contraction off -> bit-exact
    Pi2_o, Pij2_o = o.add_pseudocount(Pi_o, Pij_o, pc, q)
    Pi2, Pij2 = g.add_pseudocount(Pi_o, Pij_o, pc, q, ctx=ctx)
    assert np.array_equal(Pi2, Pi2_o) and np.array_equal(Pij2, Pij2_o)
    C_o = o.compute_C(Pi2_o, Pij2_o)
    assert np.array_equal(g.compute_C(Pi2_o, Pij2_o, ctx=ctx), C_o)

    # SPD inverse: residual and agreement with LAPACK potrf+potri, scaled by the condition number
    mJ_o = o.spd_inverse(C_o)
    mJ = g.inv_cholesky(C_o, ctx=ctx)
    n = C_o.shape[0]
    cond = np.linalg.cond(C_o)
    assert np.array_equal(mJ, mJ.T)
    assert np.linalg.norm(C_o @ mJ - np.eye(n)) <= 1e-13 * cond * np.sqrt(n)
    assert np.max(np.abs(mJ - mJ_o)) <= 1e-13 * cond * np.max(np.abs(mJ_o))

    # scores from identical inputs: 1e-9 relative to max (north_star bar is 1e-6)
    FN_o = o.compute_FN(mJ_o, q)
    FN = g.compute_FN(mJ_o, q, ctx=ctx)
    assert np.max(np.abs(FN - FN_o)) <= 1e-12 * max(1e-300, np.max(np.abs(FN_o)))
    if N >= 2:
        DI_o = o.compute_DI_gauss(mJ_o, C_o, q)
        DI = g.compute_DI_gauss(mJ_o, C_o, q, ctx=ctx)
        assert np.max(np.abs(DI - DI_o)) <= 1e-9 * max(1e-300, np.max(np.abs(DI_o)))
        assert np.max(np.abs(g.correct_APC(FN_o, ctx=ctx) - o.correct_APC(FN_o))) <= 1e-12 * np.max(np.abs(FN_o))

    # fused device path == oracle end to end: 1e-6 relative per entry (+ tiny absolute term for
    # APC-corrected scores that cross zero), for both scores
    if N >= 2:
        for score, sc in (("frob", 0), ("DI", 1)):
            S, st = ctx.run(Z, q, pc, -1.0 if theta == "auto" else float(theta), sc)
            S_o = o.scores_from_Z(Zo, q, pc, theta, score)
            ok, max_rel, _ = score_close(S, S_o, rtol=1e-6, atol_frac=1e-9)
            assert ok and max_rel <= 1e-6, (score, max_rel)
            assert st["Meff"] == Meff_o and st["thresh"] == thr_o and st["info"] == 0
            assert np.array_equal(S, S.T)


@pytest.mark.parametrize("name", ["synthetic_a", "synthetic_b", "synthetic_c"])
def test_committed_synthetic_vectors(g, ctx, name, refdata):
    """GPU path against the stored oracle vectors of tests/golden/synthetic_*.npz."""
    d = np.load(os.path.join(os.path.dirname(refdata), name + ".npz"))
    Zo, q, pc = d["Z"], int(d["q"]), float(d["pseudocount"])
    theta = -1.0 if str(d["theta_in"]) == "auto" else float(str(d["theta_in"]))
    Z = np.asfortranarray(Zo.T)
    assert np.array_equal(g.neighbour_counts(Z, int(d["thresh"]), ctx=ctx), d["n_k"])   # bit-exact
    if theta < 0:
        assert g.pair_identity_sum(Z, ctx=ctx) == int(d["pair_identity_sum"])
    for sc, key in ((0, "S_frob"), (1, "S_DI")):
        S, st = ctx.run(Z, q, pc, theta, sc)
        assert st["theta"] == float(d["theta"]) and st["thresh"] == int(d["thresh"]) and st["Meff"] == float(d["Meff"])
        ok, max_rel, _ = score_close(S, d[key], rtol=1e-6, atol_frac=1e-9)                # 1e-6 relative
        assert ok, (key, max_rel)


def test_theta_zero_and_tiny_theta(g, ctx):
    rng = np.random.default_rng(5)
    Z = np.asfortranarray(random_msa(rng, 90, 53).T)
    W, Meff = g.compute_weights(Z, 21, 0.0, ctx=ctx)
    assert Meff == 90.0 and np.all(W == 1.0)                      # theta == 0: W = 1, Meff = M
    W, Meff, th, thr = g.compute_weights(Z, 21, 0.01, ctx=ctx, return_theta=True)
    assert thr == 0 and Meff == 90.0                              # floor(theta N) == 0: n_k = 1
    Zd = np.asfortranarray(np.concatenate([Z, Z[:, :7]], axis=1))  # duplicates: d = 0 < 1
    n = g.neighbour_counts(Zd, 1, ctx=ctx)
    assert np.all(n[90:] >= 2) and np.all(n[:7] >= 2)


def test_not_positive_definite_is_reported_like_cholesky(g, ctx, o):
    rng = np.random.default_rng(2)
    A = rng.standard_normal((200, 300))
    C = A @ A.T / 300 + 0.1 * np.eye(200)
    C[150, 150] = -1.0  # leading minor 151 fails
    with pytest.raises(o.NotPositiveDefinite) as eo:
        o.spd_inverse(C)
    with pytest.raises(g.PosDefException) as eg:
        g.inv_cholesky(C, ctx=ctx)
    assert eg.value.info == eo.value.info == 151
    # pseudocount 0 with few sequences: covariance singular -> gDCA raises like the reference would
    Z = np.asfortranarray(random_msa(rng, 30, 20).T)
    with pytest.raises(g.PosDefException):
        ctx.run(Z, 21, 0.0, 0.3, 0)
    with pytest.raises(g.ArgumentError):
        ctx.run(Z, 40, 0.5, 0.3, 0)  # q >= 32 (src/GaussDCA.jl:26)
    with pytest.raises(g.ArgumentError):
        ctx.run(Z, 21, 1.5, 0.3, 0)


@pytest.mark.parametrize("N,sep", [(6, 1), (53, 5), (53, 52), (53, 60), (400, 5), (1000, 5), (1300, 1)])
def test_device_ranking_equals_host_ranking(g, ctx, N, sep):
    """compute_ranking (src/GaussDCA.jl:88-99) sorted on the device (gdca_ranking_dev: what gdca_run_ranked and gDCA() return)
    against the host form, entry for entry: indices bit-exact, ties in generation order, NaN first, 0.0 before -0.0, infinities."""
    from gaussdca.jl_amd import dcautils

    rng = np.random.default_rng(N * 131 + sep)
    variants = {
        "continuous": rng.standard_normal((N, N)),
        "heavy ties": np.round(rng.standard_normal((N, N)) * 3.0) / 3.0,           # a few dozen distinct values, many zeros
        "all equal": np.full((N, N), 0.25),
        "specials": rng.standard_normal((N, N)),
    }
    sp = variants["specials"]
    pick = rng.integers(0, 7, size=(N, N))
    sp[pick == 0] = 0.0
    sp[pick == 1] = -0.0
    sp[pick == 2] = np.nan
    sp[(pick == 3) & (rng.random((N, N)) < 0.1)] = np.inf
    sp[(pick == 4) & (rng.random((N, N)) < 0.1)] = -np.inf
    sp[(pick == 5) & (rng.random((N, N)) < 0.1)] = 5e-324                            # a subnormal
    for name, S in variants.items():
        S = np.asfortranarray(S)
        want = dcautils.compute_ranking(S, sep)
        buf = g.DeviceBuffer.from_array(ctx, S)
        try:
            ii, jj, sc = ctx.ranking_dev(buf.ptr, N, sep)
        finally:
            buf.free()
        assert len(ii) == len(want) == max(N - sep, 0) * (max(N - sep, 0) + 1) // 2
        assert np.array_equal(ii, want.i) and np.array_equal(jj, want.j), (name, N, sep)
        assert np.array_equal(sc.view(np.uint64), want.score.view(np.uint64)), (name, N, sep)   # the same bits (NaN payloads, -0.0)


def test_run_ranked_equals_run_plus_host_ranking(g, ctx):
    """gdca_run_ranked = gdca_run + compute_ranking, also when the collect computes the scores again (a refined inverse)."""
    from gaussdca.jl_amd import dcautils

    rng = np.random.default_rng(77)
    Zo = random_msa(rng, 500, 60)
    Zf = np.asfortranarray(Zo.T)
    for pc, score in ((0.8, 0), (0.2, 1), (1e-5, 0)):
        S, st = ctx.run(Zf, 21, pc, -1.0, score)
        want = dcautils.compute_ranking(S, 5)
        ii, jj, sc, st2 = ctx.run_ranked_ptr(Zf.ctypes.data, 60, 500, 21, pc, -1.0, score, 5)
        print("pc=%g: refined %d" % (pc, st["refined"]))
        assert st2["refined"] == st["refined"]
        assert np.array_equal(ii, want.i) and np.array_equal(jj, want.j) and np.array_equal(sc, want.score)
    with pytest.raises(g.ArgumentError):
        ctx.run_ranked_ptr(Zf.ctypes.data, 60, 500, 21, 0.8, -1.0, 0, 0)          # min_separation < 1 (src/GaussDCA.jl:59-60)


def test_ranked_runs_pipelined_over_peer_contexts(g, ctx):
    """gdca_run_ranked_async / _collect over a leader and its peer, driven the way `gdca_cli --batch` drives them (family k+1 is
    uploaded and enqueued before family k is collected): the same rankings as synchronous runs; a context with an enqueued run
    refuses everything else, a second collect is an error."""
    rng = np.random.default_rng(31)
    fams = [np.asfortranarray(random_msa(rng, M, N).T) for M, N in ((400, 40), (900, 75), (300, 130), (700, 55), (500, 90))]
    want = [ctx.run_ranked_ptr(Z.ctypes.data, Z.shape[0], Z.shape[1], 21, 0.8, -1.0, 0, 5) for Z in fams]
    lead = g.Context(0)
    peer = lead.peer()
    try:
        slots = [lead, peer]
        got = {}
        busy = {}
        for k, Z in enumerate(fams):
            c = slots[k % 2]
            if c in busy:
                got[busy.pop(c)] = c.run_ranked_collect()
            c.run_ranked_async_ptr(Z.ctypes.data, Z.shape[0], Z.shape[1], 21, 0.8, -1.0, 0, 5)
            busy[c] = k
            if k == 1:
                with pytest.raises(g.ArgumentError):      # enqueued: nothing else on this context
                    c.run_ranked_async_ptr(Z.ctypes.data, Z.shape[0], Z.shape[1], 21, 0.8, -1.0, 0, 5)
                with pytest.raises(g.ArgumentError):
                    c.run(Z, 21, 0.8, -1.0, 0)
        for c, k in list(busy.items()):
            got[k] = c.run_ranked_collect()
        with pytest.raises(g.ArgumentError):
            lead.run_ranked_collect()
        for k in range(len(fams)):
            for a, b in zip(got[k][:3], want[k][:3]):
                assert np.array_equal(a, b), k
            assert got[k][3]["Meff"] == want[k][3]["Meff"]
    finally:
        peer.close()
        lead.close()


def test_device_pointer_entry_and_determinism(g, ctx, o):
    import torch

    rng = np.random.default_rng(9)
    Zo = random_msa(rng, 700, 75)
    Zd = torch.from_numpy(Zo).cuda()
    S1 = torch.empty((75, 75), dtype=torch.float64, device="cuda")
    S2 = torch.empty_like(S1)
    st1 = ctx.run_dev(Zd.data_ptr(), 75, 700, 21, 0.8, -1.0, 0, S1.data_ptr())
    st2 = ctx.run_dev(Zd.data_ptr(), 75, 700, 21, 0.8, -1.0, 0, S2.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(S1, S2)  # integer tallies + fixed schedules: bit-reproducible run to run
    assert st1["Meff"] == st2["Meff"]
    S_o = o.scores_from_Z(Zo, 21, 0.8, "auto", "frob")
    ok, max_rel, _ = score_close(S1.cpu().numpy(), S_o)
    assert ok, max_rel


def test_async_pipeline_api(g, ctx, o):
    """gdca_run_dev_async / gdca_run_collect over two contexts (plain and gated peers): same bits as
    the synchronous entry point."""
    import torch

    rng = np.random.default_rng(21)
    fams = [random_msa(rng, 400 + 50 * t, 40 + 7 * t) for t in range(4)]
    Zd = [torch.from_numpy(z).cuda() for z in fams]
    ref = []
    for z, zd in zip(fams, Zd):
        S = torch.empty((z.shape[1], z.shape[1]), dtype=torch.float64, device="cuda")
        ctx.run_dev(zd.data_ptr(), z.shape[1], z.shape[0], 21, 0.8, -1.0, 0, S.data_ptr())
        ref.append(S.cpu())
    for peers in (False, True):
        c0 = g.Context(0)
        cs = [c0, c0.peer() if peers else g.Context(0)]
        outs = [torch.empty((z.shape[1], z.shape[1]), dtype=torch.float64, device="cuda") for z in fams]
        busy = [False, False]
        stats = []
        for t, (z, zd) in enumerate(zip(fams, Zd)):
            c = t % 2
            if busy[c]:
                stats.append(cs[c].collect())
            cs[c].run_dev_async(zd.data_ptr(), z.shape[1], z.shape[0], 21, 0.8, -1.0, 0, outs[t].data_ptr())
            busy[c] = True
        for c in range(2):
            stats.append(cs[c].collect())
        torch.cuda.synchronize()
        assert len(stats) == 4 and all(st["info"] == 0 for st in stats)
        for t in range(4):
            assert torch.equal(outs[t].cpu(), ref[t])
        with pytest.raises(g.ArgumentError):
            cs[0].collect()  # nothing enqueued
        for c in cs:
            c.close()


def test_phase_batched_runs_equal_single_runs(g, ctx, o):
    """gdca_run_dev_phased: K families of mixed sizes batched by phase on one stream (K front ends, K inverses back to back,
    K score stages) -- both scores, single-block and multi-block schedules (N = 40 .. 430), collected in reverse order --
    must give the bits of K single runs; a second batch on the same contexts reuses the workspaces; misuse is GDCA_EINVAL
    and leaves the contexts usable."""
    import torch

    from gaussdca.jl_amd import synth

    sizes = [(40, 500), (130, 2000), (75, 900), (430, 3000), (64, 64)]
    fams = [synth.synth_family(N, M, 21, 0xF00 + N) for N, M in sizes]
    Zd = [torch.from_numpy(z).cuda() for z in fams]
    for score, pc in ((0, 0.8), (1, 0.2)):
        ref, ref_st = [], []
        for z, zd in zip(fams, Zd):
            S = torch.empty((z.shape[1], z.shape[1]), dtype=torch.float64, device="cuda")
            ref_st.append(ctx.run_dev(zd.data_ptr(), z.shape[1], z.shape[0], 21, pc, -1.0, score, S.data_ptr()))
            ref.append(S.cpu())
        cs = [g.Context(0) for _ in fams]
        first = []
        for rep in range(2):
            outs = [torch.zeros((z.shape[1], z.shape[1]), dtype=torch.float64, device="cuda") for z in fams]
            g.run_dev_phased(cs, [zd.data_ptr() for zd in Zd], [z.shape[1] for z in fams], [z.shape[0] for z in fams],
                             [21] * len(fams), pc, -1.0, score, [x.data_ptr() for x in outs])
            with pytest.raises(g.ArgumentError):                       # a member still has its run outstanding
                cs[1].run_dev_async(Zd[1].data_ptr(), sizes[1][0], sizes[1][1], 21, pc, -1.0, score, outs[1].data_ptr())
            sts = [None] * len(cs)
            for k in reversed(range(len(cs))):
                sts[k] = cs[k].collect()
            for k in range(len(cs)):
                # (a member of a merged launch sweeps in larger pivot groups than a launch of its own would at that size: the same
                # inverse to rounding -- the scores agree to ~1e-13 --, and bit for bit from one batch to the next)
                if sizes[k][0] * 20 > 57 * 128:
                    assert torch.equal(outs[k].cpu(), ref[k]), (score, rep, k)
                else:
                    assert torch.allclose(outs[k].cpu(), ref[k], rtol=1e-10, atol=1e-12 * float(ref[k].abs().max())), (score, rep, k)
                if rep == 0:
                    first = first + [outs[k].cpu()] if k else [outs[k].cpu()]
                else:
                    assert torch.equal(outs[k].cpu(), first[k]), (score, rep, k)
                assert sts[k]["Meff"] == ref_st[k]["Meff"] and sts[k]["thresh"] == ref_st[k]["thresh"] and sts[k]["info"] == 0
                assert sts[k]["ms_inverse"] > 0 and sts[k]["sweep_ghz"] > 1.0
            # the four small members shared ONE merged sweep launch (its first member accounts for it), N = 430 had its own
            assert [st_["inverse_batch"] for st_ in sts] == [4, 4, 4, 1, 4] and sum(st_["update_launches"] for st_ in sts) == 2
        # MERGE_GROUP=1 (single-block groups in merged launches too): bit for bit the launches of their own, for members up to 44 blocks
        # (... and the front ends on two streams instead of four, then on one per member: the same bits)
        cs[0].set_options(MERGE_GROUP=1, PHASED_STREAMS=2 if score == 0 else 64)
        outs = [torch.zeros((z.shape[1], z.shape[1]), dtype=torch.float64, device="cuda") for z in fams]
        g.run_dev_phased(cs, [zd.data_ptr() for zd in Zd], [z.shape[1] for z in fams], [z.shape[0] for z in fams],
                         [21] * len(fams), pc, -1.0, score, [x.data_ptr() for x in outs])
        for c in cs:
            c.collect()
        for k in range(len(cs)):
            assert torch.equal(outs[k].cpu(), ref[k]), (score, "group 1", k)
        # Round 6: the members' kernels of a kind as ONE grid (PHASED_GRIDS: 1, 2, 4 groups side by side, -1 the rule; 0 = a launch
        # per member and kernel) -- the same bits in every form, from independent contexts and from the peers of one pipeline
        # (whose batches all go to the pipeline's stream); stage times come from device time stamps and are sane
        peers = [cs[0]] + [cs[0].peer() for _ in fams[1:]]
        for grids, cset in ((0, cs), (1, cs), (2, cs), (4, peers), (-1, peers), (1, peers)):
            cset[0].set_options(PHASED_GRIDS=grids)
            outs = [torch.zeros((z.shape[1], z.shape[1]), dtype=torch.float64, device="cuda") for z in fams]
            g.run_dev_phased(cset, [zd.data_ptr() for zd in Zd], [z.shape[1] for z in fams], [z.shape[0] for z in fams],
                             [21] * len(fams), pc, -1.0, score, [x.data_ptr() for x in outs])
            sts = [c.collect() for c in cset]
            for k in range(len(cset)):
                assert torch.equal(outs[k].cpu(), ref[k]), (score, "grids", grids, k)
                assert sts[k]["Meff"] == ref_st[k]["Meff"] and sts[k]["info"] == 0
                for key in ("ms_theta", "ms_weights", "ms_covariance", "ms_inverse", "ms_inverse_update", "ms_score"):
                    assert 0.0 < sts[k][key] < 1e3, (grids, k, key, sts[k][key])
                assert sts[k]["ms_total"] >= sts[k]["ms_inverse"]
        cs[0].set_options(MERGE_GROUP=-1, PHASED_STREAMS=4, PHASED_GRIDS=-1)
        with pytest.raises(g.ArgumentError):
            g.run_dev_phased([cs[0], cs[0]], [Zd[0].data_ptr()] * 2, [40] * 2, [500] * 2, [21] * 2, pc, -1.0, score,
                             [outs[0].data_ptr()] * 2)               # the same context twice
        S1 = torch.empty((40, 40), dtype=torch.float64, device="cuda")
        cs[0].run_dev(Zd[0].data_ptr(), 40, 500, 21, pc, -1.0, score, S1.data_ptr())   # still usable on its own stream
        assert torch.equal(S1.cpu(), ref[0])
        for c in cs:
            c.close()


# ---- BASELINE.json's full sizes: size-independent properties ----------------------------------------
def test_headline_config_properties(g, ctx):
    """N=500, M=50k, q=21 (configs[2]): sampled bit-exact neighbour counts, invariance of the scores
    under a permutation of the sequences, run-to-run determinism, ranking sortedness."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import synth_family

    N, M, q = 500, 50000, 21
    Zo = synth_family(N, M, q, 0xC500)
    Z = np.asfortranarray(Zo.T)
    S, st = ctx.run(Z, q, 0.8, -1.0, 0)
    assert st["info"] == 0 and np.isfinite(S).all() and np.array_equal(S, S.T)
    # closed-form theta from column counts (exact integer arithmetic on the host)
    tot = 0
    for a in range(1, q + 1):
        c = np.count_nonzero(Zo == a, axis=0).astype(np.int64)
        tot += int(np.sum(c * (c - 1) // 2))
    assert st["pair_identity_sum"] == tot
    theta = min(0.5, 0.38 * 0.32 / (tot / (N * (0.5 * M * (M - 1)))))
    assert st["theta"] == theta and st["thresh"] == int(np.floor(theta * N))
    # neighbour counts of 48 sampled sequences by brute force: bit-exact
    n_gpu = g.neighbour_counts(Z, st["thresh"], ctx=ctx)
    rng = np.random.default_rng(1)
    for k in rng.choice(M, size=48, replace=False):
        d = np.count_nonzero(Zo != Zo[k], axis=1)
        assert n_gpu[k] == int(np.count_nonzero(d < st["thresh"]))  # includes k itself (d = 0)
    W = 1.0 / n_gpu
    assert st["Meff"] == math.fsum(W.tolist())   # the exact sum, rounded once
    # permutation of the sequences: same integer sums, Meff differs only by summation order
    perm = rng.permutation(M)
    S2, st2 = ctx.run(np.asfortranarray(Zo[perm].T), q, 0.8, -1.0, 0)
    assert st2["thresh"] == st["thresh"] and abs(st2["Meff"] - st["Meff"]) <= 1e-12 * st["Meff"]
    ok, max_rel, _ = score_close(S2, S, rtol=1e-6, atol_frac=1e-9)
    assert ok, max_rel
    # determinism
    S3, _ = ctx.run(Z, q, 0.8, -1.0, 0)
    assert np.array_equal(S3, S)
    R = g.compute_ranking(S, 5)
    assert len(R) == (N - 5) * (N - 4) // 2 and all(R[t][2] >= R[t + 1][2] for t in range(len(R) - 1))


@pytest.mark.parametrize("n", [6000, 9100])
def test_large_spd_inverse_residual(g, ctx, n):
    """n = 6000 (47 pivot blocks: single pivots) and n = 9100 (72 blocks: groups of three, K = 384 trailing updates):
    A X v == v on random probes."""
    rng = np.random.default_rng(4)
    B = rng.standard_normal((n, 64))
    d = 0.5 + rng.random(n)
    A = (B @ B.T) / 64 + np.diag(d)
    X = g.inv_cholesky(A, ctx=ctx)
    V = rng.standard_normal((n, 8))
    R = A @ (X @ V) - V
    assert np.max(np.abs(R)) <= 1e-9 * np.max(np.abs(V))
    assert np.array_equal(X, X.T)


_SCHEDULES = [{"GDCA_GROUP": "1"}, {"GDCA_GROUP": "2"}, {"GDCA_GROUP": "3"}, {"GDCA_GROUP": "4"},
              {"GDCA_GROUP": "3", "GDCA_MCUS": "1"}, {"GDCA_GROUP": "2", "GDCA_MCUS": "16"},
              {"GDCA_GROUP": "4", "GDCA_MCUS": "3"}, {"GDCA_GROUP": "3", "GDCA_REM_TAIL": "0"},
              {"GDCA_GROUP": "2", "GDCA_REM_TAIL": "7"}, {"GDCA_GROUP": "4", "GDCA_REM_TAIL": "100000"},
              {"GDCA_GROUP": "3", "GDCA_PANEL_HALVES": "1"}, {"GDCA_GROUP": "2", "GDCA_PANEL_HALVES": "0"},
              {"GDCA_GROUP": "1", "GDCA_PANEL_HALVES": "0"}, {"GDCA_GROUP": "3", "GDCA_RAMP": "0"},
              {"GDCA_GROUP": "4", "GDCA_RAMP": "0"}, {"GDCA_GROUP": "2", "GDCA_RAMP": "0", "GDCA_MCUS": "2"},
              {"GDCA_GROUP": "3", "GDCA_RAGGED": "0"}, {"GDCA_GROUP": "4", "GDCA_SWEEP_DEBUG": "1"},
              {"GDCA_GROUP": "1", "GDCA_SWEEP_DEBUG": "1", "GDCA_MCUS": "16"},
              {"GDCA_GROUP": "1", "GDCA_SLAB": "0"}, {"GDCA_GROUP": "1", "GDCA_SLAB": "0", "GDCA_RING": "2"},
              {"GDCA_GROUP": "1", "GDCA_RING": "2"}, {"GDCA_GROUP": "1", "GDCA_RING": "3", "GDCA_MCUS": "2"},
              {"GDCA_GROUP": "1", "GDCA_MCUS": "1"},
              # the merged kernel (k_sweep_merged) carrying this one inverse (SWEEP_DEBUG bit 3), workspaces poisoned (bit 4)
              {"GDCA_SWEEP_DEBUG": "24"}, {"GDCA_SWEEP_DEBUG": "24", "GDCA_SLAB": "0"}, {"GDCA_SWEEP_DEBUG": "25", "GDCA_RING": "2"},
              {"GDCA_SWEEP_DEBUG": "24", "GDCA_MERGE_MCUS": "1"}, {"GDCA_SWEEP_DEBUG": "24", "GDCA_RAGGED": "0", "GDCA_MERGE_MCUS": "16"}]


@pytest.mark.parametrize("opts", _SCHEDULES, ids=lambda o: ",".join("%s=%s" % (k[5:], v) for k, v in o.items()))
def test_every_inverse_schedule_matches_lapack(g, opts):
    """The SPD inverse is one persistent launch that sweeps pivot groups of 1-4 blocks (the group size is chosen by matrix
    size) with its serial chain on 1-16 elected compute units; between single blocks the chain's panel and tile work is cut
    into row slabs (SLAB=0: the panel / tile items of the multi-block schedules) and Pg and the panels live in a ring of
    eight buffers (RING).  Each combination, forced through the options of a context of its own (gdca_ctx_set_option: no
    process-global switches, so all of them run in this process, beside each other), must give the LAPACK inverse on 1..14
    pivot blocks (even and odd block counts, short last groups and matrices smaller than one group included) and the same
    `info` on a non-PD matrix."""
    c = g.Context(0)
    c.set_options(**opts)
    rng = np.random.default_rng(7)
    try:
        for n in (128, 300, 640, 768, 896, 1100, 1536, 1700):      # 1 .. 14 pivot blocks, even and odd counts
            B = rng.standard_normal((n, 40))
            A = (B @ B.T) / 40 + np.diag(0.3 + rng.random(n))
            X = g.inv_cholesky(A, ctx=c)
            Xr = np.linalg.inv(A)
            rel = float(np.max(np.abs(X - Xr)) / np.max(np.abs(Xr)))
            assert rel <= 1e-10 and np.array_equal(X, X.T), (opts, n, rel)   # cond ~ 1e2: far inside the 1e-6 bar for scores
        A[5, 5] = -1.0                                            # not positive definite: leading minor 6
        with pytest.raises(g.PosDefException) as ei:
            g.inv_cholesky(A, ctx=c)
        assert ei.value.info == 6
    finally:
        c.close()


def test_context_options_are_per_context(g, ctx):
    """gdca_ctx_set_option: unknown keys and unusable values are GDCA_EINVAL and leave the context as it was; two contexts of
    one process hold different schedules and give the same bits."""
    c1, c2 = g.Context(0), g.Context(0)
    try:
        for key, val in (("NO_SUCH_SWITCH", "1"), ("GROUP", "9"), ("GROUP", "x"), ("RING", "1"), ("SWEEP_TIMEOUT_MS", "-5")):
            with pytest.raises(g.ArgumentError):
                c1.set_option(key, val)
        c1.set_options(GROUP=1, gdca_mcus=3, Hamming_Mode="full")
        c2.set_options(GROUP=4, MCUS=16, HAMMING_MODE="bound")
        rng = np.random.default_rng(3)
        n = 1300
        B = rng.standard_normal((n, 30))
        A = (B @ B.T) / 30 + np.diag(0.4 + rng.random(n))
        X1, X2, X0 = g.inv_cholesky(A, ctx=c1), g.inv_cholesky(A, ctx=c2), g.inv_cholesky(A, ctx=ctx)
        assert np.max(np.abs(X1 - X2)) <= 1e-11 * np.max(np.abs(X0)) and np.max(np.abs(X1 - X0)) <= 1e-11 * np.max(np.abs(X0))
        Z = np.asfortranarray(rng.integers(1, 22, size=(60, 900)).astype(np.int8))
        n1 = g.neighbour_counts(Z, 25, ctx=c1)
        n2 = g.neighbour_counts(Z, 25, ctx=c2)
        assert np.array_equal(n1, n2)
    finally:
        c1.close()
        c2.close()


def test_merged_inverses_equal_single_launches(g, ctx):
    """gdca_spd_inverse_batch_dev: K small matrices carried by ONE launch of k_sweep_merged (each family its own descriptor,
    flags and chain compute units; the workgroups take main-list items of the families in turn) must give, bit for bit, what K
    launches of their own give -- every K, mixed and equal sizes, ragged ends, one to 48 blocks, with the members' workspaces
    poisoned with NaNs before each run (an item that reads a panel before its producer wrote it cannot pass on the leftovers of
    an earlier identical run).  A member that is not positive definite reports its own `info`; the others are unaffected."""
    import torch

    rng = np.random.default_rng(21)

    def mat(n):
        B = rng.standard_normal((n, 24))
        return (B @ B.T) / 24 + np.diag(0.5 + rng.random(n))

    cs = [g.Context(0) for _ in range(8)]
    try:
        for rnd, ns in enumerate(([2560] * 4, [2560] * 8, [100, 1290, 3000, 777, 6016, 128, 129, 2000], [4000, 4096], [640],
                                  [1900, 1900, 1900], [5000, 300, 5000, 300, 5000])):
            As = [mat(n) for n in ns]
            ref = []
            for A in As:
                # (the reference: a launch of its own swept in the pivot groups a merged member of that size takes)
                nb = (A.shape[0] + 127) // 128
                ctx.set_option("GROUP", 4 if nb >= 24 else (2 if nb >= 12 else 1))
                d = torch.from_numpy(A).cuda()
                info = g._lib.C.c_int32()
                ctx.check(ctx.lib.gdca_spd_inverse_dev(ctx.h, g._lib.C.c_void_p(d.data_ptr()), A.shape[0], g._lib.C.byref(info)))
                ref.append(d.cpu().numpy())
                V = rng.standard_normal((A.shape[0], 2))
                assert np.max(np.abs(A @ (ref[-1] @ V) - V)) < 1e-9
            ctx.set_option("GROUP", -1)
            for merge, extra in ((8, {}), (4, {"MERGE_MCUS": 2}), (2, {"SLAB": rnd % 2}), (3, {"RING": 2 + rnd % 5})):
                K = len(ns)
                if extra.get("SLAB", 1) == 0:   # another schedule, another summation order: its own single launches are the reference
                    ctx.set_option("SLAB", 0)
                    ref_here = []
                    for A in As:
                        nb = (A.shape[0] + 127) // 128
                        ctx.set_option("GROUP", 4 if nb >= 24 else (2 if nb >= 12 else 1))
                        d = torch.from_numpy(A).cuda()
                        info = g._lib.C.c_int32()
                        ctx.check(ctx.lib.gdca_spd_inverse_dev(ctx.h, g._lib.C.c_void_p(d.data_ptr()), A.shape[0], g._lib.C.byref(info)))
                        ref_here.append(d.cpu().numpy())
                    ctx.set_options(SLAB=1, GROUP=-1)
                else:
                    ref_here = ref
                cs[0].set_options(MERGE=merge, MERGE_BLOCKS=48, MERGE_TILES=1 << 20, SWEEP_DEBUG=24, MERGE_MCUS=-1, SLAB=1, RING=8)
                cs[0].set_options(**extra)
                for c in cs[1:K]:   # (a member's schedule switches are its own context's)
                    c.set_options(SLAB=extra.get("SLAB", 1), RING=extra.get("RING", 8))
                ds = [torch.from_numpy(A).cuda() for A in As]
                infos = g.spd_inverse_batch_dev(cs[:K], [d.data_ptr() for d in ds], ns)
                assert infos == [0] * K
                for k in range(K):
                    assert np.array_equal(ds[k].cpu().numpy(), ref_here[k]), (ns, merge, extra, k)
        # one member not positive definite
        ns = [900, 1400, 700]
        As = [mat(n) for n in ns]
        As[1][300, 300] = -2.0
        ds = [torch.from_numpy(A).cuda() for A in As]
        cs[0].set_options(MERGE=4, SWEEP_DEBUG=0)
        with pytest.raises(g.PosDefException) as ei:
            g.spd_inverse_batch_dev(cs[:3], [d.data_ptr() for d in ds], ns)
        assert ei.value.info == 301
        for k in (0, 2):
            assert np.max(np.abs(ds[k].cpu().numpy() - np.linalg.inv(As[k]))) < 1e-9
        with pytest.raises(g.ArgumentError):
            g.spd_inverse_batch_dev([cs[0], cs[0]], [ds[0].data_ptr()] * 2, [900] * 2)
    finally:
        for c in cs:
            c.close()


def test_a_launch_the_watchdog_ended_gets_a_second_attempt(g):
    """The sweep's watchdog is not the end of a run (round 6): a healthy launch can lose part of its workgroups when the driver takes the
    device's queues off the hardware and back (a new stream's first command, another process starting), stands still and is ended
    by its bounded waits -- gdca_run_collect then builds the covariance again and runs the inverse once more.  SWEEP_DEBUG bit 5 lets
    the watchdog end the FIRST attempt of every inverse: every entry must return what it returns without it, bit for bit, and say so
    in gdca_stats.sweep_retries; with SWEEP_RETRIES=0 the first failure is final (GDCA_EHIP, as before round 6)."""
    import torch

    rng = np.random.default_rng(606)
    fams = [random_msa(rng, M, N) for M, N in ((500, 60), (800, 45), (400, 150), (600, 90))]       # (M, N), rows = sequences
    Zf = [np.asfortranarray(Zo.T) for Zo in fams]
    ref = g.Context(0)
    cs = [g.Context(0)]
    cs += [cs[0].peer() for _ in range(3)]
    try:
        ref.set_options(MERGE_GROUP=1)
        want = [ref.run(Z, 21, 0.8, -1.0, 0) for Z in Zf]
        want_r = [ref.run_ranked_ptr(Z.ctypes.data, Z.shape[0], Z.shape[1], 21, 0.8, -1.0, 0, 5) for Z in Zf]
        assert all(st["sweep_retries"] == 0 for _, st in want)
        for c in cs:
            c.set_options(SWEEP_DEBUG=32)
        # the fused run and its ranked form
        for k, Z in enumerate(Zf[:2]):
            S, st = cs[0].run(Z, 21, 0.8, -1.0, 0)
            assert st["sweep_retries"] == 1 and st["info"] == 0
            assert np.array_equal(S, want[k][0])
            ii, jj, sc, st2 = cs[0].run_ranked_ptr(Z.ctypes.data, Z.shape[0], Z.shape[1], 21, 0.8, -1.0, 0, 5)
            assert st2["sweep_retries"] == 1
            assert np.array_equal(ii, want_r[k][0]) and np.array_equal(jj, want_r[k][1]) and np.array_equal(sc, want_r[k][2])
        # a phase batch: its merged launch is ended, every member runs again on its own (MERGE_GROUP=1: the grouping of single launches)
        cs[0].set_options(MERGE_GROUP=1)
        dZ = [torch.from_numpy(Zo).cuda() for Zo in fams]
        dS = [torch.zeros((Zo.shape[1], Zo.shape[1]), dtype=torch.float64, device="cuda") for Zo in fams]
        g.run_dev_phased(cs, [z.data_ptr() for z in dZ], [Zo.shape[1] for Zo in fams], [Zo.shape[0] for Zo in fams], [21] * 4, 0.8, -1.0, 0,
                         [x.data_ptr() for x in dS])
        sts = [c.collect() for c in cs]
        for k in range(4):
            assert sts[k]["sweep_retries"] == 1 and sts[k]["info"] == 0, (k, sts[k])
            assert np.array_equal(dS[k].cpu().numpy(), want[k][0]), k
        # ... and ranked, from host matrices
        g.run_ranked_phased_async(cs, [Z.ctypes.data for Z in Zf], [Z.shape[0] for Z in Zf], [Z.shape[1] for Z in Zf], [21] * 4, 0.8, -1.0, 0, 5)
        for k, c in enumerate(cs):
            ii, jj, sc, st2 = c.run_ranked_collect()
            assert st2["sweep_retries"] == 1
            assert np.array_equal(ii, want_r[k][0]) and np.array_equal(jj, want_r[k][1]) and np.array_equal(sc, want_r[k][2]), k
        # the operator-level inverse
        n = 1500
        B = rng.standard_normal((n, 32))
        A = (B @ B.T) / 32 + np.diag(0.5 + rng.random(n))
        d = torch.from_numpy(A).cuda()
        assert g.spd_inverse_batch_dev(cs[:1], [d.data_ptr()], [n]) == [0]
        assert np.max(np.abs(d.cpu().numpy() - np.linalg.inv(A))) < 1e-9
        # no second attempt: the first failure is the run's
        cs[0].set_options(SWEEP_RETRIES=0)
        with pytest.raises(g.GdcaError) as ei:
            cs[0].run(Zf[0], 21, 0.8, -1.0, 0)
        assert "timed out" in str(ei.value)
        cs[0].set_options(SWEEP_RETRIES=2, SWEEP_DEBUG=0)
        S, st = cs[0].run(Zf[0], 21, 0.8, -1.0, 0)
        assert st["sweep_retries"] == 0 and np.array_equal(S, want[0][0])
    finally:
        for c in reversed(cs):
            c.close()
        ref.close()


def test_merged_sweep_watchdog(g):
    """The merged launch keeps the promise of the single one: when a member's chain never runs (SWEEP_DEBUG bit 1: nobody is
    elected) the bounded waits end the launch, EVERY member of it is reported as aborted (no member's result can be trusted
    once a workgroup has dropped its items), and the contexts stay usable."""
    import time

    import torch

    rng = np.random.default_rng(2)
    cs = [g.Context(0) for _ in range(3)]
    try:
        ns = [1500, 2000, 1000]
        As = []
        for n in ns:
            B = rng.standard_normal((n, 24))
            As.append((B @ B.T) / 24 + np.diag(0.5 + rng.random(n)))
        for c in cs:
            c.set_options(SWEEP_DEBUG=0, SWEEP_TIMEOUT_MS=300)
        cs[1].set_options(SWEEP_DEBUG=2)               # this member's chain gets no compute unit
        cs[0].set_options(MERGE=4)
        ds = [torch.from_numpy(A).cuda() for A in As]
        t = time.time()
        with pytest.raises(g.GdcaError) as ei:
            g.spd_inverse_batch_dev(cs, [d.data_ptr() for d in ds], ns)
        assert "timed out" in str(ei.value) and time.time() - t < 20.0
        cs[1].set_options(SWEEP_DEBUG=0)
        ds = [torch.from_numpy(A).cuda() for A in As]
        assert g.spd_inverse_batch_dev(cs, [d.data_ptr() for d in ds], ns) == [0, 0, 0]
        for k in range(3):
            assert np.max(np.abs(ds[k].cpu().numpy() - np.linalg.inv(As[k]))) < 1e-9
    finally:
        for c in cs:
            c.close()


_TRACE_SCRIPT = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
import gaussdca.jl_amd as g
ctx = g.Context(0)
rng = np.random.default_rng(11)
n = 1536
B = rng.standard_normal((n, 40))
A = (B @ B.T) / 40 + np.diag(0.3 + rng.random(n))
X = g.inv_cholesky(A, ctx=ctx)
Xr = np.linalg.inv(A)
print(json.dumps({"rel": float(np.max(np.abs(X - Xr)) / np.max(np.abs(Xr)))}))
"""


@pytest.mark.gpu
@pytest.mark.parametrize("group,items", [("1", 12 + 11 * 8 + 10 * 16), ("4", None)])
def test_sweep_trace_lists_every_chain_item(tmp_path, group, items):
    """GDCA_SWEEP_TRACE (the kernel's own time stamps, tools/sweep_trace.py): the inverse stays correct with the trace on, and the
    file lists every item of the serial chain with end >= start -- for single blocks (n = 1536: 12 pivots, 11 x 8 slab items of
    the next row block, 10 x (8 + 8) for the tile below it and the row after) and for a multi-block schedule."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    trace = tmp_path / "trace.txt"
    e = dict(os.environ, GDCA_GROUP=group, GDCA_SWEEP_TRACE=str(trace))
    r = subprocess.run([sys.executable, "-c", _TRACE_SCRIPT, root], capture_output=True, text=True, env=e, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["rel"] <= 1e-10
    lines = trace.read_text().splitlines()
    head = [l for l in lines if l.startswith("#")]
    rows = [l.split() for l in lines if not l.startswith("#") and not l.startswith("m ")]
    # (round 6: every main-list item too -- "m index kind p a b workgroup taken ready done", what tools/critical_path.py walks)
    mains = [l.split() for l in lines if l.startswith("m ")]
    assert len(mains) > 100 and all(float(m[9]) >= float(m[8]) >= float(m[7]) - 1e-9 for m in mains if int(m[2]) != 3)
    assert head[0].startswith("# nblk 12 g %s" % group)
    assert any(l.startswith("# pivot items: 12;") for l in head)
    if items is not None:
        assert len(rows) == items
    assert len(rows) > 12 and all(float(b) >= float(a) > -1e-9 for _, _, a, b in rows)


@pytest.mark.parametrize("N,M,score,pc", [(430, 4000, "frob", 0.8), (260, 5000, "DI", 0.2)])
def test_mid_size_families_match_oracle(g, ctx, o, N, M, score, pc):
    """Whole hot path against the oracle at sizes where the production schedules are active: N = 430 (n = 8600,
    68 pivot blocks: groups of three pivots, K = 384 trailing updates) and N = 260 (41 blocks: single pivots).
    Integers bit-exact, scores within 1e-6 relative (north_star)."""
    from gaussdca.jl_amd import synth

    Zo = synth.synth_family(N, M, 21, 0xE100 + N)
    S, st = ctx.run(np.asfortranarray(Zo.T), 21, pc, -1.0, 1 if score == "DI" else 0)
    W_o, Meff_o, th_o, thr_o = o.compute_weights(Zo, "auto")
    assert st["theta"] == th_o and st["thresh"] == thr_o and st["Meff"] == Meff_o and st["info"] == 0
    S_o = o.scores_from_Z(Zo, 21, pc, "auto", score)
    ok, max_rel, max_abs = score_close(S, S_o, rtol=1e-6, atol_frac=1e-9)
    assert ok, (max_rel, max_abs)
    R, R_o = g.compute_ranking(S, 5), o.compute_ranking(S_o, 5)
    assert [t[:2] for t in R[:200]] == [t[:2] for t in R_o[:200]]       # the contacts a user looks at: same order


@pytest.mark.parametrize("seed", [99, 2024, 31337])
def test_random_mid_size_campaign(g, ctx, o, seed):
    """Random campaign (the body of tools/campaign_midsize.py with fixed seeds): 8 synthetic families per seed with N drawn
    from [60, 440) -- 10 .. 69 pivot blocks, i.e. every group size the schedule rule picks below n = 9000 -- and M from
    [600, 3000), alternating :frob / :DI, whole hot path against the oracle: theta, threshold and Meff equal, scores within
    1e-6 relative."""
    from gaussdca.jl_amd import synth

    rng = np.random.default_rng(seed)
    for k in range(8):
        N = int(rng.integers(60, 440))
        M = int(rng.integers(600, 3000))
        score = "frob" if k % 2 == 0 else "DI"
        pc = 0.8 if score == "frob" else 0.2
        Zo = synth.synth_family(N, M, 21, int(rng.integers(1, 2**31 - 1)))
        S, st = ctx.run(np.asfortranarray(Zo.T), 21, pc, -1.0, 1 if score == "DI" else 0)
        W_o, Meff_o, th_o, thr_o = o.compute_weights(Zo, "auto")
        assert st["theta"] == th_o and st["thresh"] == thr_o and st["Meff"] == Meff_o and st["info"] == 0, (seed, k, N, M)
        S_o = o.scores_from_Z(Zo, 21, pc, "auto", score)
        ok, max_rel, max_abs = score_close(S, S_o, rtol=1e-6, atol_frac=1e-9)    # 1e-6 relative
        assert ok, (seed, k, N, M, score, max_rel, max_abs)


_WATCHDOG_SCRIPT = r"""
import os, sys, json, time
import numpy as np
sys.path.insert(0, sys.argv[1])
import gaussdca.jl_amd as g
ctx = g.Context(0)
rng = np.random.default_rng(5)
n = int(sys.argv[2])
B = rng.standard_normal((n, 40))
A = (B @ B.T) / 40 + np.diag(0.3 + rng.random(n))
out = {}
t = time.time()
try:
    X = g.inv_cholesky(A, ctx=ctx)
    out["status"] = "ok"
    out["rel"] = float(np.max(np.abs(X - np.linalg.inv(A))) / np.max(np.abs(X)))
except g.GdcaError as e:
    out["status"] = "ehip"
    out["msg"] = str(e)
out["seconds"] = time.time() - t
Z = np.asfortranarray(rng.integers(1, 22, size=(30, 200)).astype(np.int8))
out["theta_after"] = float(g.compute_theta(Z, ctx=ctx))          # the context and the device are still usable
print(json.dumps(out))
"""


def _run_watchdog(env, n):
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _WATCHDOG_SCRIPT, root, str(n)], capture_output=True, text=True,
                       env=dict(os.environ, **env), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("n", [1500, 9000])
def test_chain_election_does_not_need_xcc0(n):
    """The pivot chain's compute units are elected on whichever XCD the first workgroup of the launch runs on.
    GDCA_SWEEP_DEBUG=1 keeps every workgroup that runs on XCC 0 out of the election (what a CU mask, a partitioned mode
    or another tenant on that XCD would do): the inverse must still be the LAPACK inverse."""
    out = _run_watchdog({"GDCA_SWEEP_DEBUG": "1"}, n)
    assert out["status"] == "ok" and out["rel"] <= 1e-10, out


def test_sweep_watchdog_returns_a_status_instead_of_hanging():
    """GDCA_SWEEP_DEBUG=2 elects nobody for the chain: Pg(0) is never produced and every workgroup of the launch ends up
    waiting for it.  The bounded waits (GDCA_SWEEP_TIMEOUT_MS, here 300 ms; default 4 s) must turn that into GDCA_EHIP
    within seconds, and the context must stay usable."""
    out = _run_watchdog({"GDCA_SWEEP_DEBUG": "2", "GDCA_SWEEP_TIMEOUT_MS": "300"}, 3000)
    assert out["status"] == "ehip" and "timed out" in out["msg"], out
    assert out["seconds"] < 20.0, out
    assert 0.0 < out["theta_after"] <= 0.5


_CUMASK_SCRIPT = r"""
import ctypes, json, os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
import gaussdca.jl_amd as g
lib = g.load()
hip = ctypes.CDLL("libamdhip64.so")                       # the runtime libgdca.so is already linked against
n = int(sys.argv[2])
# CU-mask bit i <-> XCD i % 8 (measured in round 2, profiles/r02_ubench_cumap.log): drop every compute unit of XCC 0 and,
# for good measure, half of XCC 3 -- no workgroup of any launch on this stream can then run on XCC 0
words = [0] * 8
for i in range(256):
    if i % 8 != 0 and not (i % 8 == 3 and (i // 8) % 2 == 0):
        words[i // 32] |= 1 << (i % 32)
mask = (ctypes.c_uint32 * 8)(*words)
stream = ctypes.c_void_p()
rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(stream), 8, mask)
assert rc == 0, rc
ctx = g.Context(0, stream=stream.value)
rng = np.random.default_rng(5)
B = rng.standard_normal((n, 40))
A = (B @ B.T) / 40 + np.diag(0.3 + rng.random(n))
t = time.time()
X = g.inv_cholesky(A, ctx=ctx)
out = {"rel": float(np.max(np.abs(X - np.linalg.inv(A))) / np.max(np.abs(X))), "seconds": time.time() - t,
       "enabled_cus": sum(bin(w).count("1") for w in words)}
# ... and a merged launch on the same masked stream: three families, each chain claimed by whichever XCD gets there
cs = [ctx] + [g.Context(0, stream=stream.value) for _ in range(2)]
ns = [1500, 2300, 900]
As = []
for m in ns:
    B = rng.standard_normal((m, 40))
    As.append((B @ B.T) / 40 + np.diag(0.3 + rng.random(m)))
ds = [g.DeviceBuffer.from_array(c, a) for c, a in zip(cs, As)]
ctx.set_options(MERGE=4)
g.spd_inverse_batch_dev(cs, [d.ptr for d in ds], ns)
out["rel_merged"] = max(float(np.max(np.abs(d.download((m, m)) - np.linalg.inv(a))) / np.max(np.abs(np.linalg.inv(a)))) for d, a, m in zip(ds, As, ns))
print(json.dumps(out))
"""


@pytest.mark.parametrize("n", [2000, 9100])
def test_inverse_on_a_cu_masked_stream_without_xcc0(n):
    """The case the run-time election of the chain's XCD exists for: a caller's stream whose CU mask excludes every
    compute unit of XCC 0 (gdca_ctx_create_on_stream).  The persistent sweep kernel is launched with two workgroups per
    compute unit of the WHOLE device, so on 208 enabled units a third of them are not resident at first: items are handed
    out in list order, an item only ever waits for items somebody already holds, and the chain's workers are elected among
    the workgroups that do run.  Must return the LAPACK inverse (it used to spin for ever: no workgroup on XCC 0, nobody
    elected)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _CUMASK_SCRIPT, root, str(n)], capture_output=True, text=True,
                       env=dict(os.environ, GDCA_SWEEP_TIMEOUT_MS="3000"), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["enabled_cus"] == 208 and out["rel"] <= 1e-10 and out["rel_merged"] <= 1e-10, out


def test_two_ungated_sweeps_side_by_side():
    """two sweep launches that nothing orders against each other -- two contexts that are NOT peers, each driven by a
    thread of its own, and then two processes -- on one GPU, n = 9100 (72 pivot blocks, groups of four): the persistent kernels share
    the compute units, both finish, and every inverse is LAPACK's (tools/side_by_side_probe.py; a dependency wait that ran out of
    time would surface as GDCA_EHIP and a non-zero exit)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "side_by_side_probe.py"), "9100", "2"], capture_output=True, text=True,
                       env=dict(os.environ, GDCA_SWEEP_TIMEOUT_MS="6000"), timeout=600)
    print(r.stdout[-1500:])
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "two processes: exit codes [0, 0]" in r.stdout
    # ... and MERGED launches of eight beside an ungated sweep of n rows (the first workgroup of a merged launch to show up stays a
    # main-list worker: with the rest of its grid kept out by the other launch its lists still move)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "side_by_side_probe.py"), "9100", "2", "merged"], capture_output=True, text=True,
                       env=dict(os.environ, GDCA_SWEEP_TIMEOUT_MS="8000"), timeout=600)
    print(r.stdout[-800:])
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    # ... and the same from TWO PROCESSES: merged launches of eight in one, n = 9100 sweeps in the other, twice over
    for _ in range(2):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "side_by_side_probe.py"), "9100", "3", "merged_procs"], capture_output=True,
                           text=True, env=dict(os.environ, GDCA_SWEEP_TIMEOUT_MS="8000"), timeout=600)
        print(r.stdout[-600:])
        assert r.returncode == 0 and "exit codes [0, 0]" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
