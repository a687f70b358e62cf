"""Pins the CPU oracle against the reference's own golden vectors.

Mirrors /root/reference/test/runtests.jl: test1 (:52-68), test2 (:70-76); test3 (:78-86)
repeats test1's second case through DCAUtils' non-bit-packed Hamming path, which the
oracle covers by checking its two Hamming forms against each other.
"""
import json
import os

import numpy as np
import pytest

from oracle import gdca_oracle as o

from gdca_testutil import CASES, compare_with_golden, parse_golden


@pytest.mark.parametrize("golden", list(CASES))
def test_oracle_reproduces_reference_golden(golden, refdata):
    c = CASES[golden]
    R, inter = o.gDCA(os.path.join(refdata, c["fasta"]), return_intermediates=True, **c["kw"])
    rep = compare_with_golden(R, os.path.join(refdata, golden))
    # identical key sets (runtests.jl:44-45), every score within 1e-6 relative (7-digit prints)
    assert rep["keys_equal"]
    assert rep["max_rel"] <= 1e-6, rep
    # the restatement reproduces every printed digit; allow isolated 7th-digit boundary flips
    assert rep["string_mismatches"] <= 2, rep
    assert rep["order_equal_mod_ties"], rep
    if not golden.startswith("large"):
        assert rep["order_equal"], rep
    with open(os.path.join(os.path.dirname(refdata), "intermediates.json")) as f:
        want = json.load(f)[golden]
    assert inter["thresh"] == want["thresh"]
    assert inter["M"] == want["M"] and inter["N"] == want["N"] and inter["q"] == want["q"]
    assert inter["theta"] == want["theta"]
    assert inter["Meff"] == want["Meff"]


@pytest.mark.parametrize("golden", list(CASES))
def test_meff_candidates_agree_to_a_few_ulp(golden, refdata):
    """`Meff = sum(W)` is Julia's pairwise SIMD sum in the reference (DCAUtils, call site src/GaussDCA.jl:28; printed by
    test/runtests.jl's runs, compared by none of its goldens): no f64 evaluation order pins its last bit.  The fixtures record the
    three candidates side by side -- exact (what the oracle and the HIP path return), left to right, Julia's pairwise recursion with
    1024-element base blocks -- the oracle reproduces each, and they agree to <= 3 ulp on every reference input (measured: 0, 0, 2 and
    3 ulp; with M <= 1024 the pairwise form IS the left-to-right one), so north_star's "Meff bit-exact" is decidable on a box with
    Julia and off by at most that here."""
    c = CASES[golden]
    kw = c["kw"]
    Z = o.read_fasta_alignment(os.path.join(refdata, c["fasta"]), kw.get("max_gap_fraction", 0.9))
    if kw.get("remove_dups"):
        Z, _ = o.remove_duplicate_sequences(Z)
    W, Meff, _, _ = o.compute_weights(Z, kw.get("theta", "auto"))
    with open(os.path.join(os.path.dirname(refdata), "intermediates.json")) as f:
        want = json.load(f)[golden]
    got = o.meff_three_ways(W)
    assert got == want["Meff_candidates"]
    assert got["exact"] == Meff == want["Meff"]
    ulp = np.spacing(got["exact"])
    assert max(got.values()) - min(got.values()) <= 3 * ulp, (got, ulp)


def test_golden_smoke_kats(refdata):
    # SURVEY.md 4.3 "Smoke KATs": first and last rows of each reference golden
    kats = {
        "small.FNRout.txt": ("11 35 3.649475e+00", "9 46 -6.752293e-01"),
        "small.DIRout.txt": ("11 35 7.878672e-01", "4 23 -2.242551e-01"),
        "small.DIRout2.txt": ("11 35 8.057298e-01", "4 23 -2.190418e-01"),
        "large.DIRout.txt": ("14 21 3.876863e-01", "136 232 -7.159769e-02"),
    }
    for name, (first, last) in kats.items():
        lines = open(os.path.join(refdata, name)).read().split("\n")
        lines = [ln for ln in lines if ln]
        assert lines[0] == first and lines[-1] == last
        d, order = parse_golden(os.path.join(refdata, name))
        assert len(d) == len(order)


def _random_msa(rng, M, N, q=21, gap_runs=True):
    root = rng.integers(1, q, size=N)
    Z = np.tile(root, (M, 1))
    mu = rng.choice([0.02, 0.1, 0.3, 0.6], size=M)
    mask = rng.random((M, N)) < mu[:, None]
    Z[mask] = rng.integers(1, q, size=int(mask.sum()))
    if gap_runs:
        for k in range(M):
            a = rng.integers(0, N)
            Z[k, a:a + rng.integers(0, max(2, N // 8))] = q
    return np.ascontiguousarray(Z.astype(np.int8))


def test_theta_closed_form_equals_all_pairs():
    rng = np.random.default_rng(7)
    for (M, N) in [(2, 1), (17, 5), (120, 33), (301, 64)]:
        Z = _random_msa(rng, M, N)
        assert o.pair_identity_sum(Z) == o.pair_identity_sum_allpairs(Z)


def test_c_accelerators_match_numpy_forms(monkeypatch):
    if not o.have_c_kernels():
        pytest.skip("oracle C kernels not built")
    rng = np.random.default_rng(11)
    Z = _random_msa(rng, 150, 23)
    q = 21
    thresh = 9
    n_c = o.neighbour_counts(Z, thresh)
    W, Meff = o.weights_from_counts(n_c)
    Pi_c, Pij_c = o.compute_frequencies(Z, q, W, Meff)
    Pi, Pij = o.add_pseudocount(Pi_c, Pij_c, 0.5, q)
    mJ = o.spd_inverse(o.compute_C(Pi, Pij))
    FN_c = o.compute_FN(mJ, q)
    monkeypatch.setattr(o, "_ck", False)
    n_np = o.neighbour_counts(Z, thresh)
    assert np.array_equal(n_c, n_np)  # integers: bit exact
    Pi_np, Pij_np = o.compute_frequencies(Z, q, W, Meff)
    assert np.allclose(Pi_c, Pi_np, rtol=1e-13, atol=1e-16)
    assert np.allclose(Pij_c, Pij_np, rtol=1e-13, atol=1e-16)
    assert np.array_equal(Pij_c, Pij_c.T)
    FN_np = o.compute_FN(mJ, q)
    assert np.allclose(FN_c, FN_np, rtol=1e-12, atol=1e-14)


def test_weights_rules():
    rng = np.random.default_rng(3)
    Z = _random_msa(rng, 60, 53)
    # theta == 0 -> W = 1, Meff = M (early exit);  thresh == 0 (tiny theta) -> n_k = 1 as well
    W, Meff, th, thresh = o.compute_weights(Z, 0.0)
    assert Meff == 60.0 and np.all(W == 1.0) and thresh == 0
    W, Meff, th, thresh = o.compute_weights(Z, 0.01)
    assert thresh == 0 and Meff == 60.0
    # duplicates are neighbours of each other for any thresh >= 1 (strict '<': d = 0 < 1)
    Zu, _ = o.remove_duplicate_sequences(Z)
    Mu = Zu.shape[0]
    Z2 = np.concatenate([Zu, Zu[:5]], axis=0)
    n = o.neighbour_counts(Z2, 1)
    assert np.all(n[:5] == 2) and np.all(n[Mu:] == 2) and np.all(n[5:Mu] == 1)


def test_pseudocount_and_covariance_rules():
    rng = np.random.default_rng(5)
    Z = _random_msa(rng, 40, 7)
    q, s = 21, 20
    Pi_t, Pij_t, Meff, W = o.compute_weighted_frequencies(Z, q, 0.3)
    # diagonal blocks of Pij_true are diagonal with Pi on the diagonal (rule 5)
    for i in range(7):
        blk = Pij_t[i * s:(i + 1) * s, i * s:(i + 1) * s]
        assert np.allclose(np.diag(blk), Pi_t[i * s:(i + 1) * s], rtol=1e-14, atol=0)
        assert np.count_nonzero(blk - np.diag(np.diag(blk))) == 0
    Pi, Pij = o.add_pseudocount(Pi_t, Pij_t, 0.8, q)
    pcq = 0.8 / q
    assert np.allclose(Pij[0:s, s:2 * s], 0.2 * Pij_t[0:s, s:2 * s] + pcq / q)
    assert np.allclose(Pij[0:s, 0:s], 0.2 * Pij_t[0:s, 0:s] + pcq * np.eye(s))
    C = o.compute_C(Pi, Pij)
    assert np.array_equal(C, C.T)
    assert np.all(np.linalg.eigvalsh(C) > 0)
    with pytest.raises(o.NotPositiveDefinite):
        Pi0, Pij0 = o.add_pseudocount(Pi_t, Pij_t, 0.0, q)
        o.spd_inverse(o.compute_C(Pi0, Pij0))


@pytest.mark.parametrize("pc", [0.8, 0.2, 0.02, 0.001])
def test_pseudocount_bounds_the_smallest_eigenvalue_of_the_covariance(refdata, pc):
    """What the device path's refinement screen rests on (gdca_api.hip, cond_bound): with pseudocount pc the covariance is that of a
    mixture with weight pc on independent uniform columns, so lambda_min(C) >= pc / q^2 and cond_2(C) <= ||C||_1 q^2 / pc -- on the
    reference's own alignment, with and without gaps in every column (q = 21 and a gap-free q = 20 one), and on a random one."""
    Zs = [o.remove_duplicate_sequences(o.read_fasta_alignment(os.path.join(refdata, "small.fasta.gz"), 0.9))[0]]
    Zs.append(np.where(Zs[0] == 21, 1, Zs[0]).astype(np.int8))       # no gaps at all: q = 20
    Zs.append(_random_msa(np.random.default_rng(8), 60, 9))
    for Z in Zs:
        q = int(Z.max())
        Pi_t, Pij_t, Meff, W = o.compute_weighted_frequencies(Z, q, "auto")
        C = o.compute_C(*o.add_pseudocount(Pi_t, Pij_t, pc, q))
        lam = np.linalg.eigvalsh(C)
        assert lam[0] >= pc / q ** 2 * (1 - 1e-8), (q, pc, lam[0], pc / q ** 2)
        assert lam[-1] / lam[0] <= np.abs(C).sum(axis=0).max() * q * q / pc * (1 + 1e-8)


def test_apc_and_ranking_rules():
    rng = np.random.default_rng(9)
    S = rng.random((12, 12))
    S = S + S.T
    np.fill_diagonal(S, 0.0)
    A = o.correct_APC(S)
    Sa = S.sum() * (1 - 1 / 12)
    assert np.allclose(A[3, 7], S[3, 7] - S[3].sum() * S[:, 7].sum() / Sa)
    R = o.compute_ranking(A, 5)
    assert len(R) == (12 - 5) * (12 - 5 + 1) // 2
    assert all(j >= i + 5 for i, j, _ in R)
    assert all(R[t][2] >= R[t + 1][2] for t in range(len(R) - 1))
    assert o.format_rank([(11, 35, 3.649475)]) == "11 35 3.649475e+00\n"
    # stable: exact ties keep generation order (i-major, j-minor)
    T = np.zeros((8, 8))
    assert [(i, j) for i, j, _ in o.compute_ranking(T, 5)] == [(1, 6), (1, 7), (1, 8), (2, 7), (2, 8), (3, 8)]


def test_argument_checks(refdata, tmp_path):
    f = os.path.join(refdata, "small.fasta.gz")
    for kw in (dict(pseudocount=1.5), dict(theta=-0.1), dict(theta="bogus"), dict(max_gap_fraction=2),
               dict(score="plm"), dict(min_separation=0)):
        with pytest.raises(ValueError):
            o.gDCA(f, **kw)
    with pytest.raises(ValueError):
        o.gDCA(str(tmp_path / "missing.fasta"))


@pytest.mark.parametrize("name", ["synthetic_a", "synthetic_b", "synthetic_c"])
def test_oracle_reproduces_committed_synthetic_vectors(name, refdata):
    """tests/golden/synthetic_*.npz (made by tests/golden/make_synthetic.py): the oracle still produces them."""
    d = np.load(os.path.join(os.path.dirname(refdata), name + ".npz"))
    Z, q, pc = d["Z"], int(d["q"]), float(d["pseudocount"])
    theta = "auto" if str(d["theta_in"]) == "auto" else float(str(d["theta_in"]))
    W, Meff, th, thresh = o.compute_weights(Z, theta)
    assert th == float(d["theta"]) and thresh == int(d["thresh"]) and Meff == float(d["Meff"])
    assert np.array_equal(o.neighbour_counts(Z, thresh), d["n_k"])
    assert o.pair_identity_sum(Z) == int(d["pair_identity_sum"])
    for score, key in (("frob", "S_frob"), ("DI", "S_DI")):
        S = o.scores_from_Z(Z, q, pc, theta, score)
        assert np.allclose(S, d[key], rtol=1e-10, atol=1e-13)
