"""Shared helpers for the parity tests: the reference test-suite's cases and comparator
(/root/reference/test/runtests.jl:29-50) plus the tie rule of SURVEY.md 4.3."""
from itertools import groupby

import numpy as np

# the four golden cases of test/runtests.jl:52-76 (keyword names as in src/GaussDCA.jl:10-15,
# with theta spelled out)
CASES = {
    "small.FNRout.txt": dict(fasta="small.fasta.gz", kw={}),
    "small.DIRout.txt": dict(fasta="small.fasta.gz", kw=dict(pseudocount=0.2, score="DI", remove_dups=True)),
    "small.DIRout2.txt": dict(fasta="small.fasta.gz",
                              kw=dict(pseudocount=0.2, score="DI", theta=0.0, max_gap_fraction=0.8,
                                      min_separation=4)),
    "large.DIRout.txt": dict(fasta="large.fasta.gz", kw=dict(pseudocount=0.2, score="DI", remove_dups=True)),
}


def parse_golden(path):
    """'i j score' lines -> ({(i,j): (float, str)}, [(i,j) in file order])  (runtests.jl:29-39)"""
    d, order = {}, []
    with open(path) as f:
        for line in f:
            sl = line.split()
            if not sl:
                continue
            assert len(sl) == 3
            k = (int(sl[0]), int(sl[1]))
            assert k not in d
            d[k] = (float(sl[2]), sl[2])
            order.append(k)
    return d, order


def _order_mod_ties(pairs_scores):
    """Stable re-sort inside groups of equal 7-digit printed score (SURVEY.md 4.3 tie rule)."""
    out = []
    for _, grp in groupby(pairs_scores, key=lambda t: t[1]):
        out.extend(sorted(k for k, _ in grp))
    return out


def compare_with_golden(R, golden_path):
    d, order = parse_golden(golden_path)
    keys = [(i, j) for i, j, _ in R]
    rep = dict(rows=len(R), keys_equal=(set(keys) == set(d) and len(keys) == len(d)))
    if not rep["keys_equal"]:
        return rep
    max_rel, mism = 0.0, 0
    for i, j, x in R:
        g, gs = d[(i, j)]
        max_rel = max(max_rel, abs(x - g) / abs(g))
        if ("%e" % x) != gs:
            mism += 1
    rep["max_rel"] = max_rel
    rep["string_mismatches"] = mism
    rep["order_equal"] = keys == order
    mine = _order_mod_ties([((i, j), "%e" % x) for i, j, x in R])
    theirs = _order_mod_ties([(k, d[k][1]) for k in order])
    rep["order_equal_mod_ties"] = mine == theirs
    return rep


def score_close(S, S_ref, rtol=1e-6, atol_frac=1e-9, atol_abs=0.0):
    """Elementwise |S - S_ref| <= rtol |S_ref| + atol_frac max|S_ref| + atol_abs on the off-diagonal.

    rtol = 1e-6 is north_star's bar for FN/DI scores; the small absolute term only covers
    APC-corrected scores that cross zero (their relative error is unbounded by construction).
    Returns (ok, max_rel_over_entries_above_1e-3_of_max, max_abs)."""
    N = S.shape[0]
    off = ~np.eye(N, dtype=bool)
    a, b = S[off], S_ref[off]
    # magnitude reference: all entries, diagonal included (the APC-corrected diagonal is -S_i.^2 / Sa, i.e. the
    # size of the raw scores; with few sequences every off-diagonal entry can cancel to rounding noise)
    scale = np.max(np.abs(S_ref))
    ok = bool(np.all(np.abs(a - b) <= rtol * np.abs(b) + atol_frac * scale + atol_abs))
    big = np.abs(b) > 1e-3 * scale
    max_rel = float(np.max(np.abs(a[big] - b[big]) / np.abs(b[big]))) if np.any(big) else 0.0
    return ok, max_rel, float(np.max(np.abs(a - b)))


def random_msa(rng, M, N, q=21, gap_runs=True, clusters=None):
    """Small seeded 'Pfam-like' alignment: cluster centres + per-sequence mutation + gap runs."""
    root = rng.integers(1, q, size=N)
    K = clusters or max(1, M // 25)
    centres = np.tile(root, (K, 1))
    cm = rng.random((K, N)) < 0.25
    centres[cm] = rng.integers(1, q, size=int(cm.sum()))
    Z = centres[rng.integers(0, K, size=M)]
    mu = rng.choice([0.02, 0.05, 0.1, 0.2, 0.3, 0.5], size=M)
    mask = rng.random((M, N)) < mu[:, None]
    Z[mask] = rng.integers(1, q, size=int(mask.sum()))
    if gap_runs:
        for k in range(M):
            for _ in range(rng.integers(0, 4)):
                a = rng.integers(0, N)
                Z[k, a:a + rng.integers(1, max(2, N // 10) + 1)] = q
    return np.ascontiguousarray(Z.astype(np.int8))
