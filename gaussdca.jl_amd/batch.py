"""Multi-GPU model: an embarrassingly parallel batch of independent protein families, one
process (one gdca context) per GPU, no collective on the data path.

Every rank computes the same deterministic assignment from the (N, M) sizes alone, so no
communication is needed to agree on it: longest-processing-time-first over the cost model
    c = [A3 n^3 + B1 b + B2 b^2] + [H2 M^2 N + H1 M] + [T2 N^2 M + T1 N^2 + T0],   n = N (q-1), b = ceil(n / 128)
(SPD inverse: the n^3 of the tile updates plus the per-block cost of its serial pivot chain; all-pairs Hamming; pair tallies,
covariance and scores with their per-family fixed cost; seconds), ties broken by family index.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

# seconds, least squares on the measured per-family stage times of all 256 families of BASELINE.json's batch configuration run one
# after the other on one MI355X (round 6: profiles/r06_E_per_family.json, tools/fit_batch_model.py): the model's per-family totals
# are within 0.92 .. 1.08 of the measured ones (5 % .. 95 %), its LPT makespans within 1 % of those computed from the measured times
# at 2, 4 and 8 ranks (tests/test_batch_sharding.py).  (Rounds 3-5: max(17e-15 n^3, 61e-6 b) + 2.0e-15 M^2 N + 2.4e-13 N^2 M, fitted to
# configs B, C and D alone: 6 % low on the batch, small families -- whose fixed costs it ignored -- 34 % low.)
A3, B1, B2 = 1.877e-14, 8.10e-5, -1.385e-6
H2, H1 = 1.70e-15, 1.40e-8
T2, T1, T0 = 2.19e-13, 9.1e-10, 2.6e-4


def family_cost(N: int, M: int, q: int = 21) -> float:
    """Estimated device seconds of one family (the LPT weights; only their ratios matter for the sharding)."""
    n = float(N * (q - 1))
    b = float(-(-(N * (q - 1)) // 128))
    # (the quadratic in b is a fit over 16 .. 94 blocks: beyond, the chain's share is held at its value there -- n^3 dominates anyway)
    bb = min(b, 94.0)
    inverse = max(A3 * n ** 3 + B1 * bb + B2 * bb * bb, 4.5e-5 * min(b, 20.0))  # (never below a short chain's own time)
    return inverse + H2 * float(M) * M * N + H1 * M + T2 * float(N) * N * M + T1 * float(N) * N + T0


def shard_families(sizes: Sequence[Tuple[int, int]], world: int, q: int = 21) -> List[List[int]]:
    """sizes[f] = (N, M).  Returns, for every rank, the list of family indices it processes
    (in processing order, most expensive first).  Deterministic; identical on every rank."""
    if world < 1:
        raise ValueError("world must be >= 1")
    order = sorted(range(len(sizes)), key=lambda f: (-family_cost(sizes[f][0], sizes[f][1], q), f))
    load = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for f in order:
        r = min(range(world), key=lambda x: (load[x], x))
        out[r].append(f)
        load[r] += family_cost(sizes[f][0], sizes[f][1], q)
    return out


def batch_sizes(n_families: int, seed: int = 0xE000) -> List[Tuple[int, int]]:
    """BASELINE.json configs[4]: Pfam-like families, N in [100, 600], M in [5k, 80k] (SplitMix64)."""
    def splitmix(x):
        x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return x, z ^ (z >> 31)

    out = []
    for f in range(n_families):
        st = seed + f
        st, h1 = splitmix(st)
        st, h2 = splitmix(st)
        out.append((100 + h1 % 501, 5000 + h2 % 75001))
    return out
