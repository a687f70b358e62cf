"""Multi-GPU model: an embarrassingly parallel batch of independent protein families, one
process (one gdca context) per GPU, no collective on the data path.

Every rank computes the same deterministic assignment from the (N, M) sizes alone, so no
communication is needed to agree on it: longest-processing-time-first over the cost model
    c = max(alpha (N (q-1))^3, chain * blocks) + beta M^2 N + gamma N^2 M
(SPD inverse, bound below by its pivot chain, + all-pairs Hamming + pair tallies; seconds), ties broken by family index.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

# seconds, fitted to the round-3 MI355X stage times (profiles/r03_bench_*.json):
#   SPD inverse   17.3 ms at n = 10 000 (config C), 125.7 ms at n = 20 000 (D)  ->  ALPHA n^3, but never less than the pivot
#                 chain, CHAIN seconds per 128-block (config B: 20 blocks, 1.23 ms: small matrices are bound by that chain)
#   reweighting   2.96 ms at M^2 N = 1.25e12 (C), 18.5 ms at 1e13 (D)          ->  BETA M^2 N
#   tallies + covariance  3.3 ms at N^2 M = 1.25e10 (C), 21 ms at 1e11 (D)     ->  GAMMA N^2 M
ALPHA, BETA, GAMMA, CHAIN = 17.0e-15, 2.0e-15, 2.4e-13, 61e-6


def family_cost(N: int, M: int, q: int = 21) -> float:
    """Estimated device seconds of one family (the LPT weights; only their ratios matter for the sharding)."""
    n = N * (q - 1)
    inverse = max(ALPHA * float(n) ** 3, CHAIN * -(-n // 128))
    return inverse + BETA * float(M) * M * N + GAMMA * float(N) * N * M


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
