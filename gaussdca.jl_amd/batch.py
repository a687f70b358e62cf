"""Multi-GPU model: an embarrassingly parallel batch of independent protein families, one
process (one gdca context) per GPU, no collective on the data path.

Every rank computes the same deterministic assignment from the (N, M) sizes alone, so no
communication is needed to agree on it: longest-processing-time-first over the cost model
    c = alpha (N (q-1))^3 + beta M^2 N + gamma N^2 M
(SPD inverse + all-pairs Hamming + pair tallies), ties broken by family index.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

# relative weights fitted to the MI355X stage times at N=500, M=50k, q=21:
# inverse 24.8 ms / 1e12, Hamming 3.4 ms / 1.25e12, tallies 5.9 ms / 1.25e10
ALPHA, BETA, GAMMA = 24.8e-12, 2.7e-12, 4.7e-10


def family_cost(N: int, M: int, q: int = 21) -> float:
    n = N * (q - 1)
    return ALPHA * n ** 3 + BETA * float(M) * M * N + GAMMA * float(N) * N * M


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
