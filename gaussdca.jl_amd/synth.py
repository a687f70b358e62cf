"""Deterministic synthetic protein families for benchmarks and size-independent parity tests
(SURVEY.md 8d): `synth_family` calls the native generator in libgdca.so (gdca_synth_family), `synth_family_py`
is the same SplitMix64 recipe in numpy (kept for cross-checks), `write_fasta` emits the family as a FASTA file
a reference installation could read (gDCA only accepts a filename, /root/reference/src/GaussDCA.jl:8-9)."""
from __future__ import annotations

import numpy as np

from . import _lib

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_MU = np.array([85899345, 214748364, 429496729, 858993459, 1288490188, 2147483648], dtype=np.uint64)  # mu * 2^32
SEEDS = {"B": 0xB128, "C": 0xC500, "D": 0xD1000, "E": 0xE000}


def _mix(z):
    z = np.asarray(z, dtype=np.uint64)
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def _stream_start(seed, tag, idx):
    """State after the seeding draw of stream (tag, idx): s0 = mix(x + gold) with x = seed ^ tag<<56 ^ idx."""
    with np.errstate(over="ignore"):
        x = np.uint64(seed) ^ (np.uint64(tag) << np.uint64(56)) ^ np.asarray(idx, dtype=np.uint64)
        return _mix(x + _GOLD)


def _draws(s0, first, count):
    """Draws number first .. first+count-1 (0-based) of the streams with start states s0: shape s0.shape + (count,)."""
    with np.errstate(over="ignore"):
        j = (np.arange(first + 1, first + 1 + count, dtype=np.uint64) * _GOLD)
        return _mix(np.asarray(s0, dtype=np.uint64)[..., None] + j)


def _below(r, n):
    return ((r >> np.uint64(32)) * np.uint64(n)) >> np.uint64(32)


def _resample(r, thresh, nsym, keep):
    new = (1 + (((r & np.uint64(0xFFFFFFFF)) * np.uint64(nsym)) >> np.uint64(32))).astype(np.int8)
    return np.where((r >> np.uint64(32)) < thresh, new, keep)


def synth_family_py(N: int, M: int, q: int = 21, seed: int = 0xC500) -> np.ndarray:
    """numpy statement of gdca_synth_family (csrc/gdca_host.cpp); returns (M, N) int8."""
    nsym = q - 1
    root = (1 + _below(_draws(_stream_start(seed, 0, 0), 0, N), nsym)).astype(np.int8)
    K = (M + 24) // 25
    centres = _resample(_draws(_stream_start(seed, 1, np.arange(K)), 0, N), np.uint64(1 << 30), nsym, root[None, :])
    s0 = _stream_start(seed, 2, np.arange(M))
    head = _draws(s0, 0, 2)
    cen = _below(head[:, 0], K).astype(np.int64)
    thr = _MU[_below(head[:, 1], 6).astype(np.int64)]
    Z = _resample(_draws(s0, 2, N), thr[:, None], nsym, centres[cen]).astype(np.int8)
    tail = _draws(s0, 2 + N, 7)
    nruns = _below(tail[:, 0], 4).astype(np.int64)
    maxlen = max(2, N // 10)
    for r in range(3):
        sel = np.nonzero(nruns > r)[0]
        a = _below(tail[sel, 1 + 2 * r], N).astype(np.int64)
        ln = 1 + _below(tail[sel, 2 + 2 * r], maxlen).astype(np.int64)
        for k, a0, l0 in zip(sel, a, ln):
            Z[k, a0:a0 + l0] = q
    return np.ascontiguousarray(Z)


def synth_family(N: int, M: int, q: int = 21, seed: int = 0xC500) -> np.ndarray:
    """Native generator (host code in libgdca.so; no GPU needed); returns (M, N) int8 == the N x M column-major Z."""
    Z = np.empty((M, N), dtype=np.int8)
    if _lib.load().gdca_synth_family(N, M, q, seed, _lib._p(Z)) != 0:
        raise _lib.ArgumentError(f"gdca_synth_family: invalid arguments N={N} M={M} q={q}")
    return Z


def write_fasta(path: str, Z: np.ndarray) -> None:
    """Z as (M, N) int8 -> FASTA (gzip when the name ends in .gz)."""
    Z = np.ascontiguousarray(Z, dtype=np.int8)
    if _lib.load().gdca_write_fasta(str(path).encode(), _lib._p(Z), Z.shape[1], Z.shape[0]) != 0:
        raise _lib.ArgumentError(f"gdca_write_fasta: cannot write {path} (or symbols outside 1..21)")
