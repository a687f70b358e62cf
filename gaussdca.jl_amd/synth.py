"""Deterministic synthetic protein families for benchmarks and size-independent parity tests
(SURVEY.md 8d): `synth_family` calls the native generator in libgdca.so (gdca_synth_family; the same SplitMix64 recipe in
numpy is kept under tests/host_mirrors.py for cross-checks), `write_fasta` emits the family as a FASTA file
a reference installation could read (gDCA only accepts a filename, /root/reference/src/GaussDCA.jl:8-9)."""
from __future__ import annotations

import numpy as np

from . import _lib

SEEDS = {"B": 0xB128, "C": 0xC500, "D": 0xD1000, "E": 0xE000}


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
