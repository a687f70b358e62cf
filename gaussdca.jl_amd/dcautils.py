"""Host-side mirror of the DCAUtils.jl operator surface that GaussDCA.jl calls
(/root/reference/src/GaussDCA.jl:20,22,28,30,37,39) plus the in-tree helpers compute_C (:76),
correct_APC (:78-86) and compute_ranking (:88-99).  Same names, argument meaning, return
shapes and error behaviour; the arithmetic of every hot operator runs in libgdca.so on the
MI355X (no CPU fallback).  FASTA parsing, deduplication and the ranking sort are host code,
as they are in the reference.

Array conventions follow Julia: ``Z`` has shape ``(N, M)`` (column = sequence) and is passed
Fortran-contiguous, i.e. the same bytes as Julia's ``Matrix{Int8}``; indices in rankings are
1-based.
"""
from __future__ import annotations

import ctypes as C
import gzip
import math
from collections.abc import Sequence as _SequenceABC
from typing import List, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import ArgumentError, PosDefException, default_context  # noqa: F401

_LETTERS = "ACDEFGHIKLMNPQRSTVWY"
_L2N = np.full(256, 21, dtype=np.int8)
for _i, _c in enumerate(_LETTERS):
    _L2N[ord(_c)] = _i + 1


def _zf(Z) -> np.ndarray:
    Z = np.asarray(Z)
    if Z.ndim != 2:
        raise ArgumentError("Z must be an N x M matrix")
    return np.asfortranarray(Z, dtype=np.int8)


def _theta_arg(theta) -> float:
    """theta = :auto | Real in [0, 1]  ->  the C-ABI's encoding (negative = auto)."""
    if isinstance(theta, str):
        if theta.lstrip(":") == "auto":
            return -1.0
        raise ArgumentError(f"invalid θ value: {theta} (must be either :auto, or a number between 0 and 1)")
    if isinstance(theta, (int, float, np.integer, np.floating)) and 0 <= theta <= 1:
        return float(theta)
    raise ArgumentError(f"invalid θ value: {theta} (must be either :auto, or a number between 0 and 1)")


# ---- host I/O (reference: DCAUtils.read_fasta_alignment, src/GaussDCA.jl:20) -----------------------
# ---- native host utilities (libgdca.so, plain C++; pure-Python statements of the same rules: tests/host_mirrors.py) ----
def read_fasta_alignment(filename: str, max_gap_fraction: float) -> np.ndarray:
    """read_fasta_alignment(filename, max_gap_fraction) -> Z::Matrix{Int8}, shape (N, M), Fortran order
    (reference call site src/GaussDCA.jl:20)."""
    lib = _lib.load()
    h = C.c_void_p()
    N, M = C.c_int32(), C.c_int32()
    st = lib.gdca_fasta_open(str(filename).encode(), float(max_gap_fraction), C.byref(h), C.byref(N), C.byref(M))
    if st != 0:
        raise ValueError(f"cannot read FASTA alignment {filename} (empty, unreadable or not aligned)")
    try:
        Z = np.empty((N.value, M.value), dtype=np.int8, order="F")
        lib.gdca_fasta_copy(h, _lib._p(Z))
    finally:
        lib.gdca_fasta_close(h)
    return Z


class FastaAlignment:
    """A parsed alignment kept where the native reader put it (gdca_fasta_open): `.ptr` = the N x M int8 matrix
    (gdca_fasta_data), `.N`, `.M`, `.q` = maximum(Z) (gdca_fasta_max_symbol).  A context manager: the matrix is freed on exit."""

    def __init__(self, filename: str, max_gap_fraction: float):
        self.lib = _lib.load()
        self.h = C.c_void_p()
        N, M = C.c_int32(), C.c_int32()
        st = self.lib.gdca_fasta_open(str(filename).encode(), float(max_gap_fraction), C.byref(self.h), C.byref(N), C.byref(M))
        if st != 0:
            self.h = None
            raise ValueError(f"cannot read FASTA alignment {filename} (empty, unreadable or not aligned)")
        self.N, self.M = int(N.value), int(M.value)
        self.ptr = int(self.lib.gdca_fasta_data(self.h) or 0)
        self.q = int(self.lib.gdca_fasta_max_symbol(self.h))

    def close(self):
        if self.h:
            self.lib.gdca_fasta_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def remove_duplicate_sequences(Z) -> Tuple[np.ndarray, np.ndarray]:
    """-> (Z without repeated columns, 1-based indices kept)  (reference call site src/GaussDCA.jl:21-23)"""
    lib = _lib.load()
    Zf = _zf(Z)
    N, M = Zf.shape
    out = np.empty((N, M), dtype=np.int8, order="F")
    idx = np.empty(M, dtype=np.int32)
    m = C.c_int32()
    st = lib.gdca_remove_duplicates(_lib._p(Zf), N, M, _lib._p(out), _lib._p(idx), C.byref(m))
    if st != 0:
        raise ArgumentError("remove_duplicate_sequences: invalid arguments")
    return np.asfortranarray(out[:, :m.value]), idx[:m.value].astype(np.int64)


class Ranking(_SequenceABC):
    """The ranking as a sequence of (i, j, score) tuples (what the reference returns as a
    Vector{Tuple{Int,Int,Float64}}), backed by the three arrays the native sort filled: building
    122 760 Python tuples costs more than the sort, so they are made on access.  `.i`, `.j`, `.score`
    are the arrays; slicing gives a list of tuples; comparing with a list compares element-wise."""

    __slots__ = ("i", "j", "score")

    def __init__(self, i, j, score):
        self.i, self.j, self.score = i, j, score

    def __len__(self):
        return int(self.i.shape[0])

    def __getitem__(self, t):
        if isinstance(t, slice):
            return list(zip(self.i[t].tolist(), self.j[t].tolist(), self.score[t].tolist()))
        return (int(self.i[t]), int(self.j[t]), float(self.score[t]))

    def __iter__(self):
        return iter(zip(self.i.tolist(), self.j.tolist(), self.score.tolist()))

    def __eq__(self, other):
        try:
            return len(self) == len(other) and all(a == tuple(b) for a, b in zip(self, other))
        except TypeError:
            return NotImplemented

    def __repr__(self):
        head = ", ".join(repr(x) for x in self[:3])
        return f"Ranking(len={len(self)}: [{head}{', ...' if len(self) > 3 else ''}])"


def compute_ranking(S, min_separation: int = 5) -> "Ranking":
    """compute_ranking(S, min_separation)  (src/GaussDCA.jl:88-99): 1-based (i, j, S[j, i]), stable sort by
    `isless` on the score, reversed (NaN first, 0.0 before -0.0, exact ties in generation order)."""
    lib = _lib.load()
    Sf = np.asfortranarray(S, dtype=np.float64)
    N = Sf.shape[0]
    n = int(lib.gdca_ranking_length(N, int(min_separation)))
    if n <= 0:
        return Ranking(np.empty(0, np.int32), np.empty(0, np.int32), np.empty(0, np.float64))
    ii = np.empty(n, dtype=np.int32)
    jj = np.empty(n, dtype=np.int32)
    sc = np.empty(n, dtype=np.float64)
    st = lib.gdca_ranking(_lib._p(Sf), N, int(min_separation), _lib._p(ii), _lib._p(jj), _lib._p(sc))
    if st != 0:
        raise ArgumentError("compute_ranking: invalid arguments")
    return Ranking(ii, jj, sc)


# ---- hot operators (libgdca.so) -----------------------------------------------------------------------
def compute_theta(Z, ctx=None) -> float:
    ctx = ctx or default_context()
    Zf = _zf(Z)
    N, M = Zf.shape
    th = C.c_double()
    ctx.check(ctx.lib.gdca_compute_theta(ctx.h, _lib._p(Zf), N, M, C.byref(th)))
    return th.value


def pair_identity_sum(Z, ctx=None) -> int:
    ctx = ctx or default_context()
    Zf = _zf(Z)
    N, M = Zf.shape
    out = C.c_uint64()
    ctx.check(ctx.lib.gdca_pair_identity_sum(ctx.h, _lib._p(Zf), N, M, C.byref(out)))
    return int(out.value)


def neighbour_counts(Z, thresh: int, ctx=None) -> np.ndarray:
    """n_k = 1 + #{l != k : Hamming(Z[:,k], Z[:,l]) < thresh}, int32[M] (bit-exact integers)."""
    ctx = ctx or default_context()
    Zf = _zf(Z)
    N, M = Zf.shape
    n = np.empty(M, dtype=np.int32)
    ctx.check(ctx.lib.gdca_neighbour_counts(ctx.h, _lib._p(Zf), N, M, int(thresh), _lib._p(n)))
    return n


def compute_weights(Z, q=None, theta=":auto", ctx=None, return_theta: bool = False):
    """compute_weights(Z, [q,] θ) -> (W, Meff).  ``q`` is accepted for signature compatibility
    (DCAUtils uses it to pick its bit-packed path) and checked against the q <= 31 limit."""
    if q is not None and not isinstance(q, (int, np.integer)):
        q, theta = None, q  # called as compute_weights(Z, θ)
    if q is not None and q >= 32:
        raise ArgumentError(f"parameter q={q} is too big (max 31 is allowed)")
    ctx = ctx or default_context()
    Zf = _zf(Z)
    N, M = Zf.shape
    W = np.empty(M, dtype=np.float64)
    Meff, th, thr = C.c_double(), C.c_double(), C.c_int32()
    ctx.check(ctx.lib.gdca_compute_weights(ctx.h, _lib._p(Zf), N, M, _theta_arg(theta), _lib._p(W), C.byref(Meff),
                                           C.byref(th), C.byref(thr)))
    if return_theta:
        return W, Meff.value, th.value, int(thr.value)
    return W, Meff.value


def compute_weighted_frequencies(Z, q_or_W, theta_or_Meff=":auto", ctx=None):
    """compute_weighted_frequencies(Z, q, θ) -> (Pi_true, Pij_true, Meff, W)   (src/GaussDCA.jl:28)
    compute_weighted_frequencies(Z, W, Meff) -> (Pi_true, Pij_true)           (DCAUtils' 2nd method)"""
    ctx = ctx or default_context()
    Zf = _zf(Z)
    N, M = Zf.shape
    if isinstance(q_or_W, (int, np.integer)):
        q = int(q_or_W)
        if q >= 32:
            raise ArgumentError(f"parameter q={q} is too big (max 31 is allowed)")
        W, Meff = compute_weights(Zf, q, theta_or_Meff, ctx=ctx)
        Pi, Pij = _frequencies(ctx, Zf, q, W, Meff)
        return Pi, Pij, Meff, W
    W = np.ascontiguousarray(q_or_W, dtype=np.float64)
    q = int(Zf.max())
    return _frequencies(ctx, Zf, q, W, float(theta_or_Meff))


def _frequencies(ctx, Zf, q, W, Meff):
    N, M = Zf.shape
    n = N * (q - 1)
    Pi = np.empty(n, dtype=np.float64)
    Pij = np.empty((n, n), dtype=np.float64)
    ctx.check(ctx.lib.gdca_frequencies(ctx.h, _lib._p(Zf), N, M, q, _lib._p(W), float(Meff), _lib._p(Pi),
                                       _lib._p(Pij)))
    return Pi, Pij


def add_pseudocount(Pi_true, Pij_true, pc: float, q: int = 21, ctx=None):
    """add_pseudocount(Pi_true, Pij_true, pc, q) -> (Pi, Pij)   (src/GaussDCA.jl:30)"""
    ctx = ctx or default_context()
    Pi_true = np.ascontiguousarray(Pi_true, dtype=np.float64)
    Pij_true = np.ascontiguousarray(Pij_true, dtype=np.float64)
    n = Pi_true.shape[0]
    if Pij_true.shape != (n, n) or n % (q - 1) != 0:
        raise ArgumentError("incompatible sizes of Pi, Pij and q")
    N = n // (q - 1)
    Pi = np.empty_like(Pi_true)
    Pij = np.empty_like(Pij_true)
    ctx.check(ctx.lib.gdca_add_pseudocount(ctx.h, _lib._p(Pi_true), _lib._p(Pij_true), N, int(q), float(pc),
                                           _lib._p(Pi), _lib._p(Pij)))
    return Pi, Pij


def compute_C(Pi, Pij, ctx=None) -> np.ndarray:
    """compute_C(Pi, Pij) = Pij - Pi * Pi'   (src/GaussDCA.jl:76)"""
    ctx = ctx or default_context()
    Pi = np.ascontiguousarray(Pi, dtype=np.float64)
    Pij = np.ascontiguousarray(Pij, dtype=np.float64)
    n = Pi.shape[0]
    Cm = np.empty((n, n), dtype=np.float64)
    ctx.check(ctx.lib.gdca_covariance(ctx.h, _lib._p(Pi), _lib._p(Pij), n, _lib._p(Cm)))
    return Cm


def inv_cholesky(Cm, ctx=None) -> np.ndarray:
    """mJ = inv(cholesky(C))   (src/GaussDCA.jl:34).  Raises PosDefException(info) like Julia."""
    ctx = ctx or default_context()
    A = np.array(Cm, dtype=np.float64, order="C", copy=True)
    n = A.shape[0]
    if A.shape != (n, n) or not np.array_equal(A, A.T):
        raise PosDefException(-1)
    info = C.c_int32()
    rc = ctx.lib.gdca_spd_inverse(ctx.h, _lib._p(A), n, C.byref(info))
    ctx.check(rc, info.value)
    return A


def compute_FN(mJ, q: int = 21, ctx=None) -> np.ndarray:
    """compute_FN(mJ, q) -> N x N   (src/GaussDCA.jl:39)"""
    ctx = ctx or default_context()
    mJ = np.ascontiguousarray(mJ, dtype=np.float64)
    N = mJ.shape[0] // (q - 1)
    S = np.empty((N, N), dtype=np.float64)
    ctx.check(ctx.lib.gdca_fn(ctx.h, _lib._p(mJ), N, int(q), _lib._p(S)))
    return S


def compute_DI_gauss(mJ, Cm, q: int = 21, ctx=None) -> np.ndarray:
    """compute_DI_gauss(mJ, C, q) -> N x N   (src/GaussDCA.jl:37)"""
    ctx = ctx or default_context()
    mJ = np.ascontiguousarray(mJ, dtype=np.float64)
    Cm = np.ascontiguousarray(Cm, dtype=np.float64)
    N = mJ.shape[0] // (q - 1)
    S = np.empty((N, N), dtype=np.float64)
    ctx.check(ctx.lib.gdca_di(ctx.h, _lib._p(mJ), _lib._p(Cm), N, int(q), _lib._p(S)))
    return S


def correct_APC(S, ctx=None) -> np.ndarray:
    """correct_APC(S)   (src/GaussDCA.jl:78-86)"""
    ctx = ctx or default_context()
    A = np.array(S, dtype=np.float64, order="C", copy=True)
    ctx.check(ctx.lib.gdca_apc(ctx.h, _lib._p(A), A.shape[0]))
    return A


def printrank(io, R: Sequence[Tuple[int, int, float]] = None):
    """printrank(io, R) / printrank(filename, R): one "%i %i %e" line per entry
    (src/GaussDCA.jl:67-74).  printrank(R) alone writes to stdout (the reference's one-argument
    method references the undefined STDOUT and throws; here it works)."""
    import sys

    if R is None:
        io, R = sys.stdout, io
    if isinstance(io, (str, bytes)):
        lib = _lib.load()
        if isinstance(R, Ranking):
            ii, jj, sc = (np.ascontiguousarray(R.i, dtype=np.int32), np.ascontiguousarray(R.j, dtype=np.int32),
                          np.ascontiguousarray(R.score, dtype=np.float64))
        else:
            ii = np.asarray([r[0] for r in R], dtype=np.int32)
            jj = np.asarray([r[1] for r in R], dtype=np.int32)
            sc = np.asarray([r[2] for r in R], dtype=np.float64)
        path = io if isinstance(io, bytes) else io.encode()
        if lib.gdca_write_rank(path, _lib._p(ii), _lib._p(jj), _lib._p(sc), len(R)) != 0:
            raise OSError(f"cannot write {io}")
        return None
    for (i, j, x) in R:
        io.write("%i %i %e\n" % (i, j, x))
