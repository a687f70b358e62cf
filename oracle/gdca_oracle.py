"""CPU oracle for the gDCA hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``gaussdca.jl_amd``) never does, and it fails loudly
when the HIP library is missing.

What this is
------------
A numpy/scipy restatement of the algorithm behind ``gDCA()`` in the reference
(``/root/reference/src/GaussDCA.jl:8-47``).  Only three helpers of the hot path live in the
reference tree (``compute_C`` ``:76``, ``correct_APC`` ``:78-86``, ``compute_ranking``
``:88-99``); the rest is delegated to the third-party Julia package **DCAUtils.jl** (uuid
``e41cd558-3099-4f6e-a65d-5336857e40aa``, compat ``"1"``, ``Project.toml:6,12``; no
Manifest, so any 1.x) and to Julia's ``LinearAlgebra`` (``inv(cholesky(C))`` at
``src/GaussDCA.jl:34`` = LAPACK ``dpotrf('U')`` + ``dpotri`` + mirror).  Neither Julia nor
DCAUtils exists in this image, so the DCAUtils functions are restated from their published
algorithm (rules 1-13) and every function below cites the reference
call site it stands behind.

Parity pinning
--------------
PINNED against the reference's own golden vectors: ``tests/test_oracle_golden.py`` runs this
oracle end to end on the reference's two FASTA inputs and checks all 82,787 rows of the four
golden ranking files held by the reference's test-suite (``test/runtests.jl:52-86``,
``test/data/*.txt``; copied verbatim to ``tests/golden/reference/``): identical (i, j) key
sets and every score within 1e-6 relative (the goldens are printed with ``%e`` = 7
significant digits).  The intermediates theta / threshold / Meff are parity-UNPINNED by any
reference test at the last-ulp level (DCAUtils' floating-point evaluation order is not in
the tree); they are pinned only through the O(1) sensitivity of the golden scores to them.

Conventions: 0-based indices internally; ``Z`` is an ``int8`` array of shape ``(M, N)``
C-contiguous, i.e. the *same bytes* as Julia's column-major ``N x M`` ``Matrix{Int8}``
(one sequence = one contiguous run of N bytes).  ``s = q - 1``; ``n = N * s``.
"""
from __future__ import annotations

import ctypes
import gzip
import math
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# --------------------------------------------------------------------------------------
# optional C accelerators (oracle/oracle_kernels.c, built by __graft_entry__.build()).
# They restate the *same* loops in the *same* order; tests check them against the numpy
# forms below.  Without them everything still works, only slower.
# --------------------------------------------------------------------------------------
_ck = None


def _load_ck():
    global _ck
    if _ck is not None:
        return _ck
    path = os.path.join(_HERE, "_build", "liboracle_kernels.so")
    if os.path.exists(path) and os.environ.get("GDCA_ORACLE_NO_C") != "1":
        lib = ctypes.CDLL(path)
        i8p = ctypes.POINTER(ctypes.c_int8)
        i32p = ctypes.POINTER(ctypes.c_int32)
        f64p = ctypes.POINTER(ctypes.c_double)
        lib.orc_neighbour_counts.argtypes = [i8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, i32p]
        lib.orc_neighbour_counts.restype = None
        lib.orc_pair_identity_sum.argtypes = [i8p, ctypes.c_int, ctypes.c_int]
        lib.orc_pair_identity_sum.restype = ctypes.c_uint64
        lib.orc_frequencies.argtypes = [i8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, f64p,
                                        ctypes.c_double, f64p, f64p]
        lib.orc_frequencies.restype = None
        lib.orc_fn.argtypes = [f64p, ctypes.c_int, ctypes.c_int, f64p]
        lib.orc_fn.restype = None
        lib.orc_set_threads.argtypes = [ctypes.c_int]
        lib.orc_set_threads.restype = ctypes.c_int
        _ck = lib
    else:
        _ck = False
    return _ck


def have_c_kernels() -> bool:
    return bool(_load_ck())


def set_threads(n: int) -> int:
    """Threads used by the C accelerators (OpenMP); returns the count in effect."""
    ck = _load_ck()
    return ck.orc_set_threads(int(n)) if ck else 1


def _ptr(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


# --------------------------------------------------------------------------------------
# Rule 1 -- read_fasta_alignment (DCAUtils; call site src/GaussDCA.jl:20)
# --------------------------------------------------------------------------------------
_LETTERS = "ACDEFGHIKLMNPQRSTVWY"
_LETTER2NUM = np.full(256, 21, dtype=np.int8)
for _i, _c in enumerate(_LETTERS):
    _LETTER2NUM[ord(_c)] = _i + 1


def _parse_fasta(text: str) -> List[str]:
    seqs: List[str] = []
    cur: Optional[List[str]] = None
    for line in text.splitlines():
        line = line.strip()
        if not line:
            continue
        if line.startswith(">"):
            if cur is not None:
                seqs.append("".join(cur))
            cur = []
        elif cur is not None:
            cur.append(line)
    if cur is not None:
        seqs.append("".join(cur))
    return seqs


def read_fasta_alignment(filename: str, max_gap_fraction: float) -> np.ndarray:
    """FASTA(.gz) -> Z int8 (M, N).  SURVEY 4.3 rule 1; call site src/GaussDCA.jl:20.

    Kept columns = positions of the first record that are neither '.' nor lowercase; a
    sequence is kept iff (#'-' among kept columns) / N <= max_gap_fraction; letters
    ACDEFGHIKLMNPQRSTVWY -> 1..20, everything else -> 21.
    """
    opener = gzip.open if filename.endswith(".gz") else open
    with opener(filename, "rt") as f:
        seqs = _parse_fasta(f.read())
    if not seqs:
        raise ValueError("empty alignment")
    first = seqs[0]
    cols = [p for p, c in enumerate(first) if c != "." and not c.islower()]
    N = len(cols)
    cols_a = np.asarray(cols, dtype=np.int64)
    rows = []
    for sq in seqs:
        b = np.frombuffer(sq.encode("ascii"), dtype=np.uint8)
        if b.size != len(first):
            raise ValueError("inputs are not aligned")
        match = np.flatnonzero((b != ord(".")) & ~((b >= ord("a")) & (b <= ord("z"))))
        if match.size != N or not np.array_equal(match, cols_a):
            raise ValueError("inconsistent inputs")  # every record has the first record's match columns
        kept = b[cols_a]
        ngaps = int(np.count_nonzero(kept == ord("-")))
        if ngaps / N <= max_gap_fraction:
            rows.append(_LETTER2NUM[kept])
    return np.ascontiguousarray(np.stack(rows, axis=0).astype(np.int8))


def remove_duplicate_sequences(Z: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Keep the first occurrence of every distinct sequence, order preserved
    (DCAUtils; call site src/GaussDCA.jl:21-23)."""
    seen = {}
    keep = []
    for k in range(Z.shape[0]):
        key = Z[k].tobytes()
        if key not in seen:
            seen[key] = k
            keep.append(k)
    keep_a = np.asarray(keep, dtype=np.int64)
    return np.ascontiguousarray(Z[keep_a]), keep_a


# --------------------------------------------------------------------------------------
# Rule 3 -- compute_theta (inside compute_weighted_frequencies, src/GaussDCA.jl:28)
# --------------------------------------------------------------------------------------
def pair_identity_sum(Z: np.ndarray) -> int:
    """sum_{k<l} #{i : Z[k,i] == Z[l,i]} (gap==gap counts), exact integer.

    The reference does an all-pairs pass; the value has the closed form
    sum_i sum_a c_ia (c_ia - 1) / 2 with c = per-column symbol counts."""
    M, N = Z.shape
    tot = 0
    for a in range(1, 33):
        c = np.count_nonzero(Z == a, axis=0).astype(object)
        tot += int(np.sum(c * (c - 1) // 2))
    return int(tot)


def pair_identity_sum_allpairs(Z: np.ndarray) -> int:
    """The literal all-pairs form (small inputs / C accelerator); must equal the closed form."""
    ck = _load_ck()
    M, N = Z.shape
    Zc = np.ascontiguousarray(Z, dtype=np.int8)
    if ck:
        return int(ck.orc_pair_identity_sum(_ptr(Zc, ctypes.c_int8), N, M))
    tot = 0
    for k in range(M - 1):
        tot += int(np.count_nonzero(Zc[k + 1:] == Zc[k]))
    return tot


def compute_theta(Z: np.ndarray) -> float:
    """theta = min(0.5, 0.38*0.32 / mean pair identity).  SURVEY 4.3 rule 3."""
    M, N = Z.shape
    if M < 2:
        return 0.0
    phi = pair_identity_sum(Z) / (N * (0.5 * M * (M - 1)))
    if phi == 0.0:
        return 0.5  # Julia: 0.1216 / 0.0 == Inf, min(0.5, Inf) == 0.5
    return min(0.5, 0.38 * 0.32 / phi)


# --------------------------------------------------------------------------------------
# Rule 4 -- compute_weights (same call, src/GaussDCA.jl:28)
# --------------------------------------------------------------------------------------
def hamming_threshold(theta: float, N: int) -> int:
    return int(math.floor(theta * N))


def neighbour_counts(Z: np.ndarray, thresh: int) -> np.ndarray:
    """n_k = 1 + #{l != k : Hamming(k, l) < thresh}, int32[M] (strict '<')."""
    M, N = Z.shape
    Zc = np.ascontiguousarray(Z, dtype=np.int8)
    n = np.ones(M, dtype=np.int32)
    if thresh <= 0 or M < 2:
        return n
    ck = _load_ck()
    if ck:
        ck.orc_neighbour_counts(_ptr(Zc, ctypes.c_int8), N, M, int(thresh), _ptr(n, ctypes.c_int32))
        return n
    for k in range(M - 1):
        d = np.count_nonzero(Zc[k + 1:] != Zc[k], axis=1)
        close = d < thresh
        n[k] += int(np.count_nonzero(close))
        n[k + 1:] += close.astype(np.int32)
    return n


def weights_from_counts(n: np.ndarray) -> Tuple[np.ndarray, float]:
    """W_k = 1/n_k; Meff = sum(W), the exact sum rounded once (math.fsum).  DCAUtils returns ``sum(W)``: Julia's pairwise,
    SIMD-reassociated sum, whose last bit depends on the machine -- no f64 evaluation order pins it (header, "Parity pinning"),
    so the restatement uses the one value that needs no order.  (Rounds 1-4 of this repository summed left to right.)"""
    W = 1.0 / n.astype(np.float64)
    Meff = math.fsum(W.tolist())
    return W, Meff


def meff_three_ways(W: np.ndarray) -> dict:
    """The three candidate values of ``Meff = sum(W)`` (compute_weights, call site src/GaussDCA.jl:28), recorded side by side in
    tests/golden/intermediates.json so that a box WITH Julia can tell which one DCAUtils returns (VERDICT r05 weak #1a):
      exact          the sum in exact arithmetic rounded once (math.fsum) -- what this oracle and the HIP path (k_meff) return;
      left_to_right  ((W[0] + W[1]) + W[2]) + ... in f64 -- rounds 1-4 of this repository;
      pairwise_1024  the recursion of Julia's Base.mapreduce_impl (what ``sum(::Vector{Float64})`` runs): a range of at most 1024
                     elements is summed in a loop, a longer one is split at ``mid = (first + last) >> 1`` and the halves' sums are added.
                     Julia's base-case loop is ``@simd`` -- the compiler may keep several partial sums, how many depends on the
                     machine's vector width -- so even this form pins the reference's last bit only up to that; the base case
                     here is the plain left-to-right loop."""
    w = [float(x) for x in np.asarray(W, dtype=np.float64)]
    seq = 0.0
    for x in w:
        seq += x

    def pw(first, last):  # inclusive, as Base.mapreduce_impl
        if first == last:
            return w[first]
        if first + 1024 > last:
            v = w[first] + w[first + 1]
            for i in range(first + 2, last + 1):
                v += w[i]
            return v
        mid = (first + last) >> 1
        return pw(first, mid) + pw(mid + 1, last)

    return {"exact": math.fsum(w), "left_to_right": seq, "pairwise_1024": pw(0, len(w) - 1) if w else 0.0}


def compute_weights(Z: np.ndarray, theta) -> Tuple[np.ndarray, float, float, int]:
    """-> (W, Meff, theta_used, thresh).  theta may be 'auto' or a real in [0, 1]."""
    M, N = Z.shape
    th = compute_theta(Z) if (isinstance(theta, str) and theta == "auto") else float(theta)
    if th == 0.0:
        return np.ones(M), float(M), th, 0
    thresh = hamming_threshold(th, N)
    W, Meff = weights_from_counts(neighbour_counts(Z, thresh))
    return W, Meff, th, thresh


# --------------------------------------------------------------------------------------
# Rule 5 -- weighted frequencies (same call, src/GaussDCA.jl:28)
# --------------------------------------------------------------------------------------
def compute_frequencies(Z: np.ndarray, q: int, W: np.ndarray, Meff: float):
    """Pi[n], Pij[n, n] (symmetric).  State q (the gap) has no row/column.
    Accumulation is sequential over sequences k, then one division by Meff."""
    M, N = Z.shape
    s = q - 1
    n = N * s
    Zc = np.ascontiguousarray(Z, dtype=np.int8)
    Wc = np.ascontiguousarray(W, dtype=np.float64)
    ck = _load_ck()
    if ck:
        Pi = np.zeros(n)
        Pij = np.zeros((n, n))
        ck.orc_frequencies(_ptr(Zc, ctypes.c_int8), N, M, q, _ptr(Wc, ctypes.c_double),
                           float(Meff), _ptr(Pi, ctypes.c_double), _ptr(Pij, ctypes.c_double))
        return Pi, Pij
    X = np.zeros((M, n))
    kk, ii = np.nonzero(Zc < q)
    X[kk, ii * s + (Zc[kk, ii].astype(np.int64) - 1)] = 1.0
    Pi = (Wc @ X) / Meff
    Pij = (X.T @ (X * Wc[:, None])) / Meff
    Pij = 0.5 * (Pij + Pij.T)
    return Pi, Pij


def compute_weighted_frequencies(Z: np.ndarray, q: int, theta):
    """DCAUtils.compute_weighted_frequencies(Z, q, theta) -> (Pi_true, Pij_true, Meff, W)
    (call site src/GaussDCA.jl:28)."""
    W, Meff, _, _ = compute_weights(Z, theta)
    Pi, Pij = compute_frequencies(Z, q, W, Meff)
    return Pi, Pij, Meff, W


# --------------------------------------------------------------------------------------
# Rule 6 -- add_pseudocount (src/GaussDCA.jl:30)
# --------------------------------------------------------------------------------------
def add_pseudocount(Pi_true: np.ndarray, Pij_true: np.ndarray, pc: float, q: int):
    s = q - 1
    n = Pi_true.shape[0]
    N = n // s
    pcq = pc / q
    Pij = (1.0 - pc) * Pij_true + pcq / q
    Pi = (1.0 - pc) * Pi_true + pcq
    for i in range(N):
        sl = slice(i * s, (i + 1) * s)
        blk = (1.0 - pc) * Pij_true[sl, sl]
        blk[np.diag_indices(s)] += pcq
        Pij[sl, sl] = blk
    return Pi, Pij


# --------------------------------------------------------------------------------------
# Rule 7 -- compute_C (src/GaussDCA.jl:76)
# --------------------------------------------------------------------------------------
def compute_C(Pi: np.ndarray, Pij: np.ndarray) -> np.ndarray:
    return Pij - np.outer(Pi, Pi)


# --------------------------------------------------------------------------------------
# Rule 8 -- inv(cholesky(C)) (src/GaussDCA.jl:34): LAPACK dpotrf('U') + dpotri + mirror
# --------------------------------------------------------------------------------------
class NotPositiveDefinite(ArithmeticError):
    def __init__(self, info: int):
        super().__init__(f"matrix is not positive definite; Cholesky factorization failed (info={info})")
        self.info = info


def spd_inverse(C: np.ndarray) -> np.ndarray:
    from scipy.linalg import lapack

    if not np.array_equal(C, C.T):
        raise NotPositiveDefinite(-1)
    U, info = lapack.dpotrf(C, lower=0, clean=1, overwrite_a=0)
    if info != 0:
        raise NotPositiveDefinite(int(info))
    Ui, info = lapack.dpotri(U, lower=0, overwrite_c=1)
    if info != 0:
        raise NotPositiveDefinite(int(info))
    iu = np.triu(Ui)
    return iu + np.triu(Ui, 1).T


# --------------------------------------------------------------------------------------
# Rule 9 -- compute_FN (src/GaussDCA.jl:39)
# --------------------------------------------------------------------------------------
def compute_FN(mJ: np.ndarray, q: int) -> np.ndarray:
    s = q - 1
    n = mJ.shape[0]
    N = n // s
    ck = _load_ck()
    if ck:
        A = np.ascontiguousarray(mJ, dtype=np.float64)
        FN = np.zeros((N, N))
        ck.orc_fn(_ptr(A, ctypes.c_double), N, s, _ptr(FN, ctypes.c_double))
        return FN
    B = mJ.reshape(N, s, N, s).transpose(0, 2, 1, 3)  # [i, j, a, b]
    K = B - B.mean(axis=3, keepdims=True) - B.mean(axis=2, keepdims=True) + B.mean(axis=(2, 3), keepdims=True)
    FN = np.sqrt((K * K).sum(axis=(2, 3)))
    FN = np.triu(FN, 1)
    return FN + FN.T


# --------------------------------------------------------------------------------------
# Rule 10 -- compute_DI_gauss (src/GaussDCA.jl:37)
# --------------------------------------------------------------------------------------
def compute_DI_gauss(mJ: np.ndarray, C: np.ndarray, q: int, chunk: int = 4096) -> np.ndarray:
    """DI[i,j] = z + 0.5 * sum_k log(1 + sqrt(1 + 4 gamma_k)),  z = 0.5 s log 0.5,
    gamma = eigvals(Cii^(1/2) B Cjj B' Cii^(1/2)),  B = mJ block (i, j)."""
    s = q - 1
    n = mJ.shape[0]
    N = n // s
    z = 0.5 * s * math.log(0.5)
    Cd = np.stack([C[i * s:(i + 1) * s, i * s:(i + 1) * s] for i in range(N)])  # (N, s, s)
    w, V = np.linalg.eigh(Cd)
    sqrtC = np.einsum("nab,nb,ncb->nac", V, np.sqrt(np.maximum(w, 0.0)), V)
    B4 = mJ.reshape(N, s, N, s)
    iu, ju = np.triu_indices(N, 1)
    DI = np.zeros((N, N))
    for c0 in range(0, iu.size, chunk):
        ii = iu[c0:c0 + chunk]
        jj = ju[c0:c0 + chunk]
        B = B4[ii, :, jj, :]  # (P, s, s)
        T = np.matmul(sqrtC[ii], B)
        Vm = np.matmul(np.matmul(T, Cd[jj]), np.transpose(T, (0, 2, 1)))
        Vm = 0.5 * (Vm + np.transpose(Vm, (0, 2, 1)))
        g = np.linalg.eigvalsh(Vm)
        di = z + 0.5 * np.sum(np.log(1.0 + np.sqrt(1.0 + 4.0 * np.maximum(g, 0.0))), axis=1)
        DI[ii, jj] = di
        DI[jj, ii] = di
    return DI


# --------------------------------------------------------------------------------------
# Rules 11-13 -- correct_APC / compute_ranking / printrank (src/GaussDCA.jl:78-99, :67-71)
# --------------------------------------------------------------------------------------
def correct_APC(S: np.ndarray) -> np.ndarray:
    N = S.shape[0]
    Si = S.sum(axis=0, keepdims=True)
    Sj = S.sum(axis=1, keepdims=True)
    Sa = S.sum() * (1 - 1 / N)
    return S - (Sj @ Si) / Sa


def compute_ranking(S: np.ndarray, min_separation: int = 5) -> List[Tuple[int, int, float]]:
    """1-based (i, j, S[j, i]) for j >= i + min_separation, stable sort by score, descending."""
    N = S.shape[0]
    R = []
    for i in range(N - min_separation):
        for j in range(i + min_separation, N):
            R.append((i + 1, j + 1, float(S[j, i])))
    # sort!(R, by = x -> x[3], rev = true): stable, ordered by isless reversed -- NaN first, 0.0 before -0.0
    R.sort(key=lambda t: (0, 0.0, 0) if t[2] != t[2] else (1, -t[2], 1 if (t[2] == 0 and math.copysign(1.0, t[2]) < 0) else 0))
    return R


def format_rank(R: Sequence[Tuple[int, int, float]]) -> str:
    return "".join("%i %i %e\n" % (i, j, x) for (i, j, x) in R)


# --------------------------------------------------------------------------------------
# gDCA (src/GaussDCA.jl:8-47)
# --------------------------------------------------------------------------------------
def scores_from_Z(Z: np.ndarray, q: int, pseudocount: float = 0.8, theta="auto", score: str = "frob",
                  return_intermediates: bool = False):
    """The hot path proper: Z -> APC-corrected score matrix S (src/GaussDCA.jl:28-42)."""
    W, Meff, th, thresh = compute_weights(Z, theta)
    Pi_true, Pij_true = compute_frequencies(Z, q, W, Meff)
    Pi, Pij = add_pseudocount(Pi_true, Pij_true, float(pseudocount), q)
    C = compute_C(Pi, Pij)
    mJ = spd_inverse(C)
    if score == "DI":
        S = compute_DI_gauss(mJ, C, q)
    else:
        S = compute_FN(mJ, q)
    S = correct_APC(S)
    if return_intermediates:
        return S, dict(theta=th, thresh=thresh, Meff=Meff, W=W, C=C, mJ=mJ)
    return S


def gDCA(filename: str, pseudocount: float = 0.8, theta="auto", max_gap_fraction: float = 0.9,
         score: str = "frob", min_separation: int = 5, remove_dups: bool = False,
         return_intermediates: bool = False):
    if not (0 <= pseudocount <= 1):
        raise ValueError(f"invalid pseudocount value: {pseudocount} (must be between 0 and 1)")
    if not (theta == "auto" or (isinstance(theta, (int, float)) and 0 <= theta <= 1)):
        raise ValueError(f"invalid theta value: {theta} (must be either 'auto', or a number between 0 and 1)")
    if not (0 <= max_gap_fraction <= 1):
        raise ValueError(f"invalid max_gap_fraction value: {max_gap_fraction} (must be between 0 and 1)")
    if score not in ("DI", "frob"):
        raise ValueError(f"invalid score value: {score} (must be either 'DI' or 'frob')")
    if not (min_separation >= 1):
        raise ValueError(f"invalid min_separation value: {min_separation} (must be >= 1)")
    if not os.path.isfile(filename):
        raise ValueError(f"cannot open file {filename}")
    Z = read_fasta_alignment(filename, max_gap_fraction)
    if remove_dups:
        Z, _ = remove_duplicate_sequences(Z)
    q = int(Z.max())
    if q >= 32:
        raise RuntimeError(f"parameter q={q} is too big (max 31 is allowed)")
    out = scores_from_Z(Z, q, pseudocount, theta, score, return_intermediates)
    if return_intermediates:
        S, inter = out
        inter.update(M=Z.shape[0], N=Z.shape[1], q=q)
        return compute_ranking(S, min_separation), inter
    return compute_ranking(out, min_separation)
