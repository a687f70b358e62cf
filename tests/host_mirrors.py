"""Pure-Python / numpy statements of the host-side rules that libgdca.so implements natively (FASTA reader,
duplicate removal, ranking, synthetic-family generator).  TEST INFRASTRUCTURE: the tests cross-check the native
functions against these; the product package does not contain or import them."""
from __future__ import annotations

import gzip
from typing import List, Tuple

import numpy as np

_LETTERS = "ACDEFGHIKLMNPQRSTVWY"
_L2N = np.full(256, 21, dtype=np.int8)
for _i, _c in enumerate(_LETTERS):
    _L2N[ord(_c)] = _i + 1


def _zf(Z) -> np.ndarray:
    return np.asfortranarray(np.asarray(Z), dtype=np.int8)


def read_fasta_alignment_py(filename: str, max_gap_fraction: float) -> np.ndarray:
    """Pure-Python form of read_fasta_alignment (kept as the readable statement of the rules and as a
    cross-check of the native reader in the tests).  FASTA (plain or .gz) -> Z::Matrix{Int8}, shape (N, M).  Columns kept = positions of the
    first record that are neither '.' nor lowercase; sequences with more than
    ``max_gap_fraction`` gaps ('-') are dropped; ACDEFGHIKLMNPQRSTVWY -> 1..20, else 21."""
    opener = gzip.open if str(filename).endswith(".gz") else open
    seqs: List[str] = []
    cur = None
    with opener(filename, "rt") as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            if line[0] == ">":
                if cur is not None:
                    seqs.append("".join(cur))
                cur = []
            elif cur is not None:
                cur.append(line)
    if cur is not None:
        seqs.append("".join(cur))
    if not seqs:
        raise ValueError("empty alignment")
    first = seqs[0]
    cols = np.asarray([p for p, c in enumerate(first) if c != "." and not c.islower()], dtype=np.int64)
    N = cols.size
    kept = []
    for sq in seqs:
        b = np.frombuffer(sq.encode("ascii"), dtype=np.uint8)
        if b.size != len(first):
            raise ValueError("inputs are not aligned")
        match = np.flatnonzero((b != ord(".")) & ~((b >= ord("a")) & (b <= ord("z"))))
        if match.size != N or not np.array_equal(match, cols):
            raise ValueError("inconsistent inputs")
        b = b[cols]
        if np.count_nonzero(b == ord("-")) / N <= max_gap_fraction:
            kept.append(_L2N[b])
    return np.asfortranarray(np.stack(kept, axis=1).astype(np.int8))


def remove_duplicate_sequences_py(Z) -> Tuple[np.ndarray, np.ndarray]:
    """-> (Z without repeated columns, 1-based indices kept); first occurrences, order kept
    (reference call site src/GaussDCA.jl:21-23)."""
    Zf = _zf(Z)
    seen = set()
    keep = []
    for k in range(Zf.shape[1]):
        key = Zf[:, k].tobytes()
        if key not in seen:
            seen.add(key)
            keep.append(k)
    keep_a = np.asarray(keep, dtype=np.int64)
    return np.asfortranarray(Zf[:, keep_a]), keep_a + 1


def compute_ranking_py(S, min_separation: int = 5) -> List[Tuple[int, int, float]]:
    """[(i, j, S[j, i])] for 1 <= i, j = i + min_separation .. N, sorted by score descending with
    a stable sort (exact ties keep generation order, as Julia's default sort! does)."""
    S = np.asarray(S)
    N = S.shape[0]
    m = int(min_separation)
    ii, jj = [], []
    for i in range(N - m):
        js = np.arange(i + m, N)
        ii.append(np.full(js.size, i, dtype=np.int64))
        jj.append(js)
    if not ii:
        return []
    ii = np.concatenate(ii)
    jj = np.concatenate(jj)
    sc = S[jj, ii]
    # Julia's isless, reversed: NaN first, then descending, 0.0 before -0.0; stable
    nan = np.isnan(sc)
    negzero = (sc == 0) & np.signbit(sc)
    order = np.lexsort((negzero, np.where(nan, 0.0, -sc), ~nan))  # last key is the primary one; lexsort is stable
    return [(int(ii[t]) + 1, int(jj[t]) + 1, float(sc[t])) for t in order]


# ---- synthetic families: numpy statement of gdca_synth_family (csrc/gdca_host.cpp) -------------------------
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_MU = np.array([85899345, 214748364, 429496729, 858993459, 1288490188, 2147483648], dtype=np.uint64)  # mu * 2^32


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
