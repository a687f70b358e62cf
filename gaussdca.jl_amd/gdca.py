"""gDCA(): host-side mirror of the reference's only entry point
(/root/reference/src/GaussDCA.jl:8-47), same keyword arguments, defaults, validation order and
messages (:49-65).  Everything between compute_weighted_frequencies (:28) and correct_APC (:42)
and compute_ranking (:44) is one call into libgdca.so (gdca_run_ranked): Z goes to the MI355X once,
the sorted ranking comes back; text output stays on the host as in the reference."""
from __future__ import annotations

import os
from typing import List, Tuple

import numpy as np

from . import _lib
from ._lib import ArgumentError, default_context
from .dcautils import (FastaAlignment, Ranking, _theta_arg, read_fasta_alignment, remove_duplicate_sequences)

last_stats = None  # stats of the most recent gDCA call (theta, threshold, Meff, device timings)


def _score_arg(score) -> int:
    s = str(score).lstrip(":")
    if s == "DI":
        return _lib.SCORE_DI
    if s == "frob":
        return _lib.SCORE_FROB
    raise ArgumentError(f"invalid score value: {score} (must be either :DI or :frob)")


def check_arguments(filename, pseudocount, theta, max_gap_fraction, score, min_separation) -> bool:
    """Same checks, same order, same messages as src/GaussDCA.jl:49-65."""
    if not (0 <= pseudocount <= 1):
        raise ArgumentError(f"invalid pseudocount value: {pseudocount} (must be between 0 and 1)")
    _theta_arg(theta)
    if not (0 <= max_gap_fraction <= 1):
        raise ArgumentError(f"invalid max_gap_fraction value: {max_gap_fraction} (must be between 0 and 1)")
    _score_arg(score)
    if not (min_separation >= 1):
        raise ArgumentError(f"invalid min_separation value: {min_separation} (must be >= 1)")
    if not os.path.isfile(filename):
        raise ArgumentError(f"cannot open file {filename}")
    return True


def gDCA(filename: str, pseudocount: float = 0.8, theta=":auto", max_gap_fraction: float = 0.9,
         score=":frob", min_separation: int = 5, remove_dups: bool = False, ctx=None,
         **kw) -> List[Tuple[int, int, float]]:
    """Gaussian DCA contact ranking of a FASTA alignment: [(i, j, score)], best first.

    Keyword ``θ`` is accepted as an alias of ``theta``; ``score`` may be ':frob'/'frob' or
    ':DI'/'DI'; ``theta`` may be ':auto'/'auto' or a number in [0, 1]."""
    global last_stats
    if "θ" in kw:
        theta = kw.pop("θ")
    if kw:
        raise TypeError(f"gDCA() got unexpected keyword arguments {sorted(kw)}")
    check_arguments(filename, pseudocount, theta, max_gap_fraction, score, min_separation)

    ctx = ctx or default_context()
    if remove_dups:
        Z = read_fasta_alignment(filename, max_gap_fraction)
        Z, _ = remove_duplicate_sequences(Z)
        q = int(Z.max())
        if q >= 32:
            raise RuntimeError(f"parameter q={q} is too big (max 31 is allowed)")
        Zf = np.asfortranarray(Z, dtype=np.int8)
        ii, jj, sc, last_stats = ctx.run_ranked_ptr(Zf.ctypes.data, Zf.shape[0], Zf.shape[1], q, float(pseudocount), _theta_arg(theta),
                                                    _score_arg(score), int(min_separation), apc=True)
    else:
        # the parsed matrix goes to gdca_run where the native reader left it (gdca_fasta_data), q = maximum(Z) comes from the
        # reader (gdca_fasta_max_symbol): no copy into an array of the host language, no second pass over Z
        with FastaAlignment(filename, max_gap_fraction) as fa:
            q = fa.q
            if q >= 32:
                raise RuntimeError(f"parameter q={q} is too big (max 31 is allowed)")
            ii, jj, sc, last_stats = ctx.run_ranked_ptr(fa.ptr, fa.N, fa.M, q, float(pseudocount), _theta_arg(theta), _score_arg(score),
                                                        int(min_separation), apc=True)
    if last_stats.get("refined", 0) < 0:
        import warnings

        warnings.warn("gDCA: the covariance is too ill-conditioned for the block sweep even with its refinement step "
                      f"(||inv(C)||_1 = {last_stats['inverse_norm1']:.3g}; pseudocount {pseudocount}): scores are unreliable",
                      RuntimeWarning, stacklevel=2)
    return Ranking(ii, jj, sc)
