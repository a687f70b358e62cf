"""Device-resident form of the reference's statement-by-statement pipeline (src/GaussDCA.jl:28-42).

The host-pointer operators of :mod:`dcautils` move every n x n matrix over PCIe twice; a caller who keeps the
reference's six statements should instead keep the arrays in HBM (``DeviceBuffer`` = ``gdca_dbuf``) and call the
``_dev`` entry points of include/gdca.h.  Only ``Z`` goes in and ``S`` comes out.  The functions here carry the
DCAUtils / GaussDCA names with a ``_dev`` suffix and take / return ``DeviceBuffer`` objects (or raw device
pointers as ints, e.g. ``tensor.data_ptr()``); ``gDCA_stepwise`` chains them exactly as ``gDCA`` does at
:28-:42 and gives the same bits as the fused ``gdca_run``.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np

from . import _lib
from ._lib import ArgumentError, Context, DeviceBuffer, default_context
from .dcautils import _theta_arg, _zf, compute_ranking


def _ptr(x) -> C.c_void_p:
    return C.c_void_p(x.ptr if isinstance(x, DeviceBuffer) else int(x))


def compute_weights_dev(ctx: Context, dZ, N: int, M: int, theta=":auto", dW=None):
    """compute_weights(Z, q, θ) with Z and W in HBM -> (dW, Meff, theta_used, thresh)"""
    dW = dW or DeviceBuffer(ctx, 8 * M)
    Meff, th, thr = C.c_double(), C.c_double(), C.c_int32()
    ctx.check(ctx.lib.gdca_compute_weights_dev(ctx.h, _ptr(dZ), N, M, _theta_arg(theta), _ptr(dW), C.byref(Meff),
                                               C.byref(th), C.byref(thr)))
    return dW, Meff.value, th.value, int(thr.value)


def compute_weighted_frequencies_dev(ctx: Context, dZ, N: int, M: int, q: int, dW, Meff: float, dPi=None, dPij=None):
    """the accumulation of compute_weighted_frequencies (:28) -> (dPi_true[n], dPij_true[n x n])"""
    n = N * (q - 1)
    dPi = dPi or DeviceBuffer(ctx, 8 * n)
    dPij = dPij or DeviceBuffer(ctx, 8 * n * n)
    ctx.check(ctx.lib.gdca_frequencies_dev(ctx.h, _ptr(dZ), N, M, int(q), _ptr(dW), float(Meff), _ptr(dPi), _ptr(dPij)))
    return dPi, dPij


def add_pseudocount_dev(ctx: Context, dPi_true, dPij_true, N: int, q: int, pc: float, dPi=None, dPij=None):
    """add_pseudocount (:30); in place when no output buffers are given"""
    dPi = dPi or dPi_true
    dPij = dPij or dPij_true
    ctx.check(ctx.lib.gdca_add_pseudocount_dev(ctx.h, _ptr(dPi_true), _ptr(dPij_true), N, int(q), float(pc), _ptr(dPi),
                                               _ptr(dPij)))
    return dPi, dPij


def compute_C_dev(ctx: Context, dPi, dPij, n: int, dC=None):
    """compute_C (:32, :76); C may alias Pij"""
    dC = dC or DeviceBuffer(ctx, 8 * n * n)
    ctx.check(ctx.lib.gdca_covariance_dev(ctx.h, _ptr(dPi), _ptr(dPij), int(n), _ptr(dC)))
    return dC


def inv_cholesky_dev(ctx: Context, dA, n: int):
    """mJ = inv(cholesky(C)) (:34), in place; raises PosDefException(info) like Julia"""
    info = C.c_int32()
    rc = ctx.lib.gdca_spd_inverse_dev(ctx.h, _ptr(dA), int(n), C.byref(info))
    ctx.check(rc, info.value)
    return dA


def compute_FN_dev(ctx: Context, dmJ, N: int, q: int, dS=None):
    dS = dS or DeviceBuffer(ctx, 8 * N * N)
    ctx.check(ctx.lib.gdca_fn_dev(ctx.h, _ptr(dmJ), N, int(q), _ptr(dS)))
    return dS


def compute_DI_gauss_dev(ctx: Context, dmJ, dC, N: int, q: int, dS=None):
    dS = dS or DeviceBuffer(ctx, 8 * N * N)
    ctx.check(ctx.lib.gdca_di_dev(ctx.h, _ptr(dmJ), _ptr(dC), N, int(q), _ptr(dS)))
    return dS


def correct_APC_dev(ctx: Context, dS, N: int):
    ctx.check(ctx.lib.gdca_apc_dev(ctx.h, _ptr(dS), N))
    return dS


def scores_stepwise(Z, q: int, pseudocount: float = 0.8, theta=":auto", score: str = "frob", ctx: Context = None
                    ) -> Tuple[np.ndarray, dict]:
    """The six statements of src/GaussDCA.jl:28-42 one by one, arrays resident in HBM.  Z: (N, M) int8.
    Returns (S[N, N], dict(theta, thresh, Meff))."""
    ctx = ctx or default_context()
    Zf = _zf(Z)
    N, M = Zf.shape
    if q >= 32:
        raise ArgumentError(f"parameter q={q} is too big (max 31 is allowed)")
    n = N * (q - 1)
    dZ = DeviceBuffer.from_array(ctx, Zf)
    # Pi_true, Pij_true, Meff, _ = compute_weighted_frequencies(Z, q, θ)                      (:28)
    dW, Meff, th, thr = compute_weights_dev(ctx, dZ, N, M, theta)
    dPi, dPij = compute_weighted_frequencies_dev(ctx, dZ, N, M, q, dW, Meff)
    # Pi, Pij = add_pseudocount(Pi_true, Pij_true, Float64(pseudocount), q)                   (:30)
    add_pseudocount_dev(ctx, dPi, dPij, N, q, pseudocount)
    # C = compute_C(Pi, Pij)                                                                  (:32)
    is_di = str(score).lstrip(":") == "DI"
    dC = compute_C_dev(ctx, dPi, dPij, n) if is_di else None      # compute_DI_gauss needs C as well as mJ
    dmJ = compute_C_dev(ctx, dPi, dPij, n, dC=dPij)
    # mJ = inv(cholesky(C))                                                                   (:34)
    inv_cholesky_dev(ctx, dmJ, n)
    # S = compute_DI_gauss(mJ, C, q) | compute_FN(mJ, q)                                      (:36-40)
    dS = compute_DI_gauss_dev(ctx, dmJ, dC, N, q) if is_di else compute_FN_dev(ctx, dmJ, N, q)
    # S = correct_APC(S)                                                                      (:42)
    correct_APC_dev(ctx, dS, N)
    S = dS.download((N, N))
    for b in (dZ, dW, dPi, dPij, dS) + ((dC,) if dC is not None else ()):
        b.free()
    return S, dict(theta=th, thresh=thr, Meff=Meff)


def gDCA_stepwise(Z, q: int, pseudocount: float = 0.8, theta=":auto", score: str = "frob", min_separation: int = 5,
                  ctx: Context = None):
    S, _ = scores_stepwise(Z, q, pseudocount, theta, score, ctx)
    return compute_ranking(S, int(min_separation))
