"""ctypes binding of libgdca.so (include/gdca.h).  There is no CPU fallback: if the shared
library is missing or no HIP device is usable, every entry point raises."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GDCA_LIB: another build of the same library (kernel experiments: tools/exp_*.sh build variants beside the product)
LIB_PATH = os.environ.get("GDCA_LIB") or os.path.join(_HERE, "libgdca.so")

GDCA_OK, GDCA_EINVAL, GDCA_ENOTPD, GDCA_EHIP, GDCA_ENOMEM, GDCA_ENOCONV = 0, 1, 2, 3, 4, 5
SCORE_FROB, SCORE_DI = 0, 1
ABI_VERSION = 6  # GDCA_VERSION_MAJOR * 1000 + GDCA_VERSION_MINOR of the header this binding mirrors


class GdcaError(RuntimeError):
    pass


class ArgumentError(ValueError):
    """Mirror of Julia's ArgumentError thrown by check_arguments (src/GaussDCA.jl:49-65)."""


class PosDefException(ArithmeticError):
    """Mirror of LinearAlgebra.PosDefException(info) thrown by cholesky (src/GaussDCA.jl:34)."""

    def __init__(self, info: int):
        super().__init__(f"matrix is not positive definite; Cholesky factorization failed (info={info})")
        self.info = int(info)


class ConvergenceError(ArithmeticError):
    """Mirror of the LAPACKException an eigenvalue routine raises inside DCAUtils.compute_DI_gauss
    (src/GaussDCA.jl:37) when its iteration does not converge."""

    def __init__(self, npairs: int):
        super().__init__(f"eigenvalue iteration did not converge for {npairs} site pair(s)")
        self.npairs = int(npairs)


class Params(C.Structure):
    _fields_ = [("pseudocount", C.c_double), ("theta", C.c_double), ("score", C.c_int32), ("apc", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [
        ("theta", C.c_double), ("Meff", C.c_double), ("pair_identity_sum", C.c_uint64),
        ("thresh", C.c_int32), ("info", C.c_int32),
        ("N", C.c_int32), ("M", C.c_int32), ("q", C.c_int32), ("n", C.c_int32), ("n_pad", C.c_int32),
        ("update_launches", C.c_int32), ("inverse_batch", C.c_int32), ("refined", C.c_int32),
        ("ms_total", C.c_double), ("ms_theta", C.c_double), ("ms_weights", C.c_double),
        ("ms_covariance", C.c_double), ("ms_inverse", C.c_double), ("ms_inverse_update", C.c_double),
        ("ms_score", C.c_double), ("inverse_flops", C.c_double), ("update_flops", C.c_double),
        ("sweep_ghz", C.c_double), ("inverse_norm1", C.c_double), ("matrix_norm1", C.c_double), ("cond_bound", C.c_double),
        ("ms_fn", C.c_double), ("ms_pair_tally", C.c_double), ("sweep_retries", C.c_int32), ("reserved0", C.c_int32),   # (new fields go to the END: an older build of the library fills a prefix)
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/gdca.h declares: name -> (restype, argtypes)
_i8p, _i32p, _f64p, _u64p = C.POINTER(C.c_int8), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_uint64)
_ctx = C.c_void_p
SYMBOLS = {
    "gdca_version": (C.c_int32, []),
    "gdca_stats_bytes": (C.c_int32, []),
    "gdca_params_bytes": (C.c_int32, []),
    "gdca_device_count": (C.c_int32, []),
    "gdca_ctx_create": (C.c_int, [C.c_int32, C.POINTER(_ctx)]),
    "gdca_ctx_create_on_stream": (C.c_int, [C.c_int32, C.c_void_p, C.POINTER(_ctx)]),
    "gdca_ctx_create_peer": (C.c_int, [_ctx, C.POINTER(_ctx)]),
    "gdca_ctx_destroy": (C.c_int, [_ctx]),
    "gdca_ctx_synchronize": (C.c_int, [_ctx]),
    "gdca_last_error": (C.c_char_p, [_ctx]),
    "gdca_ctx_set_timing": (C.c_int, [_ctx, C.c_int32]),
    "gdca_ctx_set_option": (C.c_int, [_ctx, C.c_char_p, C.c_char_p]),
    "gdca_run": (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Params), C.c_void_p,
                           C.POINTER(Stats)]),
    "gdca_run_dev": (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Params), C.c_void_p,
                               C.POINTER(Stats)]),
    "gdca_run_dev_async": (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Params), C.c_void_p]),
    "gdca_run_collect": (C.c_int, [_ctx, C.POINTER(Stats)]),
    "gdca_run_dev_phased": (C.c_int, [C.POINTER(_ctx), C.c_int32, C.POINTER(C.c_void_p), _i32p, _i32p, _i32p, C.POINTER(Params),
                                      C.POINTER(C.c_void_p)]),
    "gdca_dbuf_alloc": (C.c_int, [_ctx, C.c_uint64, C.POINTER(C.c_void_p)]),
    "gdca_dbuf_free": (C.c_int, [C.c_void_p]),
    "gdca_dbuf_ptr": (C.c_void_p, [C.c_void_p]),
    "gdca_dbuf_bytes": (C.c_uint64, [C.c_void_p]),
    "gdca_dbuf_upload": (C.c_int, [_ctx, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]),
    "gdca_dbuf_download": (C.c_int, [_ctx, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]),
    "gdca_pair_identity_sum_dev": (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_int32, _u64p]),
    "gdca_compute_theta_dev": (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_int32, _f64p]),
    "gdca_neighbour_counts_dev": (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "gdca_compute_weights_dev": (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_void_p, _f64p,
                                           _f64p, _i32p]),
    "gdca_frequencies_dev": (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_double,
                                       C.c_void_p, C.c_void_p]),
    "gdca_add_pseudocount_dev": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_void_p,
                                           C.c_void_p]),
    "gdca_covariance_dev": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "gdca_spd_inverse_dev": (C.c_int, [_ctx, C.c_void_p, C.c_int32, _i32p]),
    "gdca_spd_inverse_batch_dev": (C.c_int, [C.POINTER(_ctx), C.c_int32, C.POINTER(C.c_void_p), _i32p, _i32p]),
    "gdca_fn_dev": (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "gdca_di_dev": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "gdca_apc_dev": (C.c_int, [_ctx, C.c_void_p, C.c_int32]),
    "gdca_pair_identity_sum": (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_int32, _u64p]),
    "gdca_compute_theta": (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_int32, _f64p]),
    "gdca_neighbour_counts": (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "gdca_compute_weights": (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_void_p, _f64p, _f64p,
                                       _i32p]),
    "gdca_frequencies": (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_double,
                                   C.c_void_p, C.c_void_p]),
    "gdca_add_pseudocount": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_void_p,
                                       C.c_void_p]),
    "gdca_covariance": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "gdca_spd_inverse": (C.c_int, [_ctx, C.c_void_p, C.c_int32, _i32p]),
    "gdca_fn": (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "gdca_di": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "gdca_apc": (C.c_int, [_ctx, C.c_void_p, C.c_int32]),
    "gdca_host_cpus": (C.c_int32, []),
    "gdca_fasta_open": (C.c_int, [C.c_char_p, C.c_double, C.POINTER(C.c_void_p), _i32p, _i32p]),
    "gdca_fasta_copy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gdca_fasta_data": (C.c_void_p, [C.c_void_p]),
    "gdca_fasta_max_symbol": (C.c_int32, [C.c_void_p]),
    "gdca_fasta_close": (C.c_int, [C.c_void_p]),
    "gdca_remove_duplicates": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, _i32p]),
    "gdca_ranking_length": (C.c_int64, [C.c_int32, C.c_int32]),
    "gdca_ranking": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gdca_ranking_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gdca_run_ranked_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]),
    "gdca_run_ranked_phased_async": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "gdca_run_ranked_collect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gdca_run_ranked": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "gdca_write_rank": (C.c_int, [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "gdca_synth_family": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_void_p]),
    "gdca_write_fasta": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]),
    "gdca_probe_mfma_f64": (C.c_int, [_ctx, C.c_int32, _f64p]),
}

_lib = None


def load() -> C.CDLL:
    """Load libgdca.so and bind every declared symbol.  Raises GdcaError if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GdcaError(
            f"{LIB_PATH} not found: the HIP library has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    # the library writes sizeof(gdca_stats) bytes into the caller's struct: a build whose struct differs from this binding's
    # would overrun it (or leave fields unfilled) -- refuse it here instead (include/gdca.h, note at the end of gdca_stats)
    if lib.gdca_version() != ABI_VERSION or lib.gdca_stats_bytes() != C.sizeof(Stats) or lib.gdca_params_bytes() != C.sizeof(Params):
        raise GdcaError(f"{LIB_PATH} is version {lib.gdca_version()} with gdca_stats of {lib.gdca_stats_bytes()} bytes; this binding is "
                        f"written for version {ABI_VERSION} with {C.sizeof(Stats)} bytes: rebuild the library or update the binding")
    _lib = lib
    return lib


def _p(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


class Context:
    """One gdca_ctx = one HIP device + one stream + its workspace."""

    def __init__(self, device: int = 0, stream: int | None = None, _peer_of: "Context | None" = None):
        self.lib = load()
        h = _ctx()
        if _peer_of is not None:
            st = self.lib.gdca_ctx_create_peer(_peer_of.h, C.byref(h))
            device = _peer_of.device
        elif stream is None:
            st = self.lib.gdca_ctx_create(int(device), C.byref(h))
        else:
            st = self.lib.gdca_ctx_create_on_stream(int(device), C.c_void_p(stream), C.byref(h))
        if st != GDCA_OK:
            raise GdcaError(f"gdca_ctx_create(device={device}) failed with status {st}: no usable HIP device "
                            "(the gDCA hot path has no CPU fallback)")
        self.h = h
        self.device = int(device)

    def peer(self) -> "Context":
        """Another context on the same GPU whose inverse stages alternate with this one's (pipeline)."""
        return Context(_peer_of=self)

    def close(self):
        if getattr(self, "h", None):
            self.lib.gdca_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, st: int, info: int = 0):
        if st == GDCA_OK:
            return
        msg = self.lib.gdca_last_error(self.h)
        msg = msg.decode() if msg else ""
        if st == GDCA_EINVAL:
            raise ArgumentError(msg or "invalid argument")
        if st == GDCA_ENOTPD:
            raise PosDefException(info)
        if st == GDCA_ENOMEM:
            raise MemoryError(msg)
        if st == GDCA_ENOCONV:
            raise ConvergenceError(-info if info < 0 else 0)
        raise GdcaError(f"HIP error: {msg}")

    def synchronize(self):
        self.check(self.lib.gdca_ctx_synchronize(self.h))

    def set_timing(self, on: bool):
        self.check(self.lib.gdca_ctx_set_timing(self.h, 1 if on else 0))

    def set_option(self, key: str, value) -> None:
        """Tuning switch of this context (gdca_ctx_set_option): key = a GDCA_* variable's name, with or without the prefix."""
        self.check(self.lib.gdca_ctx_set_option(self.h, str(key).encode(), str(value).encode()))

    def set_options(self, **kv) -> "Context":
        for k, v in kv.items():
            self.set_option(k, v)
        return self

    # ---- fused path ----
    def run(self, Zf: np.ndarray, q: int, pseudocount: float, theta: float, score: int, apc: bool = True):
        """Zf: int8, shape (N, M), Fortran-contiguous.  Returns (S[N,N], stats dict)."""
        N, M = Zf.shape
        S = np.empty((N, N), dtype=np.float64, order="F")  # (the library writes column-major; compute_ranking takes it as it is)
        prm = Params(float(pseudocount), float(theta), int(score), 1 if apc else 0)
        st = Stats()
        rc = self.lib.gdca_run(self.h, _p(Zf), N, M, int(q), C.byref(prm), _p(S), C.byref(st))
        self.check(rc, st.info)
        return S, st.as_dict()

    def run_ptr(self, Z_ptr: int, N: int, M: int, q: int, pseudocount: float, theta: float, score: int, apc: bool = True):
        """gdca_run on a HOST matrix given by address (N x M int8, column-major: e.g. gdca_fasta_data).  Returns (S, stats)."""
        S = np.empty((N, N), dtype=np.float64, order="F")
        prm = Params(float(pseudocount), float(theta), int(score), 1 if apc else 0)
        st = Stats()
        rc = self.lib.gdca_run(self.h, C.c_void_p(Z_ptr), int(N), int(M), int(q), C.byref(prm), _p(S), C.byref(st))
        self.check(rc, st.info)
        return S, st.as_dict()

    def run_ranked_ptr(self, Z_ptr: int, N: int, M: int, q: int, pseudocount: float, theta: float, score: int, min_separation: int,
                       apc: bool = True):
        """gdca_run_ranked on a HOST matrix given by address: src/GaussDCA.jl:28-44 in one call, the ranking is sorted on the
        device and only it comes back.  Returns (i, j, score, stats): int32, int32, float64 arrays."""
        n = max(int(self.lib.gdca_ranking_length(int(N), int(min_separation))), 0)
        ii = np.empty(n, dtype=np.int32)
        jj = np.empty(n, dtype=np.int32)
        sc = np.empty(n, dtype=np.float64)
        prm = Params(float(pseudocount), float(theta), int(score), 1 if apc else 0)
        st = Stats()
        rc = self.lib.gdca_run_ranked(self.h, C.c_void_p(Z_ptr), int(N), int(M), int(q), C.byref(prm), int(min_separation), _p(ii), _p(jj),
                                      _p(sc), C.byref(st))
        self.check(rc, st.info)
        return ii, jj, sc, st.as_dict()

    def run_ranked_async_ptr(self, Z_ptr: int, N: int, M: int, q: int, pseudocount: float, theta: float, score: int, min_separation: int,
                             apc: bool = True):
        """First half of gdca_run_ranked: upload + enqueue, no waiting for the GPU; pair with run_ranked_collect()."""
        prm = Params(float(pseudocount), float(theta), int(score), 1 if apc else 0)
        self.check(self.lib.gdca_run_ranked_async(self.h, C.c_void_p(Z_ptr), int(N), int(M), int(q), C.byref(prm), int(min_separation)))
        self._ranked = (int(N), int(min_separation))

    def run_ranked_collect(self):
        """Second half: (i, j, score, stats) of the run enqueued by run_ranked_async_ptr."""
        N, sep = getattr(self, "_ranked", (1, 1))
        n = max(int(self.lib.gdca_ranking_length(N, sep)), 0)
        ii = np.empty(n, dtype=np.int32)
        jj = np.empty(n, dtype=np.int32)
        sc = np.empty(n, dtype=np.float64)
        st = Stats()
        rc = self.lib.gdca_run_ranked_collect(self.h, _p(ii), _p(jj), _p(sc), C.byref(st))
        self.check(rc, st.info)
        return ii, jj, sc, st.as_dict()

    def ranking_dev(self, S_ptr: int, N: int, min_separation: int):
        """compute_ranking of a score matrix in HBM (device pointer), sorted on the device: (i, j, score) host arrays."""
        n = max(int(self.lib.gdca_ranking_length(int(N), int(min_separation))), 0)
        ii = np.empty(n, dtype=np.int32)
        jj = np.empty(n, dtype=np.int32)
        sc = np.empty(n, dtype=np.float64)
        self.check(self.lib.gdca_ranking_dev(self.h, C.c_void_p(S_ptr), int(N), int(min_separation), _p(ii), _p(jj), _p(sc)))
        return ii, jj, sc

    def run_dev(self, Z_ptr: int, N: int, M: int, q: int, pseudocount: float, theta: float, score: int,
                S_ptr: int, apc: bool = True):
        """Device-pointer form (Z and S resident in HBM).  Returns the stats dict."""
        prm = Params(float(pseudocount), float(theta), int(score), 1 if apc else 0)
        st = Stats()
        rc = self.lib.gdca_run_dev(self.h, C.c_void_p(Z_ptr), N, M, int(q), C.byref(prm), C.c_void_p(S_ptr),
                                   C.byref(st))
        self.check(rc, st.info)
        return st.as_dict()


    def run_dev_async(self, Z_ptr: int, N: int, M: int, q: int, pseudocount: float, theta: float, score: int,
                      S_ptr: int, apc: bool = True):
        """Enqueue only (no host synchronisation); pair with collect()."""
        prm = Params(float(pseudocount), float(theta), int(score), 1 if apc else 0)
        self.check(self.lib.gdca_run_dev_async(self.h, C.c_void_p(Z_ptr), N, M, int(q), C.byref(prm),
                                               C.c_void_p(S_ptr)))

    def collect(self):
        st = Stats()
        rc = self.lib.gdca_run_collect(self.h, C.byref(st))
        self.check(rc, st.info)
        return st.as_dict()


def run_dev_phased(ctxs, Z_ptrs, Ns, Ms, qs, pseudocount: float, theta: float, score: int, S_ptrs, apc: bool = True):
    """K families batched by phase on one GPU (gdca_run_dev_phased): K front ends, K inverses back to back, K score
    stages.  Enqueues only; collect() every context afterwards."""
    K = len(ctxs)
    assert K >= 1 and len(Z_ptrs) == len(Ns) == len(Ms) == len(qs) == len(S_ptrs) == K
    prm = Params(float(pseudocount), float(theta), int(score), 1 if apc else 0)
    hs = (_ctx * K)(*[c.h for c in ctxs])
    zp = (C.c_void_p * K)(*[C.c_void_p(int(z)) for z in Z_ptrs])
    sp = (C.c_void_p * K)(*[C.c_void_p(int(x)) for x in S_ptrs])
    i32 = lambda v: (C.c_int32 * K)(*[int(x) for x in v])  # noqa: E731
    # (the library copies a failing member's message to the leader: ctxs[0]'s last_error names the member)
    ctxs[0].check(ctxs[0].lib.gdca_run_dev_phased(hs, K, zp, i32(Ns), i32(Ms), i32(qs), C.byref(prm), sp))


def run_ranked_phased_async(ctxs, Z_ptrs, Ns, Ms, qs, pseudocount: float, theta: float, score: int, min_separation: int, apc: bool = True):
    """K families (HOST matrices given by address) through gdca_run_ranked_phased_async: uploads, the phase-batched hot path with
    the small inverses merged, every member's ranking; nothing waited for.  run_ranked_collect() each context afterwards."""
    K = len(ctxs)
    assert K >= 1 and len(Z_ptrs) == len(Ns) == len(Ms) == len(qs) == K
    prm = Params(float(pseudocount), float(theta), int(score), 1 if apc else 0)
    hs = (_ctx * K)(*[c.h for c in ctxs])
    zp = (C.c_void_p * K)(*[C.c_void_p(int(z)) for z in Z_ptrs])
    i32 = lambda v: (C.c_int32 * K)(*[int(x) for x in v])  # noqa: E731
    ctxs[0].check(ctxs[0].lib.gdca_run_ranked_phased_async(hs, K, zp, i32(Ns), i32(Ms), i32(qs), C.byref(prm), int(min_separation)))
    for c, n in zip(ctxs, Ns):
        c._ranked = (int(n), int(min_separation))


def spd_inverse_batch_dev(ctxs, A_ptrs, ns):
    """K independent SPD inverses in place on device matrices (gdca_spd_inverse_batch_dev): the small ones share merged launches
    of the sweep kernel.  Returns the list of info values; raises PosDefException(info of the first failing member)."""
    K = len(ctxs)
    assert K >= 1 and len(A_ptrs) == len(ns) == K
    hs = (_ctx * K)(*[c.h for c in ctxs])
    ap = (C.c_void_p * K)(*[C.c_void_p(int(a)) for a in A_ptrs])
    nn = (C.c_int32 * K)(*[int(x) for x in ns])
    info = (C.c_int32 * K)()
    rc = ctxs[0].lib.gdca_spd_inverse_batch_dev(hs, K, ap, nn, info)
    ctxs[0].check(rc, next((int(i) for i in info if i), 0))
    return [int(i) for i in info]


class DeviceBuffer:
    """An owned HBM allocation (gdca_dbuf): what a host language without a GPU array type keeps the
    intermediates of the statement-by-statement pipeline in.  `.ptr` is the device pointer the `_dev`
    operators take."""

    def __init__(self, ctx: Context, nbytes: int):
        self.ctx = ctx
        h = C.c_void_p()
        ctx.check(ctx.lib.gdca_dbuf_alloc(ctx.h, int(nbytes), C.byref(h)))
        self.h = h
        self.nbytes = int(ctx.lib.gdca_dbuf_bytes(h))
        self.ptr = int(ctx.lib.gdca_dbuf_ptr(h) or 0)

    @classmethod
    def from_array(cls, ctx: Context, a: np.ndarray) -> "DeviceBuffer":
        a = np.ascontiguousarray(a) if not (a.flags.c_contiguous or a.flags.f_contiguous) else a
        b = cls(ctx, a.nbytes)
        b.upload(a)
        return b

    def upload(self, a: np.ndarray, offset: int = 0):
        assert a.flags.c_contiguous or a.flags.f_contiguous
        self.ctx.check(self.ctx.lib.gdca_dbuf_upload(self.ctx.h, self.h, int(offset), _p(a), a.nbytes))

    def download(self, shape, dtype=np.float64, order="F", offset: int = 0) -> np.ndarray:
        out = np.empty(shape, dtype=dtype, order=order)
        self.ctx.check(self.ctx.lib.gdca_dbuf_download(self.ctx.h, self.h, int(offset), _p(out), out.nbytes))
        return out

    def free(self):
        if getattr(self, "h", None):
            self.ctx.lib.gdca_dbuf_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


_default_ctx = None


def default_context() -> Context:
    """Process-wide context on the device named by LOCAL_RANK (one process per GPU), else 0."""
    global _default_ctx
    if _default_ctx is None:
        dev = int(os.environ.get("GDCA_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        ndev = load().gdca_device_count()
        if ndev <= 0:
            raise GdcaError("no HIP device visible: the gDCA hot path has no CPU fallback")
        ids = visible_devices(ndev)
        _default_ctx = Context(ids[dev % len(ids)])
    return _default_ctx


def visible_devices(ndev: int):
    """GDCA_VISIBLE_DEVICES=0,2,5: the HIP devices this process may use, in that order; unset = all.
    Same rule as gdca_cli --batch: ids outside 0..ndev-1 and repeats are an error."""
    env = os.environ.get("GDCA_VISIBLE_DEVICES", "")
    if not env:
        return list(range(ndev))
    ids = []
    for tok in env.split(","):
        if not tok.strip().isdigit() or int(tok) >= ndev or int(tok) in ids:
            raise ArgumentError(f"invalid GDCA_VISIBLE_DEVICES entry '{tok}' ({ndev} HIP device(s) present)")
        ids.append(int(tok))
    return ids
