/* gdca.h -- C-ABI of libgdca.so: the MI355X (gfx950) Gaussian-DCA hot path.
 *
 * This is the drop-in boundary for the one path of carlobaldassi/GaussDCA.jl that this
 * project accelerates: everything between src/GaussDCA.jl:28 and :42 of the reference
 *
 *     Pi_true, Pij_true, Meff, _ = compute_weighted_frequencies(Z, q, theta)   (:28)
 *     Pi, Pij = add_pseudocount(Pi_true, Pij_true, Float64(pseudocount), q)     (:30)
 *     C  = compute_C(Pi, Pij)                                                   (:32, :76)
 *     mJ = inv(cholesky(C))                                                     (:34)
 *     S  = score == :DI ? compute_DI_gauss(mJ, C, q) : compute_FN(mJ, q)        (:36-40)
 *     S  = correct_APC(S)                                                       (:42, :78-86)
 *
 * The reference is Julia; it would bind these entry points with `ccall((:sym, libgdca), ...)`
 * (the stub a maintainer would add is in INTEGRATION.md).  Plain pointers and sizes only: no
 * C++ types, no exceptions, no Julia/Python/torch types cross this ABI.
 *
 * Conventions
 *   - Matrices are COLUMN-MAJOR (Julia native), so Julia arrays are passed without copies.
 *   - Z is the reference's `Z::Matrix{Int8}`, N x M column-major: sequence k is the N
 *     contiguous bytes Z[k*N .. k*N+N-1], symbols 1..q, q <= 31 (src/GaussDCA.jl:24-26).
 *   - s = q - 1 (state q, the gap, has no row/column); n = N * s.
 *   - "host" pointers are ordinary host memory owned by the caller; "_dev" entry points take
 *     device (HBM) pointers owned by the caller.  The library never returns memory it
 *     allocated except the opaque gdca_ctx.
 *   - Every function returns a gdca_status; it never aborts and never throws.
 *   - A gdca_ctx owns one HIP device, one stream and a grow-only device workspace.  A ctx is
 *     used by one host thread at a time; different ctxs are independent (this is the
 *     multi-GPU model: one ctx per GPU, no collectives).
 *   - There is NO CPU fallback: without a usable HIP device gdca_ctx_create fails with
 *     GDCA_EHIP and nothing else can be called.
 */
#ifndef GDCA_H
#define GDCA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Largest covariance the entry points accept: n = N (q-1) <= GDCA_MAX_N (a 60 000 x 60 000 f64 matrix is 28.8 GB of the
 * 288 GB of HBM; the sweep kernel's item tables are 32-bit).  Exercised up to n = 48 000. */
#define GDCA_MAX_N 60000

#define GDCA_VERSION_MAJOR 0
#define GDCA_VERSION_MINOR 6

typedef enum gdca_status {
    GDCA_OK = 0,
    GDCA_EINVAL = 1, /* bad argument: mirrors ArgumentError (src/GaussDCA.jl:50-62) and q >= 32 (:26) */
    GDCA_ENOTPD = 2, /* covariance not positive definite: mirrors PosDefException(info) from :34 */
    GDCA_EHIP = 3,   /* HIP runtime error; see gdca_last_error() */
    GDCA_ENOMEM = 4, /* device or host allocation failed */
    GDCA_ENOCONV = 5 /* eigenvalue iteration inside compute_DI_gauss did not converge: mirrors the LAPACKException
                        eigen()/eigvals() would raise inside DCAUtils (call site :37); stats.info = -(pairs affected) */
} gdca_status;

enum { GDCA_SCORE_FROB = 0, GDCA_SCORE_DI = 1 }; /* score = :frob | :DI (src/GaussDCA.jl:14) */

typedef struct gdca_ctx gdca_ctx;
typedef struct gdca_dbuf gdca_dbuf; /* an owned HBM allocation (see "device buffers" below) */

/* Keyword arguments of gDCA() that reach the hot path (src/GaussDCA.jl:10-15). */
typedef struct gdca_params {
    double pseudocount; /* 0 <= pc <= 1            (default 0.8, :11)                       */
    double theta;       /* 0 <= theta <= 1, or any negative value for theta = :auto (:12)   */
    int32_t score;      /* GDCA_SCORE_FROB | GDCA_SCORE_DI (:14)                            */
    int32_t apc;        /* 1: apply correct_APC (what gDCA does, :42); 0: raw scores        */
} gdca_params;

/* What DCAUtils prints (theta, threshold, Meff) plus device timings of the last run. */
typedef struct gdca_stats {
    double theta;               /* theta used (the estimate when theta = :auto)             */
    double Meff;                /* effective number of sequences                            */
    uint64_t pair_identity_sum; /* sum_{k<l} #{i: Z[i,k]==Z[i,l]} (0 unless theta = :auto)  */
    int32_t thresh;             /* floor(theta * N)                                         */
    int32_t info;               /* 0; k>0: leading minor k of C not positive definite;
                                   k<0: -k site pairs whose DI eigenvalue iteration failed  */
    int32_t N, M, q, n, n_pad;  /* n = N(q-1); n_pad = n rounded up to the tile size        */
    int32_t update_launches;    /* launches of the dominant kernel this run accounts for (a merged launch: its first member) */
    int32_t inverse_batch;      /* families that shared this run's SPD-inverse launch (gdca_run_dev_phased merges the small ones;
                                   1 = a launch of its own).  ms_inverse and ms_inverse_update are the launch's time divided by it */
    int32_t refined;            /* 1: the covariance is ill-conditioned (kappa_1 = matrix_norm1 * inverse_norm1 beyond REFINE_COND) and the inverse got a Newton-Schulz
                                   step, the scores were computed again from it (the ms_* are those of the first pass);
                                   -1: the step was taken but cannot have converged (residual |I - X C| >= 1: cond(C) beyond ~1e10,
                                   where the sweep's own error is of order one) and option CHOLESKY is 0 -- the scores are not
                                   to be trusted;
                                   2: the sweep gave up (a non-positive pivot of its own, or the step above could not converge) and
                                   the inverse was computed again by blocked Cholesky (dpotrf + dpotri, as the reference does);
                                   `info` and the scores are those of that factorisation */
    /* device time (HIP events on the ctx stream), milliseconds */
    double ms_total;            /* Z in HBM -> S in HBM                                     */
    double ms_theta;            /* column histograms + theta                                */
    double ms_weights;          /* bit-plane pack + all-pairs Hamming + W, Meff             */
    double ms_covariance;       /* Pi + pair tallies + pseudocount + covariance build       */
    double ms_inverse;          /* SPD inverse, whole stage                                 */
    double ms_inverse_update;   /* sum over launches of the dominant kernel                 */
    double ms_score;            /* FN or DI, + APC                                          */
    double inverse_flops;       /* n^3/3+n^2/2+n/6 + 2n^3/3+n^2/2+5n/6 (dpotrf+dpotri)      */
    double update_flops;        /* flops executed by all launches of the dominant kernel    */
    double sweep_ghz;           /* shader clock during the SPD-inverse kernel, measured by the kernel itself
                                   (s_memtime cycles / 100 MHz wall-clock ticks, summed over its workgroups); 0 if no inverse ran */
    double inverse_norm1;       /* ||inv(C)||_1 as the sweep left it, measured (one pass over the inverse) only where cond_bound is
                                   beyond REFINE_COND: the decision on a refinement is kappa_1 = matrix_norm1 * inverse_norm1 >
                                   REFINE_COND.  0: not measured (the bound settled it, or option REFINE=0) */
    double matrix_norm1;        /* ||C||_1 of the covariance (src/GaussDCA.jl:32), measured (one pass over C before the sweep overwrites
                                   it) only where the bound that costs nothing, ||C||_1 <= 2 N max_i Pi(i), leaves cond_bound beyond
                                   REFINE_COND: 64 .. 94 on the reference's test/data/large.fasta.gz, not "of order one".  0: not
                                   measured (the cheap bound settled it, or option REFINE=0) */
    double cond_bound;          /* a-priori bound of cond_2(C): ||C||_1 q^2 / pseudocount, with matrix_norm1 where it was measured
                                   and 2 N max Pi otherwise.  The covariance with pseudocount pc is that of a mixture with weight pc
                                   on independent uniform columns, hence lambda_min(C) >= pc / q^2 (sharp whenever some column has
                                   no gap).  Every run whose bound is below REFINE_COND is KNOWN to be well enough conditioned
                                   for the sweep and pays nothing for the screen; +inf at pseudocount 0; 0 with option REFINE=0 */
    /* Fields are only ever added at the END, and every addition bumps GDCA_VERSION_MINOR (0.5: the five fields from
     * matrix_norm1 on; 0.6: sweep_retries).  The LIBRARY fills sizeof(gdca_stats) as IT was built: a binding built against a newer header than
     * the library reads a valid prefix (the rest stays as the caller initialised it), but a binding built against an OLDER,
     * shorter struct would be overrun -- so a binding compares gdca_stats_bytes() with its own struct size (and
     * gdca_version() with the minor it was written for) once at load time and refuses to run on a mismatch, as the Python
     * and Julia bindings of this repository do. */
    double ms_fn;               /* the FN kernel alone (HBM-bound: one pass over the lower block triangle of the inverse,
                                   8 n (n - s) / 2 bytes); 0 for the DI score and for a run refined at collect time */
    double ms_pair_tally;       /* the pair-tally kernel alone, with its pseudocount + covariance epilogue (its one
                                   compulsory HBM write: 8 n^2 bytes); 0 with option REFINE=0                      */
    int32_t sweep_retries;      /* 0.6: times this run's SPD inverse was run AGAIN because the sweep kernel's watchdog had ended
                                   its launch (normally 0; see "Contexts and the device's queues" below).  The result is that
                                   of the last attempt, the ms_* of the inverse and of the score stage are the last attempt's */
    int32_t reserved0;          /* (keeps the struct a multiple of 8 bytes; 0) */
} gdca_stats;

/* ---- library / context ---------------------------------------------------------------- */
int32_t gdca_version(void); /* major*1000 + minor */
/* sizeof(gdca_stats) / sizeof(gdca_params) as this build of the library writes / reads them (see the note at the end of gdca_stats) */
int32_t gdca_stats_bytes(void);
int32_t gdca_params_bytes(void);
int32_t gdca_device_count(void);

/* Contexts and the device's queues.  A context owns a stream, and a stream's hardware queue comes into being with its first command;
 * the driver maps a new queue by taking EVERY queue of the device off the hardware and back -- the waves of all running kernels are
 * saved and restored (~1.7 ms).  A persistent sweep restored beside the kernels of other queues may not get all its workgroups back
 * and then stands still until its watchdog ends it (round 6; gdca_api.hip, warm_stream; k_inverse.hip, spin_until).  Therefore:
 *   - gdca_ctx_create / _create_peer make the stream's queue at once (a small fill and a synchronisation: ~0.2 ms more per context);
 *   - make the contexts a program needs BEFORE it enqueues work, as the batch driver and bench.py do.  A context on a caller's stream
 *     (gdca_ctx_create_on_stream) is the caller's business: run something on that stream first;
 *   - what a program cannot rule out (another process starting on the GPU, a library of its own that makes a stream) is survived: the
 *     sweep notices that its waves were off the hardware and ends a launch that no longer moves after ~0.1 s instead of seconds, and
 *     gdca_run_collect / gdca_spd_inverse_dev run that inverse again -- up to option SWEEP_RETRIES (default 2) times, reported in
 *     gdca_stats.sweep_retries; the results are those of an undisturbed run, bit for bit.  Only an inverse that fails every attempt
 *     is GDCA_EHIP ("a dependency wait inside the sweep kernel timed out"). */
gdca_status gdca_ctx_create(int32_t device_id, gdca_ctx **out);
/* same, but enqueue on an existing hipStream_t (passed as void*); NULL = the null stream */
gdca_status gdca_ctx_create_on_stream(int32_t device_id, void *hip_stream, gdca_ctx **out);
/* A further context on the leader's device that forms a PIPELINE with it: runs enqueued on the members
 * execute their SPD-inverse stages one after the other (device-side event chain), so that with
 * gdca_run_dev_async the reweighting/tally stages of the next family overlap the inverse of the current
 * one.  All members of a pipeline are driven by one host thread.  Contexts that run on one GPU at the same time SHOULD be such
 * peers: the SPD inverse is a persistent kernel, and two of them side by side only take turns for the compute units.  Sharing the device
 * WITHOUT a gate is slower, not unsafe: launches of one family each and merged launches (gdca_run_dev_phased) beside another sweep -- two
 * non-peer contexts driven by two threads, two processes -- all finish with the right inverses (tested: INTEGRATION.md, "Sharing a device"). */
gdca_status gdca_ctx_create_peer(gdca_ctx *leader, gdca_ctx **out);
gdca_status gdca_ctx_destroy(gdca_ctx *ctx);
gdca_status gdca_ctx_synchronize(gdca_ctx *ctx);
const char *gdca_last_error(gdca_ctx *ctx); /* valid until the next call on ctx */
/* per-stage device timing (HIP events + one stream sync per run); default on */
gdca_status gdca_ctx_set_timing(gdca_ctx *ctx, int32_t enabled);
/* Tuning switch of THIS context (no process-global state: a context reads the GDCA_* environment variables once, when it is
 * created; afterwards only this call changes them, and two contexts of one process may differ).  key = the variable's name with
 * or without the GDCA_ prefix, any case; value = what the variable would hold.  Schedule of the SPD inverse: GROUP (1..4, -1 = the
 * measured rule), RAMP, RAGGED, REM_TAIL, PANEL_HALVES, SLAB, RING, MCUS, MCU_SOLO; SWEEP_TIMEOUT_MS (bound of one dependency wait inside
 * the sweep kernel; 0 = scaled with the problem, at least 4 s), SWEEP_RETRIES (0..5: further attempts of an inverse whose launch the watchdog ended), SWEEP_DEBUG, SWEEP_TRACE (file); HAMMING_MODE (auto | full |
 * bound | mfma: the bit-count lower bound on the fp4 matrix pipe -- exact counts, never the automatic choice), FORCE_FALLBACK (the independent byte-compare Hamming kernel, cf. DCAUTILS_FORCE_FALLBACK in test/runtests.jl:78-86),
 * TALLY_TJ; MERGE (families per merged SPD-inverse launch in gdca_run_dev_phased, 1 = off), MERGE_BLOCKS (largest member, in
 * 128-blocks), MERGE_TILES, MERGE_GROUP, MERGE_MCUS, PHASED_FRONTS (1: the front ends of a phase batch run side by side on the
 * members' streams, 0: one after the other), PHASED_STREAMS (how many of those streams, the first members', they are spread over: default 4), PHASED_GRIDS (the kernels of a phase batch as ONE grid per kernel kind carrying all members of a group: -1 = 1 group for small families, 4 for big ones [default], 1 .. 8 = that many groups side by side, 0 = a launch per member and kernel as PHASED_FRONTS describes); REFINE (auto | 0 | 1: one Newton-Schulz step on an inverse that looks
 * ill-conditioned / never / always) and REFINE_COND (the threshold of auto, default 1e6); CHOLESKY (0 | 1 | 2: the blocked
 * dpotrf + dpotri fallback never / where the sweep gave up [default] / for every inverse).  The schedule switches change results
 * at rounding level at most (another summation order); REFINE improves an ill-conditioned inverse.  GDCA_EINVAL: unknown key
 * or unusable value. */
gdca_status gdca_ctx_set_option(gdca_ctx *ctx, const char *key, const char *value);

/* ---- fused hot path: replaces src/GaussDCA.jl:28-42 in one call ------------------------ */
/* Z_host: N x M int8 (host).  S_host: N x N f64 column-major (host), caller-owned.
 * Only Z goes in and S + stats come out over PCIe. */
gdca_status gdca_run(gdca_ctx *ctx, const int8_t *Z_host, int32_t N, int32_t M, int32_t q,
                     const gdca_params *p, double *S_host, gdca_stats *st);
/* Same with Z and S resident in HBM (device pointers). Synchronises the ctx stream before
 * returning so that *st is complete. */
gdca_status gdca_run_dev(gdca_ctx *ctx, const int8_t *Z_dev, int32_t N, int32_t M, int32_t q,
                         const gdca_params *p, double *S_dev, gdca_stats *st);

/* Split form for pipelining independent families over several contexts on one GPU: _async only
 * enqueues (no host synchronisation); gdca_run_collect waits for that run, fills *st and returns the run's
 * status.  One run may be outstanding per ctx (a second _async before the collect is GDCA_EINVAL); the Z and S
 * buffers must stay valid until it is collected. */
gdca_status gdca_run_dev_async(gdca_ctx *ctx, const int8_t *Z_dev, int32_t N, int32_t M, int32_t q,
                               const gdca_params *p, double *S_dev);
gdca_status gdca_run_collect(gdca_ctx *ctx, gdca_stats *st);
/* K independent families batched BY PHASE on one GPU (throughput form of the same path; the reference's only statement on
 * parallelism is "independent families", README.md:92-94): the K front ends (theta, reweighting, tallies, covariance), then
 * the K SPD inverses back to back, then the K score stages, all on the stream of ctxs[0].  The matrix pipes see one
 * uninterrupted stretch of MFMA work per batch instead of K load steps (the clock governor answers every step from the
 * front end's VALU/LDS work to the inverse with a dip that costs ~10 % of an inverse at n = 10 000), and results are bit for
 * bit those of K single runs.  K distinct contexts of one device (each keeps its own workspace: K covariance matrices
 * live at once), none with a run outstanding; arrays of K entries; one parameter set.  Enqueues only: collect every
 * member with gdca_run_collect (any order).  K <= 64. */
gdca_status gdca_run_dev_phased(gdca_ctx *const *ctxs, int32_t K, const int8_t *const *Z_dev, const int32_t *N,
                                const int32_t *M, const int32_t *q, const gdca_params *p, double *const *S_dev);

/* ---- device buffers ---------------------------------------------------------------------- */
/* For callers without a GPU array type of their own (the Julia shim): an owned allocation in the HBM of ctx's
 * device.  gdca_dbuf_ptr() is the plain device pointer the `_dev` entry points take (any other device pointer of
 * the same GPU, e.g. a torch tensor's data_ptr, is as good).  upload/download are synchronous. */
gdca_status gdca_dbuf_alloc(gdca_ctx *ctx, uint64_t bytes, gdca_dbuf **out);
gdca_status gdca_dbuf_free(gdca_dbuf *buf);
void *gdca_dbuf_ptr(const gdca_dbuf *buf);
uint64_t gdca_dbuf_bytes(const gdca_dbuf *buf);
gdca_status gdca_dbuf_upload(gdca_ctx *ctx, gdca_dbuf *buf, uint64_t offset, const void *host, uint64_t bytes);
gdca_status gdca_dbuf_download(gdca_ctx *ctx, const gdca_dbuf *buf, uint64_t offset, void *host, uint64_t bytes);

/* ---- operator level, device-resident: the statements of src/GaussDCA.jl:28-42 one by one with every array in HBM --
 * Same meaning as the host-pointer operators below (which are "upload, call the _dev form, download"); array
 * arguments are device pointers, scalars come back through host pointers.  Dense matrices are n x n column-major
 * with leading dimension n (n = N (q-1)).  A maintainer who keeps the reference's statement-by-statement gDCA()
 * moves only Z in and S out over PCIe:
 *     W, Meff           <- gdca_compute_weights_dev            (compute_weights inside :28)
 *     Pi_true, Pij_true <- gdca_frequencies_dev                (:28)
 *     Pi, Pij           <- gdca_add_pseudocount_dev            (:30; may run in place)
 *     C                 <- gdca_covariance_dev                 (:32, :76; C may alias Pij)
 *     mJ                <- gdca_spd_inverse_dev                (:34; in place)
 *     S                 <- gdca_fn_dev | gdca_di_dev           (:37, :39)
 *     S                 <- gdca_apc_dev                        (:42; in place)                                         */
gdca_status gdca_pair_identity_sum_dev(gdca_ctx *ctx, const int8_t *Z_dev, int32_t N, int32_t M, uint64_t *out);
gdca_status gdca_compute_theta_dev(gdca_ctx *ctx, const int8_t *Z_dev, int32_t N, int32_t M, double *theta);
gdca_status gdca_neighbour_counts_dev(gdca_ctx *ctx, const int8_t *Z_dev, int32_t N, int32_t M, int32_t thresh,
                                      int32_t *n_dev);
gdca_status gdca_compute_weights_dev(gdca_ctx *ctx, const int8_t *Z_dev, int32_t N, int32_t M, double theta,
                                     double *W_dev, double *Meff, double *theta_used, int32_t *thresh);
gdca_status gdca_frequencies_dev(gdca_ctx *ctx, const int8_t *Z_dev, int32_t N, int32_t M, int32_t q,
                                 const double *W_dev, double Meff, double *Pi_dev, double *Pij_dev);
gdca_status gdca_add_pseudocount_dev(gdca_ctx *ctx, const double *Pi_true_dev, const double *Pij_true_dev, int32_t N,
                                     int32_t q, double pc, double *Pi_dev, double *Pij_dev);
gdca_status gdca_covariance_dev(gdca_ctx *ctx, const double *Pi_dev, const double *Pij_dev, int32_t n, double *C_dev);
gdca_status gdca_spd_inverse_dev(gdca_ctx *ctx, double *A_dev, int32_t n, int32_t *info);
/* K independent inv(cholesky(.)) on one GPU (:34, for K families at once): A_dev[k] (n[k] x n[k], full symmetric, device) is
 * replaced by its inverse; info[k] as gdca_spd_inverse (info may be NULL).  The small members -- up to MERGE_BLOCKS 128-blocks,
 * which leave most of the chip idle when they run alone -- share launches of the sweep kernel, MERGE of them at a time
 * (gdca_ctx_set_option on ctxs[0]); every result is bit for bit that of gdca_spd_inverse_dev.  K <= 64 distinct contexts of
 * one device (a member's workspace is its context's).  Synchronous.  GDCA_ENOTPD if any member is not positive definite.  Like
 * gdca_spd_inverse_dev, a member whose kappa_1 = ||A||_1 ||inv A||_1 is beyond REFINE_COND gets one Newton-Schulz step. */
gdca_status gdca_spd_inverse_batch_dev(gdca_ctx *const *ctxs, int32_t K, double *const *A_dev, const int32_t *n, int32_t *info);
gdca_status gdca_fn_dev(gdca_ctx *ctx, const double *mJ_dev, int32_t N, int32_t q, double *S_dev);
gdca_status gdca_di_dev(gdca_ctx *ctx, const double *mJ_dev, const double *C_dev, int32_t N, int32_t q, double *S_dev);
gdca_status gdca_apc_dev(gdca_ctx *ctx, double *S_dev, int32_t N);

/* ---- operator level (host pointers): what the DCAUtils-named wrappers bind -------------- */
/* compute_theta's all-pairs identity sum (inside compute_weighted_frequencies, :28) */
gdca_status gdca_pair_identity_sum(gdca_ctx *ctx, const int8_t *Z, int32_t N, int32_t M, uint64_t *out);
/* theta = min(0.5, 0.38*0.32 / mean pair identity) */
gdca_status gdca_compute_theta(gdca_ctx *ctx, const int8_t *Z, int32_t N, int32_t M, double *theta);
/* n_out[k] = 1 + #{l != k : Hamming(k,l) < thresh}  (compute_weights, :28) */
gdca_status gdca_neighbour_counts(gdca_ctx *ctx, const int8_t *Z, int32_t N, int32_t M, int32_t thresh,
                                  int32_t *n_out);
/* compute_weights(Z, q, theta): W[M] = 1/n_k, Meff = the sum of the W[k], exact and rounded once (order-independent).
 * theta < 0 selects :auto.  theta_used / thresh may be NULL. */
gdca_status gdca_compute_weights(gdca_ctx *ctx, const int8_t *Z, int32_t N, int32_t M, double theta,
                                 double *W, double *Meff, double *theta_used, int32_t *thresh);
/* weighted one-/two-site frequencies (compute_weighted_frequencies' accumulation, :28):
 * Pi[n], Pij[n x n] full symmetric.  Every W[k] must lie in [0, 1] and every byte of Z in 1..q (GDCA_EINVAL). */
gdca_status gdca_frequencies(gdca_ctx *ctx, const int8_t *Z, int32_t N, int32_t M, int32_t q,
                             const double *W, double Meff, double *Pi, double *Pij);
/* add_pseudocount(Pi_true, Pij_true, pc, q) (:30) */
gdca_status gdca_add_pseudocount(gdca_ctx *ctx, const double *Pi_true, const double *Pij_true, int32_t N,
                                 int32_t q, double pc, double *Pi, double *Pij);
/* compute_C(Pi, Pij) = Pij - Pi*Pi' (:32, :76) */
gdca_status gdca_covariance(gdca_ctx *ctx, const double *Pi, const double *Pij, int32_t n, double *C);
/* inv(cholesky(C)) (:34): A (n x n, full symmetric) is replaced by its inverse (full
 * symmetric).  *info = 0, or k > 0 when the leading minor of order k is not positive
 * definite (then GDCA_ENOTPD is returned and A is unspecified). */
gdca_status gdca_spd_inverse(gdca_ctx *ctx, double *A, int32_t n, int32_t *info);
/* compute_FN(mJ, q) (:39): S[N x N], diagonal 0 */
gdca_status gdca_fn(gdca_ctx *ctx, const double *mJ, int32_t N, int32_t q, double *S);
/* compute_DI_gauss(mJ, C, q) (:37) */
gdca_status gdca_di(gdca_ctx *ctx, const double *mJ, const double *C, int32_t N, int32_t q, double *S);
/* correct_APC(S) (:42, :78-86), in place */
gdca_status gdca_apc(gdca_ctx *ctx, double *S, int32_t N);

/* ---- host-side utilities around the hot path (plain C++, no GPU): the reference's callers of the path -------- */
/* CPUs this process can really use: hardware threads capped by the cgroup CPU quota (a container may see 256 threads and be given
 * 16 CPUs' worth of time; threads beyond the quota are throttled).  What the reader's and the batch driver's thread counts go by. */
int32_t gdca_host_cpus(void);
typedef struct gdca_fasta gdca_fasta;
/* DCAUtils.read_fasta_alignment(filename, max_gap_fraction) (src/GaussDCA.jl:20): parses a FASTA file (plain or
 * gzip), keeps the columns of the first record that are neither '.' nor lowercase, drops sequences with more
 * than max_gap_fraction '-' among them, maps ACDEFGHIKLMNPQRSTVWY -> 1..20, anything else -> 21.
 * open: parse, report N and M;  copy: fill the caller's N x M int8 matrix;  close: free the handle. */
gdca_status gdca_fasta_open(const char *path, double max_gap_fraction, gdca_fasta **out, int32_t *N, int32_t *M);
gdca_status gdca_fasta_copy(const gdca_fasta *h, int8_t *Z);
/* The parsed matrix in place (N x M column-major, valid until gdca_fasta_close; for callers that hand it straight to
 * gdca_run instead of copying it into an array of their own) and its largest symbol, q = maximum(Z) (src/GaussDCA.jl:25). */
const int8_t *gdca_fasta_data(const gdca_fasta *h);
int32_t gdca_fasta_max_symbol(const gdca_fasta *h);
gdca_status gdca_fasta_close(gdca_fasta *h);
/* DCAUtils.remove_duplicate_sequences(Z) (:21-23): first occurrences, order kept.  Z_out may alias Z.
 * keep_idx (optional, M entries): 1-based indices kept. */
gdca_status gdca_remove_duplicates(const int8_t *Z, int32_t N, int32_t M, int8_t *Z_out, int32_t *keep_idx,
                                   int32_t *M_out);
/* compute_ranking(S, min_separation) (:88-99): (i, j, S[j,i]) for j >= i + min_separation, 1-based, stable sort
 * by score descending.  Outputs hold gdca_ranking_length(N, min_separation) entries. */
int64_t gdca_ranking_length(int32_t N, int32_t min_separation);
gdca_status gdca_ranking(const double *S, int32_t N, int32_t min_separation, int32_t *i_out, int32_t *j_out,
                         double *score_out);
/* The same ranking computed on the device from a score matrix in HBM (N x N column-major, N <= 65535; a stable radix sort on the
 * same key: entry for entry the result of gdca_ranking); outputs are host arrays of gdca_ranking_length entries. */
gdca_status gdca_ranking_dev(gdca_ctx *ctx, const double *S_dev, int32_t N, int32_t min_separation, int32_t *i_out,
                             int32_t *j_out, double *score_out);
/* src/GaussDCA.jl:28-44 in one call: gdca_run followed by compute_ranking on the device -- Z (host) in, the ranking (host
 * arrays of gdca_ranking_length(N, min_separation) entries) out; the N x N score matrix never crosses PCIe. */
gdca_status gdca_run_ranked(gdca_ctx *ctx, const int8_t *Z_host, int32_t N, int32_t M, int32_t q, const gdca_params *p,
                            int32_t min_separation, int32_t *i_out, int32_t *j_out, double *score_out, gdca_stats *st);
/* The same in two halves, for a host thread that keeps two contexts of one GPU busy (gdca_ctx_create_peer): _async uploads Z (the
 * calling thread is held while a pageable Z is staged -- the GPU meanwhile works on what the other context enqueued) and enqueues
 * hot path + ranking without waiting for them; _collect waits and brings the ranking back.  Between the two calls the context
 * accepts nothing else (GDCA_EINVAL), and Z_host may be released as soon as _async has returned. */
gdca_status gdca_run_ranked_async(gdca_ctx *ctx, const int8_t *Z_host, int32_t N, int32_t M, int32_t q, const gdca_params *p,
                                  int32_t min_separation);
gdca_status gdca_run_ranked_collect(gdca_ctx *ctx, int32_t *i_out, int32_t *j_out, double *score_out, gdca_stats *st);
/* K families at once on K contexts of one device (K <= 64; ctxs[0] leads: its options decide the merging, its last_error carries a
 * member's message): uploads, gdca_run_dev_phased -- K front ends, the SPD inverses with the small ones sharing launches of the
 * sweep kernel (options MERGE, MERGE_BLOCKS, ...), K score stages -- and every member's ranking, nothing waited for.  Each member
 * is then collected by itself with gdca_run_ranked_collect, in any order.  What `gdca_cli --batch` runs its small families through. */
gdca_status gdca_run_ranked_phased_async(gdca_ctx *const *ctxs, int32_t K, const int8_t *const *Z_host, const int32_t *N,
                                         const int32_t *M, const int32_t *q, const gdca_params *p, int32_t min_separation);
/* printrank(filename, R) (:67-74): one "%i %i %e" line per entry */
gdca_status gdca_write_rank(const char *path, const int32_t *i, const int32_t *j, const double *score, int64_t len);

/* ---- measurement helpers (bench.py / profiles; not part of the reference surface) ------- */
/* Deterministic "Pfam-like" synthetic family: SplitMix64 streams, integer-threshold draws only, so
 * the same (N, M, q, seed) gives the same bytes everywhere.  Root sequence uniform over 1..q-1; ceil(M/25) cluster
 * centres = root with each site resampled w.p. 1/4; each sequence = a uniformly chosen centre with per-sequence
 * mutation rate from {.02,.05,.1,.2,.3,.5}, then 0-3 gap runs (symbol q) of length U[1, max(2, N/10)].
 * Z is [M][N] (= the N x M column-major matrix).  Needs 2 <= q <= 31. */
gdca_status gdca_synth_family(int32_t N, int32_t M, int32_t q, uint64_t seed, int8_t *Z);
/* Write Z as FASTA (letters of read_fasta_alignment's map, q=21 symbol '-'; headers ">s<k>"); gzip when the path
 * ends in ".gz".  So a reference installation can be fed the same family gdca_synth_family produced. */
gdca_status gdca_write_fasta(const char *path, const int8_t *Z, int32_t N, int32_t M);
/* Dense f64 MFMA issue-rate probe: returns achieved TFLOP/s of a register-resident
 * v_mfma_f64_16x16x4_f64 loop (16 independent accumulators per wave) on every CU: 76-77 on MI355X. */
gdca_status gdca_probe_mfma_f64(gdca_ctx *ctx, int32_t iters, double *tflops);

#ifdef __cplusplus
}
#endif
#endif /* GDCA_H */
