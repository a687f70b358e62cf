// Internal declarations shared by the HIP translation units of libgdca.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gdca.h"

#define GDCA_TILE 128     // tile edge of the SPD-inverse kernels (f64 elements)
#define GDCA_HTILE 128    // sequences per side of a Hamming tile
#define GDCA_MAXQ 31
#define GDCA_STAMPS 20

// Scalars that live in HBM so the whole pipeline can be enqueued without a host round trip.
struct gdca_dev_scalars {
    double theta;
    double Meff;
    unsigned long long pair_sum;
    int thresh;
    int info;
    int bad_symbol;  // bit 0: a byte of Z is outside 1..q; bit 1: a caller-given weight is outside [0, 1] (GDCA_EINVAL)
    int di_noconv;   // number of site pairs whose tridiagonal QL iteration did not converge (DI score)
    int ham_mode;    // all-pairs Hamming kernel chosen for this family: 0 = exact distances, 1 = three-plane lower bound + refinement, 2 = the bit-count
                     // lower bound on the fp4 matrix pipe + refinement (k_hamming_fp4.hip)
    int ham_cand;    // candidate pairs (bound below the threshold) in the sampled tiles of k_hamming_probe
    unsigned long long ham_ncand;  // pairs the bound form has put (or tried to put) into its candidate list: beyond the list's capacity the exact form counts
    unsigned long long sweep_cycles, sweep_ticks;  // k_sweep, summed over its workgroups: shader-clock cycles (s_memtime) and 100 MHz ticks they ran for
    double inv_norm1;  // ||inverse||_1 as the sweep left it (0: not measured)
    double ns_resid;     // max |I - X0 C| seen by the Newton-Schulz step (0: no step)
    double mat_norm1;  // ||C||_1: measured on the covariance before the sweep where the cheap bound leaves the question open (fused path), or on the
                       // caller's matrix (operator-level inverse); 0: not measured
    double pi_max;     // max over the n single-site frequencies with pseudocount (k_pi_finalize): ||C||_1 <= 2 N pi_max
    // device time stamps (100 MHz wall clock) of a run whose kernels were issued as batched grids (gdca_run_dev_phased: no HIP events
    // between them -- an event per member and stage was most of the host's work for a batch): slot = the index of the event a
    // launch of its own would have recorded (gdca_api.hip: EV_*)
    unsigned long long stamp[GDCA_STAMPS];
    int ham_cand2;   // pairs of the sampled tiles the fp4 form would list (D < 3 thresh)
    int pad_;
};

// Tuning switches of one context (gdca_ctx_set_option): initialised from the GDCA_* environment variables when the context is
// created, never read from the environment afterwards -- two contexts of one process may differ, and a switch changed through
// gdca_ctx_set_option takes effect at the next call on that context.  -1 = "the measured rule" where a rule exists.
struct gdca_tuning {
    int group;              // GDCA_GROUP: pivot blocks per group of the sweep, 1..4
    int ramp;               // GDCA_RAMP: 0 = uniform groups, the remainder last
    int ragged;             // GDCA_RAGGED: 0 = sweep the matrix padded to 128 instead of 16
    int rem_tail;           // GDCA_REM_TAIL: remainder tiles of update p listed after panel(p+1)
    int panel_halves;       // GDCA_PANEL_HALVES: 1 = two 128 x 64 panel items per pivot block and row, 0 = one 128 x 128
    int slab;               // GDCA_SLAB: 0 = panel and tile items between single blocks instead of the fused row slabs
    int ring;               // GDCA_RING: panel / Pg slots between single blocks (2..8)
    int mcus;               // GDCA_MCUS: compute units elected for the pivot chain (1..16)
    int mcu_solo;           // GDCA_MCU_SOLO: 1 = one chain worker per elected compute unit (its second workgroup leaves; MCUS then counts up to 32), 0 = two, -1 = the measured rule (groups of two and three)
    int sweep_debug;        // GDCA_SWEEP_DEBUG (tests): bit 0 = XCC 0 stays out of the election, bit 1 = nobody is elected, bit 5 = the watchdog ends the first attempt of every inverse
    long sweep_timeout_ms;  // GDCA_SWEEP_TIMEOUT_MS: bound of one dependency wait; 0 = scaled with the problem (>= 4 s)
    int sweep_retries;      // GDCA_SWEEP_RETRIES: attempts an inverse gets after its launch was ended by the watchdog (default 2; 0 = none: GDCA_EHIP at once)
    int tally_tj;           // GDCA_TALLY_TJ: 32 = the wide pair-tally form
    int hamming_mode;       // GDCA_HAMMING_MODE: -1 = probe, 0 = full (exact five-plane distances), 1 = bound (three planes + refinement), 2 = mfma (bit counts on the fp4 matrix pipe + refinement)
    int force_fallback;     // GDCA_FORCE_FALLBACK: 1 = the independent byte-compare Hamming kernel
    int merge;              // GDCA_MERGE: families one merged sweep launch may carry in gdca_run_dev_phased (1 = never merge)
    int merge_blocks;       // GDCA_MERGE_BLOCKS: largest member of a merged launch, in 128-blocks
    int merge_mcus;         // GDCA_MERGE_MCUS: chain compute units per member of a merged launch
    int merge_group;        // GDCA_MERGE_GROUP: pivot blocks per group of a member of a merged launch, 1..4
    int merge_tiles;        // GDCA_MERGE_TILES: a merged launch is closed once its members hold this many tiles per update step
    int phased_fronts;      // GDCA_PHASED_FRONTS: 1 = the front ends of a phase batch run side by side on the members' own streams (default), 0 = one after the other on the leader's
    int phased_grids;       // GDCA_PHASED_GRIDS: the kernels of a phase batch's front ends and score stages go out as ONE grid per kernel kind carrying all members of a group: 1 .. 8 = that many groups of members side by side (front ends; the score stages are always one group), -1 = by the batch's work (default: one group for small families, four for big ones), 0 = one launch per member and kernel
    int phased_streams;     // GDCA_PHASED_STREAMS: streams the side-by-side front ends and score stages of a phase batch are spread over (the first members' own; default 4 = the hardware queues)
    int refine;             // GDCA_REFINE: -1 = one Newton-Schulz step where the inverse looks ill-conditioned (auto), 0 = never, 1 = always
    double refine_cond;     // GDCA_REFINE_COND: the threshold of auto: the a-priori bound of cond_2(C) first, beyond it kappa_1 = ||C||_1 ||X||_1
    int cholesky;           // GDCA_CHOLESKY: the blocked dpotrf + dpotri fallback: 0 = never, 1 = where the sweep gave up (default), 2 = always
    char sweep_trace[256];  // GDCA_SWEEP_TRACE: file the in-kernel trace of the next inverse is written to ("" = off)
};
void gdca_tuning_from_env(gdca_tuning *t);
// key: the variable's name with or without the GDCA_ prefix, any case.  false: unknown key or unusable value.
bool gdca_tuning_set(gdca_tuning *t, const char *key, const char *value);

// ---- k_theta.hip -------------------------------------------------------------------------
void gdca_launch_transpose_i8(hipStream_t s, const int8_t *Z, int8_t *Zt, int N, int M);
// cnt: uint32 [N][32], zeroed by the caller
void gdca_launch_column_hist(hipStream_t s, const int8_t *Z, uint32_t *cnt, int N, int M);
// theta_in < 0: theta = :auto from cnt; else theta = theta_in.  Writes theta, thresh, pair_sum.
void gdca_launch_theta_finalize(hipStream_t s, const uint32_t *cnt, int N, int M, double theta_in,
                                gdca_dev_scalars *sc);
// sets sc->thresh directly (operator-level neighbour counts with a caller-given threshold)
void gdca_launch_set_thresh(hipStream_t s, gdca_dev_scalars *sc, int thresh);

// ---- k_hamming.hip -----------------------------------------------------------------------
// bit-plane image: uint32 [Mt][5][NW][128], Mt = ceil(M/128), NW = ceil(N/32)
size_t gdca_bitplane_bytes(int N, int M);
// q: largest legal symbol (bytes outside 1..q set sc->bad_symbol)
void gdca_launch_bitplane_pack(hipStream_t s, const int8_t *Z, uint32_t *Zb, int N, int M, int q,
                               gdca_dev_scalars *sc);
// cnt: int32 [Mt*128], zeroed by the caller; adds #{l != k: d(k,l) < sc->thresh}
// force: -1 = decide per family from a sample of tiles, 0 = the exact form, 1 = the lower bound with refinement
size_t gdca_hamming_cand_cap(int M);  // pairs the bound forms' candidate list holds (8 bytes each)
// ---- k_hamming_fp4.hip: the bit-count lower bound on the fp4 matrix pipe (sc->ham_mode == 2) ----
size_t gdca_fp4_image_bytes(int N, int M);
void gdca_launch_hamming_fp4_probe(hipStream_t s, const uint32_t *Zb, int N, int M, int nprobe, gdca_dev_scalars *sc);
void gdca_launch_hamming_fp4(hipStream_t s, const uint32_t *Zb, void *img, int N, int M, gdca_dev_scalars *sc, void *cand_list, unsigned cap);
void gdca_launch_hamming(hipStream_t s, const uint32_t *Zb, const int8_t *Z, int32_t *cnt, int N, int M, gdca_dev_scalars *sc, int force,
                         void *cand_list, void *fp4_img);
// the same counts by an independent plain byte-compare kernel straight from Z (GDCA_FORCE_FALLBACK; overwrites cnt[0..M-1])
void gdca_launch_hamming_fallback(hipStream_t s, const int8_t *Z, int32_t *cnt, int N, int M, const gdca_dev_scalars *sc);
// n_out[k] = 1 + cnt[k]; W[k] = 1/n_k; Wfix[k] = rint(W[k] * 2^fix_shift)
void gdca_launch_weights(hipStream_t s, const int32_t *cnt, int M, int fix_shift, int32_t *n_out, double *W,
                         unsigned long long *Wfix);
// Meff = the exact sum of the weights, rounded once (one workgroup; order-independent)
void gdca_launch_meff(hipStream_t s, const double *W, int M, gdca_dev_scalars *sc);
// Wfix from caller-given W (operator-level gdca_frequencies)
void gdca_launch_fix_weights(hipStream_t s, const double *W, int M, int fix_shift, unsigned long long *Wfix,
                             gdca_dev_scalars *sc);
int gdca_fix_shift(int M);

// ---- k_tally.hip -------------------------------------------------------------------------
// Pifix: u64 [N][32] zeroed by the caller; adds sum_k Wfix[k] [Z[i,k]==a] at [i][a-1]
void gdca_launch_pi_tally(hipStream_t s, const int8_t *Z, const unsigned long long *Wfix,
                          unsigned long long *Pifix, int N, int M, int q, gdca_dev_scalars *sc);
// Pi_true[i*s+a] = Pifix * 2^-shift / Meff;  Pi_pc = (1-pc) Pi_true + pc/q
void gdca_launch_pi_finalize(hipStream_t s, const unsigned long long *Pifix, int N, int q, int fix_shift,
                             const double *Meff_dev, double pc, double *Pi_true, double *Pi_pc, double *pi_max = nullptr);
// ||C||_1 of the covariance at C (full symmetric, ld) into sc->mat_norm1 -- only where 2 N pi_max q^2 / pc, the bound of cond(C) that
// costs nothing, exceeds cond_limit (or `always`): otherwise every workgroup leaves at once
void gdca_launch_cov_norm1(hipStream_t s, const double *C, size_t ld, int N, int q, double pc, double cond_limit, gdca_dev_scalars *sc,
                           int always);
// Pair tallies.  mode 0: out = Pij_true (full symmetric, ld);  mode 1: out = C =
// add_pseudocount + compute_C fused (full symmetric, ld).  Pi_pc used by mode 1 only.
// Zc: the alignment regrouped as [ceil(N/TJ)][M][TJ] (gdca_launch_colblock), TJ = gdca_tally_tj(q).
int gdca_tally_tj(int q, int tj_wanted);
void gdca_launch_colblock(hipStream_t s, const int8_t *Z, int8_t *Zc, int N, int M, int TJ);
void gdca_launch_pair_tally(hipStream_t s, const int8_t *Zc, const int8_t *Zt, const unsigned long long *Wfix,
                            int N, int M, int q, int fix_shift, const double *Meff_dev, double pc,
                            const double *Pi_pc, int mode, double *out, size_t ld, int TJ);

// ---- k_elementwise.hip ---------------------------------------------------------------------
void gdca_launch_add_pseudocount(hipStream_t s, const double *Pi_true, const double *Pij_true, int N, int q,
                                 double pc, double *Pi, double *Pij);
void gdca_launch_covariance(hipStream_t s, const double *Pi, const double *Pij, int n, double *C);
// A[n_pad x n_pad] (ld = n_pad): rows/cols >= n become identity
void gdca_launch_pad_identity(hipStream_t s, double *A, int n, int n_pad);
// copy a dense n x n (ld_src) into the padded buffer (ld_dst) / back, with optional negate+mirror
void gdca_launch_copy_in(hipStream_t s, const double *src, int n, double *dst, int n_pad);
// same with the real block negated (mJ -> the "-mJ" image the score kernels read)
void gdca_launch_copy_in_neg(hipStream_t s, const double *src, int n, double *dst, int n_pad);
// dst (n x n, ld n) = full symmetric  -lower(A)  (A holds -inverse in its lower triangle)
void gdca_launch_copy_out_neg_sym(hipStream_t s, const double *A, int n_pad, double *dst, int n);
// *host_mapped (pinned host memory, the device's address of it) = *sc, system-scope visible when the kernel has completed
void gdca_launch_publish_scalars(hipStream_t s, const gdca_dev_scalars *sc, gdca_dev_scalars *host_mapped);
// D[i] (s x s, packed) = diagonal block i of C (ld)
void gdca_launch_save_diag_blocks(hipStream_t s, const double *C, size_t ld, int N, int sdim, double *D);

// ---- k_inverse.hip -------------------------------------------------------------------------
struct gdca_inverse_ws {
    double *G[8];      // n_pad x 128 panels: [4 (p & 1) + w] = column block w of pivot group p (double-buffered by group parity)
    double *H[8];      // n_pad x 128 panels, -G * Pg
    double *P;         // 128 x 128 inverse of one pivot block
    double *Sg[2];     // 512 x 512 dense scratch copies of a group's diagonal super-block (ping-pong)
    double *Pg[2];     // 512 x 512: inverse of a group's diagonal super-block, by group parity
    unsigned *flags;   // dependency flags of the sweep (zeroed per inverse by the launcher)
    size_t flags_bytes;
    int *item0_host;   // pinned: first work item of every group's sequence in the main list and in the M list (2 x (n_pad / 128 + 2) entries)
    const int *item0_host_dev;  // item0_host as the device addresses it (nullptr: the table goes through hipMemcpyAsync)
    int *item0_dev;
    int update_cus;    // compute units of the device
};
// One inverse: in place on A (n_pad x n_pad, ld = n_pad, lower triangle + full diagonal tiles authoritative): A <- -inverse(A) by
// the block symmetric sweep.  sc->info gets the 1-based index of the first non-positive pivot, if any (INT_MIN: the kernel's
// watchdog abandoned a dependency wait).
struct gdca_inverse_job {
    double *A;
    int n_pad, n_real;
    gdca_inverse_ws ws;
    gdca_dev_scalars *sc;
    const gdca_tuning *tune;
    bool doomed;  // tests (option SWEEP_DEBUG bit 5): this launch's watchdog fires at its first unsatisfied wait
};
size_t gdca_inverse_flag_bytes(int n_pad);
// ONE persistent launch on stream s; upd_ev (optional, 2 events) are recorded around the launch.
void gdca_launch_spd_inverse(hipStream_t s, const gdca_inverse_job &job, hipEvent_t *upd_ev, int max_upd_ev, int *n_upd_launch,
                             double *upd_flops);
// K <= gdca_inverse_max_merge() independent inverses (single-block schedules: small matrices) carried by ONE persistent launch;
// every member's arithmetic is that of a launch of its own.  upd_flops: K entries.
int gdca_inverse_max_merge(void);
void gdca_launch_spd_inverse_merged(hipStream_t s, const gdca_inverse_job *jobs, int K, hipEvent_t *upd_ev, int max_upd_ev,
                                    double *upd_flops);
// ---- k_rank.hip --------------------------------------------------------------------------
// compute_ranking (src/GaussDCA.jl:88-99) of S_dev (N x N column-major): `len` = gdca_ranking_length(N, sep) > 0 entries, sorted,
// left in three arrays inside the workspace `ws` (gdca_ranking_ws_bytes(len) bytes)
size_t gdca_ranking_ws_bytes(long long len);
void gdca_launch_ranking(hipStream_t s, const double *S_dev, int N, int sep, long long len, void *ws, int32_t **ii, int32_t **jj, double **sc);
void gdca_launch_probe_mfma_f64(hipStream_t s, double *out, int iters, int blocks);
// The reference's own way (dpotrf + dpotri) as a fallback for matrices the sweep cannot handle: C2 (n_pad x n_pad, identity
// padding; overwritten by its Cholesky factor), U and Tm: n_pad x n_pad workspaces, Wd: (n_pad / 128) tiles of 128 x 128; Aout
// receives -inverse in its lower block triangle (the sweep's storage); sc->info = dpotrf's index of a non-positive pivot
void gdca_launch_cholesky_inverse(hipStream_t s, double *C2, double *U, double *Tm, double *Wd, double *Aout, int n_pad, int n_real,
                                  gdca_dev_scalars *sc);
// *out = ||X||_1 of the symmetric matrix whose lower block triangle is A (= -X, the sweep's storage; first n rows / columns);
// colsum_ws: n_pad doubles
void gdca_launch_inverse_norm1(hipStream_t s, const double *A, int n_pad, int n, double *colsum_ws, double *out);
// *out = max_i |A(i, i)|, i < n
// *out = ||C||_1 of a plain n x n matrix
void gdca_launch_matrix_norm1(hipStream_t s, const double *C, size_t ld, int n, double *colsum_ws, double *out);
// one Newton-Schulz step on the sweep's result: A (-X0 lower block triangle -> -X1), C2 = the matrix that was inverted (full
// symmetric, n_pad x n_pad, identity padding), B0 and Rt: n_pad x n_pad workspaces
// *resid (optional): max |I - X0 C|, the residual the step squares -- below one or the step cannot have converged
void gdca_launch_newton_schulz(hipStream_t s, double *A, const double *C2, double *B0, double *Rt, int n_pad, double *resid);

// ---- k_score.hip ---------------------------------------------------------------------------
// S (N x N) from the lower triangle of A = -mJ (ld).  Diagonal 0.
void gdca_launch_fn(hipStream_t s, const double *A, size_t ld, int N, int sdim, double *S, int ncu = 0);
// Ld[i] = chol(D[i]) lower, packed s x s
void gdca_launch_diag_chol(hipStream_t s, const double *D, int N, int sdim, double *Ld);
// Tws: workspace of gdca_di_ws_bytes(N, sdim) bytes (tridiagonals of all site pairs)
size_t gdca_di_ws_bytes(int N, int sdim);
void gdca_launch_di(hipStream_t s, const double *A, size_t ld, const double *Ld, int N, int sdim, double *S,
                    double *Tws, gdca_dev_scalars *sc);
void gdca_launch_apc(hipStream_t s, double *S, int N, double *rowsum_ws);
