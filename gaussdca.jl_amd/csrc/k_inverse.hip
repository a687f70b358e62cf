// Dense SPD inverse  mJ = inv(cholesky(C))  (reference call site src/GaussDCA.jl:34; there:
// LAPACK dpotrf + dpotri, n^3 flops).
//
// MI355X design: a BLOCK SYMMETRIC SWEEP (block Gauss-Jordan on an SPD matrix) with 128-wide
// pivots.  For pivot block K with D = A_KK (the Schur complement at that point -- the same
// matrix Cholesky would factor, so the positive-definiteness test and the `info` index are the
// same as dpotrf's), P = D^-1, G = A_{.,K}:
//       A_ij <- A_ij - G_i P G_j^T   (i, j != K),   A_{.,K} <- G P,   A_KK <- -P.
// After all pivots A = -C^-1.  Same n^3 flop count as dpotrf+dpotri, but every step is ONE
// launch of ~(n/128)^2/2 identical 128x128x128 tile products over the whole lower triangle:
// no shrinking trailing matrix, no trtri/lauum dependency chains, no tail of tiny launches --
// the shape a 256-CU chip wants.  The matrix stays symmetric throughout, only the lower
// triangle (with full diagonal tiles) is touched.
//
// Tile product: 256 threads = 4 waves in 2 x 2, each wave a 64 x 64 sub-tile as 4 x 4
// v_mfma_f64_16x16x4_f64 accumulators (128 VGPRs).  The f64 C/D fragment is
// col = lane & 15, row = (lane >> 4) + 4 * reg; the MFMA "column" index is mapped to the
// matrix ROW (the contiguous direction of the column-major storage), so every accumulator
// load/store instruction moves 4 columns x 128 contiguous bytes.  Operands go through LDS as
// [k][row] with a 128-byte pad per k-row, which puts the four k-rows a ds_read_b64 touches on
// disjoint bank halves (conflict-free).
#include "gdca_internal.h"

typedef double double4_t __attribute__((ext_vector_type(4)));

#define T 128          // tile edge
#define KC 16          // k-chunk staged per pass
#define LDS_LD (T + 16)  // f64 elements per k-row in LDS (128-byte pad)

// -------------------------------------------------------------------------------------------------
// Pivot: P = inverse of the 128 x 128 SPD block at (k0, k0), by a scalar symmetric sweep held in
// registers (each of 1024 threads owns a 4 x 4 patch; the pivot column is broadcast through LDS).
// Writes P (128 x 128, column-major) and A_KK <- -P.  Non-positive pivot -> sc->info.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_pivot(double *__restrict__ A, size_t ld, int k0, double *__restrict__ P,
                                                 gdca_dev_scalars *sc, int n_real)
{
    __shared__ __attribute__((aligned(16))) double g[2][T];
    const int tid = threadIdx.x;
    const int tr = tid & 31, tc = tid >> 5;
    const int r0 = tr * 4, c0 = tc * 4;
    double D[4][4];
    double *Akk = A + (size_t)k0 + (size_t)k0 * ld;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int r = r0 + a, c = c0 + b;
            const int rr = r >= c ? r : c, cc = r >= c ? c : r;  // lower triangle is authoritative
            D[a][b] = Akk[(size_t)rr + (size_t)cc * ld];
        }
    int bad = 0;
    for (int j = 0; j < T; ++j) {
        double *gj = g[j & 1];
        if (tc == (j >> 2)) {
            const int jb = j & 3;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const double v = (jb == 0) ? D[a][0] : (jb == 1) ? D[a][1] : (jb == 2) ? D[a][2] : D[a][3];
                gj[r0 + a] = v;
            }
        }
        __syncthreads();
        const double d = gj[j];
        if (!(d > 0.0) && bad == 0) bad = j + 1;
        const double p = 1.0 / d;
        double gr[4], gc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) gr[a] = gj[r0 + a];
#pragma unroll
        for (int b = 0; b < 4; ++b) gc[b] = gj[c0 + b];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int r = r0 + a, c = c0 + b;
                const double t = gr[a] * gc[b];  // commutative: keeps the block bitwise symmetric
                double v = fma(-t, p, D[a][b]);
                if (r == j) v = (c == j) ? -p : gc[b] * p;
                else if (c == j) v = gr[a] * p;
                D[a][b] = v;
            }
    }
    // D = -inverse.  P = -D;  A_KK = D (full tile)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int r = r0 + a, c = c0 + b;
            P[(size_t)r + (size_t)c * T] = -D[a][b];
            Akk[(size_t)r + (size_t)c * ld] = D[a][b];
        }
    if (tid == 0 && bad != 0 && (k0 + bad) <= n_real) {
        if (sc->info == 0) sc->info = k0 + bad;
    }
}

// -------------------------------------------------------------------------------------------------
// 128 x 128 x 128 tile product on f64 MFMA, shared by the panel and the update kernels:
//   acc(r, c) += sum_k Gsrc(r, k) * Hsrc(c, k)
// Gsrc(r,k) = gsrc[r + k*gld] (or gsrc[k + r*gld] when GT);  Hsrc(c,k) = hsrc[c + k*hld].
// -------------------------------------------------------------------------------------------------
struct StageRegs {
    double g[8], h[8];
};

template <bool GT>
__device__ __forceinline__ void stage_load(StageRegs &R, const double *__restrict__ gsrc, size_t gld,
                                           const double *__restrict__ hsrc, size_t hld, int kc, int tid)
{
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int idx = tid + 256 * u;
        if (!GT) {
            const int kk = idx >> 6, r2 = idx & 63;
            const double2 v = *reinterpret_cast<const double2 *>(gsrc + (size_t)(r2 * 2) + (size_t)(kc + kk) * gld);
            R.g[2 * u] = v.x;
            R.g[2 * u + 1] = v.y;
        } else {
            const int r = idx >> 3, k2 = idx & 7;
            const double2 v = *reinterpret_cast<const double2 *>(gsrc + (size_t)(kc + k2 * 2) + (size_t)r * gld);
            R.g[2 * u] = v.x;
            R.g[2 * u + 1] = v.y;
        }
        const int kk = idx >> 6, c2 = idx & 63;
        const double2 w = *reinterpret_cast<const double2 *>(hsrc + (size_t)(c2 * 2) + (size_t)(kc + kk) * hld);
        R.h[2 * u] = w.x;
        R.h[2 * u + 1] = w.y;
    }
}

template <bool GT>
__device__ __forceinline__ void stage_store(const StageRegs &R, double (*Gs)[LDS_LD], double (*Hs)[LDS_LD], int tid)
{
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int idx = tid + 256 * u;
        if (!GT) {
            const int kk = idx >> 6, r2 = idx & 63;
            *reinterpret_cast<double2 *>(&Gs[kk][r2 * 2]) = make_double2(R.g[2 * u], R.g[2 * u + 1]);
        } else {
            const int r = idx >> 3, k2 = idx & 7;
            Gs[k2 * 2][r] = R.g[2 * u];
            Gs[k2 * 2 + 1][r] = R.g[2 * u + 1];
        }
        const int kk = idx >> 6, c2 = idx & 63;
        *reinterpret_cast<double2 *>(&Hs[kk][c2 * 2]) = make_double2(R.h[2 * u], R.h[2 * u + 1]);
    }
}

__device__ __forceinline__ void chunk_mma(double4_t (&acc)[4][4], double (*Gs)[LDS_LD], double (*Hs)[LDS_LD], int wr,
                                          int wc, int lane)
{
    const int l15 = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int k4 = 0; k4 < KC; k4 += 4) {
        double a[4], b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a[t] = Hs[k4 + lq][wc * 64 + t * 16 + l15];
            b[t] = Gs[k4 + lq][wr * 64 + t * 16 + l15];
        }
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
    }
}

template <bool GT>
__device__ __forceinline__ void tile_product(double4_t (&acc)[4][4], const double *__restrict__ gsrc, size_t gld,
                                             const double *__restrict__ hsrc, size_t hld, double (*Gs)[LDS_LD],
                                             double (*Hs)[LDS_LD], double *__restrict__ gcopy, size_t gcopy_ld)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv & 1, wc = wv >> 1;
    StageRegs R;
    stage_load<GT>(R, gsrc, gld, hsrc, hld, 0, tid);
    for (int kc = 0; kc < T; kc += KC) {
        __syncthreads();  // previous chunk's LDS reads are done
        stage_store<GT>(R, Gs, Hs, tid);
        if (gcopy) {
            // keep an untransposed copy of the G panel: gcopy[r + k * gcopy_ld]
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = tid + 256 * u;
                if (!GT) {
                    const int kk = idx >> 6, r2 = idx & 63;
                    *reinterpret_cast<double2 *>(gcopy + (size_t)(r2 * 2) + (size_t)(kc + kk) * gcopy_ld) =
                        make_double2(R.g[2 * u], R.g[2 * u + 1]);
                } else {
                    const int r = idx >> 3, k2 = idx & 7;
                    gcopy[(size_t)r + (size_t)(kc + k2 * 2) * gcopy_ld] = R.g[2 * u];
                    gcopy[(size_t)r + (size_t)(kc + k2 * 2 + 1) * gcopy_ld] = R.g[2 * u + 1];
                }
            }
        }
        __syncthreads();
        if (kc + KC < T) stage_load<GT>(R, gsrc, gld, hsrc, hld, kc + KC, tid);
        chunk_mma(acc, Gs, Hs, wr, wc, lane);
    }
}

// Panel: for every row block i != k:  G_i = column block k of the symmetric matrix (read from the
// lower triangle: A[i,k] for i > k, A[k,i]^T for i < k);  GP = G_i P;  writes
//   Gbuf[i] = G_i,  Hbuf[i] = -GP,  and the new column block  A[i,k] = GP  (A[k,i] = GP^T for i < k).
__global__ __launch_bounds__(256, 2) void k_panel(double *__restrict__ A, size_t ld, int kblk,
                                                   const double *__restrict__ P, double *__restrict__ Gbuf,
                                                   double *__restrict__ Hbuf, size_t pld)
{
    __shared__ __attribute__((aligned(16))) double Gs[KC][LDS_LD];
    __shared__ __attribute__((aligned(16))) double Hs[KC][LDS_LD];
    int i = blockIdx.x;
    if (i >= kblk) ++i;  // skip the pivot block itself
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv & 1, wc = wv >> 1, l15 = lane & 15, lq = lane >> 4;
    double4_t acc[4][4];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = (double4_t){0.0, 0.0, 0.0, 0.0};

    double *gcopy = Gbuf + (size_t)i * T;
    if (i > kblk) {
        const double *gsrc = A + (size_t)i * T + (size_t)kblk * T * ld;
        tile_product<false>(acc, gsrc, ld, P, T, Gs, Hs, gcopy, pld);
    } else {
        const double *gsrc = A + (size_t)kblk * T + (size_t)i * T * ld;
        tile_product<true>(acc, gsrc, ld, P, T, Gs, Hs, gcopy, pld);
    }
    // acc[tm][tn][reg] = GP(r, c):  r = wr*64 + tn*16 + l15,  c = wc*64 + tm*16 + lq + 4*reg
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = wr * 64 + tn * 16 + l15;
                const int c = wc * 64 + tm * 16 + lq + 4 * reg;
                const double v = acc[tm][tn][reg];
                Hbuf[(size_t)i * T + r + (size_t)c * pld] = -v;
                if (i > kblk)
                    A[(size_t)i * T + r + (size_t)(kblk * T + c) * ld] = v;
                else
                    A[(size_t)kblk * T + c + (size_t)(i * T + r) * ld] = v;
            }
}

// Update: every lower-triangle tile (I >= J) with I, J != k:  A_IJ += G_I * H_J^T   (H = -G P).
__global__ __launch_bounds__(256, 2) void k_sweep_update(double *__restrict__ A, size_t ld, int kblk, int nblk,
                                                          const double *__restrict__ Gbuf,
                                                          const double *__restrict__ Hbuf, size_t pld)
{
    __shared__ __attribute__((aligned(16))) double Gs[KC][LDS_LD];
    __shared__ __attribute__((aligned(16))) double Hs[KC][LDS_LD];
    // decode blockIdx.x -> (ii >= jj) over the nblk-1 active block indices
    const int t = blockIdx.x;
    int ii = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((long long)ii * (ii + 1) / 2 > t) --ii;
    while ((long long)(ii + 1) * (ii + 2) / 2 <= t) ++ii;
    const int jj = t - (int)((long long)ii * (ii + 1) / 2);
    const int I = ii + (ii >= kblk ? 1 : 0), J = jj + (jj >= kblk ? 1 : 0);
    (void)nblk;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv & 1, wc = wv >> 1, l15 = lane & 15, lq = lane >> 4;
    double *At = A + (size_t)I * T + (size_t)J * T * ld;
    double4_t acc[4][4];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = wr * 64 + tn * 16 + l15;
                const int c = wc * 64 + tm * 16 + lq + 4 * reg;
                acc[tm][tn][reg] = At[(size_t)r + (size_t)c * ld];
            }
    tile_product<false>(acc, Gbuf + (size_t)I * T, pld, Hbuf + (size_t)J * T, pld, Gs, Hs, nullptr, 0);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = wr * 64 + tn * 16 + l15;
                const int c = wc * 64 + tm * 16 + lq + 4 * reg;
                At[(size_t)r + (size_t)c * ld] = acc[tm][tn][reg];
            }
}

void gdca_launch_spd_inverse(hipStream_t s, double *A, int n_pad, const gdca_inverse_ws &ws, gdca_dev_scalars *sc,
                             int n_real, hipEvent_t *upd_ev, int max_ev, int *n_upd_launch)
{
    const int nblk = n_pad / T;
    const size_t ld = (size_t)n_pad;
    int nl = 0;
    for (int k = 0; k < nblk; ++k) {
        hipLaunchKernelGGL(k_pivot, dim3(1), dim3(1024), 0, s, A, ld, k * T, ws.P, sc, n_real);
        if (nblk > 1) {
            hipLaunchKernelGGL(k_panel, dim3(nblk - 1), dim3(256), 0, s, A, ld, k, ws.P, ws.G, ws.H, ld);
            const int m = nblk - 1;
            const unsigned ntile = (unsigned)((long long)m * (m + 1) / 2);
            if (upd_ev && 2 * nl + 1 < max_ev) (void)hipEventRecord(upd_ev[2 * nl], s);
            hipLaunchKernelGGL(k_sweep_update, dim3(ntile), dim3(256), 0, s, A, ld, k, nblk, ws.G, ws.H, ld);
            if (upd_ev && 2 * nl + 1 < max_ev) (void)hipEventRecord(upd_ev[2 * nl + 1], s);
            ++nl;
        }
    }
    if (n_upd_launch) *n_upd_launch = nl;
}

// -------------------------------------------------------------------------------------------------
// f64 MFMA issue-rate probe (register-resident, 8 independent accumulators per wave).
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_probe_mfma_f64(double *out, int iters)
{
    double4_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double sacc = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) sacc += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = sacc;
}

void gdca_launch_probe_mfma_f64(hipStream_t s, double *out, int iters, int blocks)
{
    hipLaunchKernelGGL(k_probe_mfma_f64, dim3(blocks), dim3(256), 0, s, out, iters);
}
