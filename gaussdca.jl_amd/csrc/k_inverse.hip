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
#include <algorithm>

#include "gdca_internal.h"

typedef double double4_t __attribute__((ext_vector_type(4)));

#define T 128          // tile edge
#define KC 16          // k-chunk staged per pass
#define LDS_LD (T + 16)  // f64 elements per k-row in LDS (128-byte pad)

// -------------------------------------------------------------------------------------------------
// Pivot: P = inverse of a 128 x 128 SPD block by a BLOCKED symmetric sweep, one workgroup, the block's lower
// triangle resident in MFMA accumulators for the whole kernel.
//
// The block is cut into 8 x 8 micro-blocks of 16 x 16; the 36 lower-triangular tiles live in the accumulators of
// 12 waves (3 tiles each; waves 0..7 own one diagonal tile and two others).  Sweeping micro-block K
//       D_ij <- D_ij - G_i Pm G_j^T   (i, j != K),   D_iK <- G_i Pm,   D_KK <- -Pm,      G = D_{.,K},  Pm = D_KK^-1
// is, for EVERY tile, four v_mfma_f64_16x16x4_f64 on two operand images in LDS:
//       Gs = the old column block K (128 x 16), with the rows of micro-block K replaced by -I
//       Ns = -(G Pm)               (128 x 16), with the rows of micro-block K replaced by +Pm
//       tile(rb, cb) <- [rb == K or cb == K ? 0 : tile] + Gs[rb] Ns[cb]^T
// (the same -1 / -p device the scalar sweep uses for its pivot row and column), so the three kinds of tiles need
// no special code.  The only serial part is Pm = (16 x 16 diagonal tile)^-1: a 16-step scalar sweep inside ONE
// wave's registers (the tile is already there in accumulator layout; the pivot column travels by cross-lane
// shuffles, no LDS round trip, no barrier), and it runs one micro-block AHEAD: in the update phase of micro-block
// K the owner of tile (K+1, K+1) updates that tile first and inverts it at once, beside the other waves' MFMAs.
// Per micro-block: three barriers, 12 MFMAs per wave -- 24 barrier-separated phases for the whole block instead
// of the 128 of the element-wise sweep (measured 84 us there).
//
// Operand images are [kk][row] with a swizzled row offset (pv_off): the MFMA operand reads (16 consecutive rows of
// two adjacent kk per 32 lanes) and the transposed stores of the tiles left of the diagonal (16 different kk, one
// row) are both (nearly) conflict-free.
// 1/d is v_rcp_f64 + two Newton steps.  A non-positive pivot (the same test dpotrf makes) is reported through
// sc->info.  Writes P (full symmetric, ld = pld) and A_KK <- -P (full tile).
// -------------------------------------------------------------------------------------------------
#define MB 16                      // micro-block edge
#define NMB (T / MB)               // micro-blocks per side
#define PIVOT_THREADS 768          // 12 waves
#define PV_ROW 160                 // doubles per kk-row of an operand image

__device__ __forceinline__ int pv_off(int kk)
{
    return kk * PV_ROW + 16 * (kk & 1) + 2 * (kk >> 1);
}

__device__ __forceinline__ double shfl_f64(double v, int src_lane)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(b & 0xffffffffll));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Broadcast, inside every row of 16 lanes, the value lane J of that row holds (DPP row_newbcast: VALU speed, no LDS).
template <int J>
__device__ __forceinline__ double row_bcast_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), 0x150 + J, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x150 + J, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <int L>
__device__ __forceinline__ double read_lane_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), L);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), L);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// One step of the 16 x 16 sweep on v[reg] = D[i = lane & 15][j = (lane >> 4) + 4 reg] (full storage).  The pivot
// column reaches lane (i, .) by one cross-row shuffle (ds_bpermute, address colsrc[JJ & 3] = lane (JJ & 3) * 16 + i),
// the pivot row's entries D[JJ][j] by DPP broadcasts inside the lane's own row of 16, the pivot itself by v_readlane.
template <int JJ>
struct MicroStep {
    static __device__ __forceinline__ void run(double (&v)[4], int l15, int lq, const int (&colsrc)[4], int index_base,
                                               int *badj)
    {
        const double col = v[JJ >> 2];
        const double ui = shfl_f64(col, colsrc[JJ & 3]);
        const double d = read_lane_f64<(JJ & 3) * 16 + JJ>(col);
        if (!(d > 0.0) && *badj == 0) *badj = index_base + JJ + 1;
        double p = __builtin_amdgcn_rcp(d);
        p = fma(p, fma(-d, p, 1.0), p);
        p = fma(p, fma(-d, p, 1.0), p);
        const double u = (l15 == JJ) ? -1.0 : ui;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int j = lq + 4 * reg;
            const double uj = row_bcast_f64<JJ>(v[reg]);  // D[JJ][j]
            const double w = (j == JJ) ? -p : uj * p;
            const double base = (l15 == JJ || j == JJ) ? 0.0 : v[reg];
            v[reg] = fma(-u, w, base);
        }
        if constexpr (JJ + 1 < MB) MicroStep<JJ + 1>::run(v, l15, lq, colsrc, index_base, badj);
    }
};

// Inverse of the 16 x 16 SPD tile held by one wave as v[reg] = D[i = lane & 15][j = (lane >> 4) + 4 reg] (lower
// triangle authoritative): on return v = -D^-1 (full storage, equal to its transpose up to rounding).
__device__ __forceinline__ void micro_pivot(double (&v)[4], int lane, int index_base, int *badj)
{
    const int l15 = lane & 15, lq = lane >> 4;
    {
        // upper triangle := mirror of the lower: element (j, i) sits in lane (i & 3) * 16 + j, register i >> 2
        double s[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) s[reg] = v[reg];
#pragma unroll
        for (int r2 = 0; r2 < 4; ++r2)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const double cand = shfl_f64(v[r2], (l15 & 3) * 16 + lq + 4 * reg);
                if ((l15 >> 2) == r2) s[reg] = cand;
            }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            if (l15 < lq + 4 * reg) v[reg] = s[reg];
    }
    int colsrc[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) colsrc[h] = h * 16 + l15;
    MicroStep<0>::run(v, l15, lq, colsrc, index_base, badj);
}

__global__ __launch_bounds__(PIVOT_THREADS) void k_pivot(double *__restrict__ Akk, size_t ld, double *__restrict__ P,
                                                          size_t pld, gdca_dev_scalars *sc, int index0, int n_real)
{
    __shared__ __attribute__((aligned(16))) double Gs[MB * PV_ROW];
    __shared__ __attribute__((aligned(16))) double Ns[MB * PV_ROW];
    __shared__ __attribute__((aligned(16))) double Pms[2][MB][MB];  // -Pm of micro-block K in Pms[K & 1]
    __shared__ int badj;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) badj = 0;

    // tile ownership: waves 0..7: (w, w) + off-diagonal tiles 2w, 2w+1; waves 8..11: off-diagonal tiles 16 + 3 (w-8) + {0,1,2}
    // (off-diagonal tile e <-> (rb, cb), rb > cb, e = rb (rb-1) / 2 + cb)
    int trb[3], tcb[3];
    auto offdiag = [](int e, int &rb, int &cb) {
        int r = 1;
        while ((r + 1) * r / 2 <= e) ++r;
        rb = r;
        cb = e - r * (r - 1) / 2;
    };
    if (wv < NMB) {
        trb[0] = tcb[0] = wv;
        offdiag(2 * wv, trb[1], tcb[1]);
        offdiag(2 * wv + 1, trb[2], tcb[2]);
    } else {
#pragma unroll
        for (int t = 0; t < 3; ++t) offdiag(16 + 3 * (wv - NMB) + t, trb[t], tcb[t]);
    }

    // load: acc[t][reg] = D[16 rb + l15][16 cb + lq + 4 reg]; diagonal tiles mirror their lower triangle
    double acc[3][4];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            int r = MB * trb[t] + l15, c = MB * tcb[t] + lq + 4 * reg;
            if (r < c) {
                const int x = r;
                r = c;
                c = x;
            }
            acc[t][reg] = Akk[(size_t)r + (size_t)c * ld];
        }
    __syncthreads();  // badj initialised
    if (wv == 0) {
        // Pm of micro-block 0 (no update precedes it)
        double v[4] = {acc[0][0], acc[0][1], acc[0][2], acc[0][3]};
        micro_pivot(v, lane, 0, &badj);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) Pms[0][lq + 4 * reg][l15] = v[reg];  // [j][i] = -Pm(i, j)
    }

#pragma unroll 1
    for (int K = 0; K < NMB; ++K) {
        // ---- phase A: the old column block K into Gs ([kk][row]); rows of micro-block K of both images ----
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (tcb[t] == K && trb[t] > K) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) Gs[pv_off(lq + 4 * reg) + MB * trb[t] + l15] = acc[t][reg];
            } else if (trb[t] == K && tcb[t] < K) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) Gs[pv_off(l15) + MB * tcb[t] + lq + 4 * reg] = acc[t][reg];
            }
        }
        __syncthreads();  // Gs complete; Pms[K & 1] (written in the previous update phase) visible
        // ---- phase B: Ns = -(G Pm) for the row blocks != K (waves 0..7, one row block each); specials by waves 8, 9 ----
        if (wv < NMB && wv != K) {
            double4_t g = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                const int kk = 4 * t4 + lq;
                const int hi = l15 > kk ? l15 : kk, lo = l15 > kk ? kk : l15;
                const double a = Pms[K & 1][lo][hi];                       // -Pm(l15, kk), lower triangle authoritative
                const double b = Gs[pv_off(kk) + MB * wv + l15];           // G(16 wv + l15, kk)
                g = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, g, 0, 0, 0);
            }
            // lane holds -(G Pm)(16 wv + l15, lq + 4 reg)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) Ns[pv_off(lq + 4 * reg) + MB * wv + l15] = g[reg];
        } else if (wv == NMB) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                Gs[pv_off(lq + 4 * reg) + MB * K + l15] = (l15 == lq + 4 * reg) ? -1.0 : 0.0;
        } else if (wv == NMB + 1) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int kk = lq + 4 * reg;
                const int hi = l15 > kk ? l15 : kk, lo = l15 > kk ? kk : l15;
                Ns[pv_off(kk) + MB * K + l15] = -Pms[K & 1][lo][hi];       // +Pm(l15, kk)
            }
        }
        __syncthreads();
        // ---- phase C: every tile <- [in row or column K ? 0 : tile] + Gs[rb] Ns[cb]^T ----
        auto update_tile = [&](int t) {
            double4_t c4;
            const bool fresh = trb[t] == K || tcb[t] == K;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) c4[reg] = fresh ? 0.0 : acc[t][reg];
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                const int kk = 4 * t4 + lq;
                const double a = Ns[pv_off(kk) + MB * tcb[t] + l15];
                const double b = Gs[pv_off(kk) + MB * trb[t] + l15];
                c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c4, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) acc[t][reg] = c4[reg];
        };
        update_tile(0);
        if (K + 1 < NMB && wv == K + 1) {
            // look-ahead: the next micro-pivot, beside the other waves' updates
            double v[4] = {acc[0][0], acc[0][1], acc[0][2], acc[0][3]};
            micro_pivot(v, lane, MB * (K + 1), &badj);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) Pms[(K + 1) & 1][lq + 4 * reg][l15] = v[reg];
        }
        update_tile(1);
        update_tile(2);
        __syncthreads();
    }

    // D = -inverse (lower-triangular tiles).  P = -D and A_KK = D, both as full symmetric matrices
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int r = MB * trb[t] + l15, c = MB * tcb[t] + lq + 4 * reg;
            if (r >= c) {
                const double v = acc[t][reg];
                P[(size_t)r + (size_t)c * pld] = -v;
                Akk[(size_t)r + (size_t)c * ld] = v;
                if (r > c) {
                    P[(size_t)c + (size_t)r * pld] = -v;
                    Akk[(size_t)c + (size_t)r * ld] = v;
                }
            }
        }
    if (tid == 0 && badj != 0 && (index0 + badj) <= n_real) {
        if (sc->info == 0) sc->info = index0 + badj;
    }
}

// -------------------------------------------------------------------------------------------------
// 128 x 128 x 128 tile product on f64 MFMA, shared by the panel and the update kernels:
//   acc(r, c) += sum_k Gsrc(r, k) * Hsrc(c, k)
// Gsrc(r,k) = gsrc[r + k*gld] (or gsrc[k + r*gld] when GT);  Hsrc(c,k) = hsrc[c + k*hld].
// -------------------------------------------------------------------------------------------------
// TM = MFMA tiles per wave along the column direction: the workgroup tile is 128 rows x (32 TM) columns
// (TM = 4: 128 x 128, TM = 2: 128 x 64 for the latency-critical panel product).
template <int TM>
struct StageRegs {
    double g[8], h[2 * TM];
};

template <bool GT, int TM>
__device__ __forceinline__ void stage_load(StageRegs<TM> &R, const double *__restrict__ gsrc, size_t gld,
                                           const double *__restrict__ hsrc, size_t hld, int kc, int tid)
{
    // addresses = (wave-uniform row base) + (one 32-bit per-thread offset): the bases stay in SGPRs and the loads
    // take the saddr form, so the staging costs one address VGPR per operand instead of a 64-bit pointer per load
    if (!GT) {
        const unsigned boff = ((unsigned)((tid & 63) * 2) + (unsigned)(tid >> 6) * (unsigned)gld) * 8u;  // bytes
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const char *rowbase = reinterpret_cast<const char *>(gsrc + (size_t)(kc + 4 * u) * gld);
            const double2 v = *reinterpret_cast<const double2 *>(rowbase + boff);
            R.g[2 * u] = v.x;
            R.g[2 * u + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = tid + 256 * u;
            const int r = idx >> 3, k2 = idx & 7;
            const double2 v = *reinterpret_cast<const double2 *>(gsrc + (size_t)(kc + k2 * 2) + (size_t)r * gld);
            R.g[2 * u] = v.x;
            R.g[2 * u + 1] = v.y;
        }
    }
    constexpr int CP = 16 * TM;  // double2 per k-row of the H chunk
    if (CP == 64) {
        const unsigned boff = ((unsigned)((tid & 63) * 2) + (unsigned)(tid >> 6) * (unsigned)hld) * 8u;
#pragma unroll
        for (int u = 0; u < TM; ++u) {
            const char *rowbase = reinterpret_cast<const char *>(hsrc + (size_t)(kc + 4 * u) * hld);
            const double2 w = *reinterpret_cast<const double2 *>(rowbase + boff);
            R.h[2 * u] = w.x;
            R.h[2 * u + 1] = w.y;
        }
    } else {
#pragma unroll
        for (int u = 0; u < TM; ++u) {
            const int idx = tid + 256 * u;
            const int kk = idx / CP, c2 = idx % CP;
            const double2 w = *reinterpret_cast<const double2 *>(hsrc + (size_t)(c2 * 2) + (size_t)(kc + kk) * hld);
            R.h[2 * u] = w.x;
            R.h[2 * u + 1] = w.y;
        }
    }
}

template <bool GT, int TM>
__device__ __forceinline__ void stage_store(const StageRegs<TM> &R, double (*Gs)[LDS_LD], double (*Hs)[LDS_LD], int tid)
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
    }
    constexpr int CP = 16 * TM;
#pragma unroll
    for (int u = 0; u < TM; ++u) {
        const int idx = tid + 256 * u;
        const int kk = idx / CP, c2 = idx % CP;
        *reinterpret_cast<double2 *>(&Hs[kk][c2 * 2]) = make_double2(R.h[2 * u], R.h[2 * u + 1]);
    }
}

template <int TM, int K4_BEGIN = 0, int K4_END = KC>
__device__ __forceinline__ void chunk_mma(double4_t (&acc)[TM][4], double (*Gs)[LDS_LD], double (*Hs)[LDS_LD], int wr,
                                          int wc, int lane)
{
    const int l15 = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int k4 = K4_BEGIN; k4 < K4_END; k4 += 4) {
        double a[TM], b[4];
#pragma unroll
        for (int t = 0; t < TM; ++t) a[t] = Hs[k4 + lq][wc * (16 * TM) + t * 16 + l15];
#pragma unroll
        for (int t = 0; t < 4; ++t) b[t] = Gs[k4 + lq][wr * 64 + t * 16 + l15];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
    }
}

template <bool GT, int TM>
__device__ __forceinline__ void tile_product(double4_t (&acc)[TM][4], const double *__restrict__ gsrc, size_t gld,
                                             const double *__restrict__ hsrc, size_t hld, double (*Gs)[LDS_LD],
                                             double (*Hs)[LDS_LD], double *__restrict__ gcopy, size_t gcopy_ld)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv & 1, wc = wv >> 1;
    StageRegs<TM> R;
    stage_load<GT, TM>(R, gsrc, gld, hsrc, hld, 0, tid);
    for (int kc = 0; kc < T; kc += KC) {
        __syncthreads();  // previous chunk's LDS reads are done
        stage_store<GT, TM>(R, Gs, Hs, tid);
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
        if (kc + KC < T) stage_load<GT, TM>(R, gsrc, gld, hsrc, hld, kc + KC, tid);
        chunk_mma<TM>(acc, Gs, Hs, wr, wc, lane);
    }
}

// The update kernel's variant of tile_product: acc starts at ZERO and the 128 x 128 tile of C it is added to is
// fetched in eight pieces, one per k-chunk, each consumed one chunk after it was requested.  The tile's 128 KB
// then stream in underneath the MFMAs instead of as one blocking burst in front of them (every workgroup of a
// launch runs in lockstep, so that burst was 64 MB at once: measured 6 of the 39 us a tile takes).
// Piece ci = accumulators (tm = ci / 2, tn = 2 (ci % 2) + {0, 1}, reg 0..3): 8 doubles per lane.
template <int CI>
__device__ __forceinline__ void cpiece_load(double (&cp)[8], const double *__restrict__ At, size_t ld, int wr, int wc,
                                            int l15, int lq)
{
    constexpr int tm = CI / 2;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int tn = 2 * (CI % 2) + h;
            const int r = wr * 64 + tn * 16 + l15;
            const int c = wc * 64 + tm * 16 + lq + 4 * reg;
            cp[h * 4 + reg] = At[(size_t)r + (size_t)c * ld];
        }
}

template <int CI>
__device__ __forceinline__ void cpiece_add(double4_t (&acc)[4][4], const double (&cp)[8])
{
    constexpr int tm = CI / 2;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) acc[tm][2 * (CI % 2) + h][reg] += cp[h * 4 + reg];
}

// DUAL: two pivots fused in one pass -- the k loop runs over (G, H) of the first pivot (these unrolled chunks,
// which also bring in the C tile) and then over (G2, H2) of the second (a plain rolled loop in the kernel), K = 256:
// the C tile is read and written once per TWO rank-128 updates.
// The operand chunks are double-buffered in LDS (chunk c in buffer c & 1): while chunk c is multiplied, chunk c+1
// goes registers -> LDS (after the first quarter of the MFMAs, so its global loads have had more than a chunk to
// land) and chunk c+2's global loads are issued; ONE barrier per chunk.
typedef double (*lds_chunk_t)[LDS_LD];

template <int CI, bool DUAL>
struct UpdateChunks {
    static constexpr int PER = T / KC;  // chunks per operand pair
    static __device__ __forceinline__ void run(double4_t (&acc)[4][4], StageRegs<4> &R, double (&cp)[8],
                                               const double *__restrict__ g1, const double *__restrict__ h1,
                                               const double *__restrict__ g2, const double *__restrict__ h2, size_t pld,
                                               double (*Gs)[KC][LDS_LD], double (*Hs)[KC][LDS_LD],
                                               const double *__restrict__ At, size_t ld)
    {
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        const int wr = wv & 1, wc = wv >> 1, l15 = lane & 15, lq = lane >> 4;
        // on entry: LDS buffer CI & 1 holds chunk CI (barrier passed); R holds (or is receiving) chunk CI + 1
        if constexpr (CI > 0) cpiece_add<(CI > 0 ? CI - 1 : 0)>(acc, cp);  // requested one chunk ago
        cpiece_load<CI>(cp, At, ld, wr, wc, l15, lq);
        chunk_mma<4, 0, 4>(acc, Gs[CI & 1], Hs[CI & 1], wr, wc, lane);
        if constexpr (CI + 1 < PER || DUAL) stage_store<false, 4>(R, Gs[(CI + 1) & 1], Hs[(CI + 1) & 1], tid);
        if constexpr (CI + 2 < PER)
            stage_load<false, 4>(R, g1, pld, h1, pld, (CI + 2) * KC, tid);
        else if constexpr (DUAL)
            stage_load<false, 4>(R, g2, pld, h2, pld, (CI + 2 - PER) * KC, tid);
        chunk_mma<4, 4, KC>(acc, Gs[CI & 1], Hs[CI & 1], wr, wc, lane);
        __syncthreads();  // buffer CI & 1 is free, buffer (CI + 1) & 1 is complete
        if constexpr (CI + 1 < PER)
            UpdateChunks<CI + 1, DUAL>::run(acc, R, cp, g1, h1, g2, h2, pld, Gs, Hs, At, ld);
        else
            cpiece_add<CI>(acc, cp);
    }
};

// Panel: for every row block i != k:  G_i = column block k of the symmetric matrix (read from the
// lower triangle: A[i,k] for i > k, A[k,i]^T for i < k);  GP = G_i P;  writes
//   Gbuf[i] = G_i,  Hbuf[i] = -GP,  and the new column block  A[i,k] = GP  (A[k,i] = GP^T for i < k).
// Two workgroups per row block (64 columns of GP each): the panel sits on the critical path of the
// look-ahead chain, so it is cut finer than the throughput-bound update.
__global__ __launch_bounds__(256, 2) void k_panel(double *__restrict__ A, size_t ld, int kblk,
                                                   const double *__restrict__ P, double *__restrict__ Gbuf,
                                                   double *__restrict__ Hbuf, size_t pld)
{
    __shared__ __attribute__((aligned(16))) double Gs[KC][LDS_LD];
    __shared__ __attribute__((aligned(16))) double Hs[KC][LDS_LD];
    int i = blockIdx.x;
    if (i >= kblk) ++i;  // skip the pivot block itself
    const int ch = blockIdx.y;  // which 64-column half of GP
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv & 1, wc = wv >> 1, l15 = lane & 15, lq = lane >> 4;
    double4_t acc[2][4];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = (double4_t){0.0, 0.0, 0.0, 0.0};

    double *gcopy = (ch == 0) ? Gbuf + (size_t)i * T : nullptr;
    const double *hsrc = P + (size_t)ch * 64;  // rows (= columns, P symmetric) ch*64 .. ch*64+63 of P
    if (i > kblk) {
        const double *gsrc = A + (size_t)i * T + (size_t)kblk * T * ld;
        tile_product<false, 2>(acc, gsrc, ld, hsrc, T, Gs, Hs, gcopy, pld);
    } else {
        const double *gsrc = A + (size_t)kblk * T + (size_t)i * T * ld;
        tile_product<true, 2>(acc, gsrc, ld, hsrc, T, Gs, Hs, gcopy, pld);
    }
    // acc[tm][tn][reg] = GP(r, c):  r = wr*64 + tn*16 + l15,  c = ch*64 + wc*32 + tm*16 + lq + 4*reg
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = wr * 64 + tn * 16 + l15;
                const int c = ch * 64 + wc * 32 + tm * 16 + lq + 4 * reg;
                const double v = acc[tm][tn][reg];
                Hbuf[(size_t)i * T + r + (size_t)c * pld] = -v;
            }
}

// Write-back of the new column block: A[i,k] = G_i P = -H_i  (A[k,i] = (G_i P)^T for i < k).  Not done by
// the panel kernel because the two panel workgroups of a row block both read the OLD A[i,k] as their G
// operand.  Runs as extra workgroups of the look-ahead update launch (or as a launch of its own at the
// last step): every thread first loads all of its 64 values, then stores them.
__device__ __forceinline__ void panel_writeback_tile(double *__restrict__ A, size_t ld, int kblk, int i,
                                                     const double *__restrict__ Hbuf, size_t pld, double (*Ts)[LDS_LD])
{
    const int tid = threadIdx.x;
    const double *H = Hbuf + (size_t)i * T;
    if (i > kblk) {
        double *dst = A + (size_t)i * T + (size_t)kblk * T * ld;
        double v[64];
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            const int idx = tid + 256 * u;  // r = idx & 127, c = idx >> 7
            v[u] = H[(size_t)(idx & 127) + (size_t)(idx >> 7) * pld];
        }
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            const int idx = tid + 256 * u;
            dst[(size_t)(idx & 127) + (size_t)(idx >> 7) * ld] = -v[u];
        }
    } else {
        double *dst = A + (size_t)kblk * T + (size_t)i * T * ld;  // dst(c, r) = -H(r, c)
        // 16 columns of H at a time through LDS: Ts[c][r], then rows of dst are read across c
        for (int cb = 0; cb < T; cb += KC) {
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = tid + 256 * u;  // r = idx & 127, c = idx >> 7 (0..15)
                Ts[idx >> 7][idx & 127] = -H[(size_t)(idx & 127) + (size_t)(cb + (idx >> 7)) * pld];
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = tid + 256 * u;  // c = idx & 15, r = idx >> 4
                dst[(size_t)(cb + (idx & 15)) + (size_t)(idx >> 4) * ld] = Ts[idx & 15][idx >> 4];
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_panel_writeback(double *__restrict__ A, size_t ld, int kblk,
                                                          const double *__restrict__ Hbuf, size_t pld)
{
    __shared__ __attribute__((aligned(16))) double Ts[KC][LDS_LD];
    int i = blockIdx.x;
    if (i >= kblk) ++i;
    panel_writeback_tile(A, ld, kblk, i, Hbuf, pld, Ts);
}

// Update: lower-triangle tiles  A_IJ += G_I H_J^T (+ G2_I H2_J^T when G2 != nullptr),  H = -G P, in one of two tile
// sets; blocks in the contiguous range [skip_lo, skip_lo + skip_n) never take part:
//   colblk <  0 : every tile (I >= J) over the remaining blocks
//   SLICE       : the tiles that involve block `colblk` as row or column (nslice1 = nblk - skip_n of them) and,
//                 when colblk2 >= 0, those that involve colblk2 = skip_lo + skip_n + 1 (nslice - nslice1 more) --
//                 look-ahead slices; workgroups past nslice do the write-back of the pivot column block wb_col
//                 from wbH (see panel_writeback_tile) when wb_col >= 0.
// SLICE is a template parameter so that the big trailing-update launches are their own kernel symbol
// (k_sweep_update<DUAL, false>): profiler summaries then report them apart from the small look-ahead launches.
template <bool DUAL, bool SLICE>
__global__ __launch_bounds__(256, 2) void k_sweep_update(double *__restrict__ A, size_t ld, int skip_lo, int skip_n,
                                                          int colblk, int nslice, const double *__restrict__ Gbuf,
                                                          const double *__restrict__ Hbuf,
                                                          const double *__restrict__ G2buf,
                                                          const double *__restrict__ H2buf, size_t pld, int wb_col,
                                                          const double *__restrict__ wbH, int colblk2, int nslice1)
{
    __shared__ __attribute__((aligned(16))) double Gs[2][KC][LDS_LD];
    __shared__ __attribute__((aligned(16))) double Hs[2][KC][LDS_LD];
    const int t = blockIdx.x;
    int I, J;
    if constexpr (SLICE) {
        if (t >= nslice) {
            int b = t - nslice;
            if (b >= wb_col) ++b;
            panel_writeback_tile(A, ld, wb_col, b, wbH, pld, Gs[0]);
            return;
        }
    }
    if constexpr (SLICE) {
        // tiles [0, nslice1) involve block colblk; tiles [nslice1, nslice) involve block colblk2, whose skip range
        // is one longer (it also leaves out colblk = skip_lo + skip_n: that tile belongs to the first set)
        const bool second = t >= nslice1;
        int b = second ? t - nslice1 : t;
        if (b >= skip_lo) b += skip_n + (second ? 1 : 0);
        const int cb = second ? colblk2 : colblk;
        I = b > cb ? b : cb;
        J = b > cb ? cb : b;
    } else {
        int ii = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((long long)ii * (ii + 1) / 2 > t) --ii;
        while ((long long)(ii + 1) * (ii + 2) / 2 <= t) ++ii;
        int jj = t - (int)((long long)ii * (ii + 1) / 2);
        if (ii >= skip_lo) ii += skip_n;
        if (jj >= skip_lo) jj += skip_n;
        I = ii;
        J = jj;
    }

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv & 1, wc = wv >> 1, l15 = lane & 15, lq = lane >> 4;
    double *At = A + (size_t)I * T + (size_t)J * T * ld;
    double4_t acc[4][4];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = (double4_t){0.0, 0.0, 0.0, 0.0};
    {
        StageRegs<4> R;
        double cp[8];
        const double *g1 = Gbuf + (size_t)I * T, *h1 = Hbuf + (size_t)J * T;
        stage_load<false, 4>(R, g1, pld, h1, pld, 0, tid);
        stage_store<false, 4>(R, Gs[0], Hs[0], tid);
        stage_load<false, 4>(R, g1, pld, h1, pld, KC, tid);
        __syncthreads();
        if constexpr (DUAL) {
            const double *g2 = G2buf + (size_t)I * T, *h2 = H2buf + (size_t)J * T;
            UpdateChunks<0, true>::run(acc, R, cp, g1, h1, g2, h2, pld, Gs, Hs, At, ld);
            // second pivot of the pair: chunk c of it is chunk 8 + c of the pass (buffer c & 1); on entry chunk 0 is
            // in LDS buffer 0 and chunk 1 in R
#pragma unroll 1
            for (int c = 0; c < T / KC; ++c) {
                chunk_mma<4, 0, 4>(acc, Gs[c & 1], Hs[c & 1], wr, wc, lane);
                if (c + 1 < T / KC) stage_store<false, 4>(R, Gs[(c + 1) & 1], Hs[(c + 1) & 1], tid);
                if (c + 2 < T / KC) stage_load<false, 4>(R, g2, pld, h2, pld, (c + 2) * KC, tid);
                chunk_mma<4, 4, KC>(acc, Gs[c & 1], Hs[c & 1], wr, wc, lane);
                __syncthreads();
            }
        } else {
            UpdateChunks<0, false>::run(acc, R, cp, g1, h1, nullptr, nullptr, pld, Gs, Hs, At, ld);
        }
    }
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

// ---- groups of up to four pivots per trailing update ----------------------------------------------------------------
// Generalisation of the pair kernel above for large matrices: one launch applies the rank-128 updates of nop <= 4
// pivots (K = 128 nop) -- the C tile is read and written once per nop updates and a workgroup's launch / first-chunk /
// store-drain overhead is paid once per nop times the work.  Big mode: every tile outside [skip_lo, skip_lo + skip_n).
// Slice mode: the tiles that involve one of up to four column blocks col[m], the other index running over all blocks
// outside that column's own contiguous skip range [cskip_lo[m], cskip_lo[m] + cskip_n[m]) (first[m] .. first[m+1]-1
// are column m's tiles); workgroups from first[ncol] on write pivot column wb_col back from wbH.
struct GroupUpd {
    const double *G[4];
    const double *H[4];
    int nop;
    int skip_lo, skip_n;
    int ncol;
    int col[4], first[5], cskip_lo[4], cskip_n[4];
    int wb_col;
    const double *wbH;
};

template <bool SLICE, bool MULTI>
__global__ __launch_bounds__(256, 2) void k_group_update(double *__restrict__ A, size_t ld, size_t pld, const GroupUpd P)
{
    __shared__ __attribute__((aligned(16))) double Gs[2][KC][LDS_LD];
    __shared__ __attribute__((aligned(16))) double Hs[2][KC][LDS_LD];
    const int t = blockIdx.x;
    int I, J;
    if constexpr (SLICE) {
        if (t >= P.first[P.ncol]) {
            int b = t - P.first[P.ncol];
            if (b >= P.wb_col) ++b;
            panel_writeback_tile(A, ld, P.wb_col, b, P.wbH, pld, Gs[0]);
            return;
        }
        int m = 0;
        while (m + 1 < P.ncol && t >= P.first[m + 1]) ++m;
        int b = t - P.first[m];
        if (b >= P.cskip_lo[m]) b += P.cskip_n[m];
        const int cb = P.col[m];
        I = b > cb ? b : cb;
        J = b > cb ? cb : b;
    } else {
        int ii = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((long long)ii * (ii + 1) / 2 > t) --ii;
        while ((long long)(ii + 1) * (ii + 2) / 2 <= t) ++ii;
        int jj = t - (int)((long long)ii * (ii + 1) / 2);
        if (ii >= P.skip_lo) ii += P.skip_n;
        if (jj >= P.skip_lo) jj += P.skip_n;
        I = ii;
        J = jj;
    }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv & 1, wc = wv >> 1, l15 = lane & 15, lq = lane >> 4;
    double *At = A + (size_t)I * T + (size_t)J * T * ld;
    double4_t acc[4][4];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = (double4_t){0.0, 0.0, 0.0, 0.0};
    {
        StageRegs<4> R;
        double cp[8];
        const size_t go = (size_t)I * T, ho = (size_t)J * T;
        const double *g1 = P.G[0] + go, *h1 = P.H[0] + ho;
        stage_load<false, 4>(R, g1, pld, h1, pld, 0, tid);
        stage_store<false, 4>(R, Gs[0], Hs[0], tid);
        stage_load<false, 4>(R, g1, pld, h1, pld, KC, tid);
        __syncthreads();
        if constexpr (MULTI) {
            UpdateChunks<0, true>::run(acc, R, cp, g1, h1, P.G[1] + go, P.H[1] + ho, pld, Gs, Hs, At, ld);
            // pivots 2 .. nop of the group: chunk c of this loop is chunk 8 + c of the pass (LDS buffer c & 1); on entry
            // chunk 0 is in LDS buffer 0 and chunk 1 in R
            const int total = (T / KC) * (P.nop - 1);
#pragma unroll 1
            for (int c = 0; c < total; ++c) {
                chunk_mma<4, 0, 4>(acc, Gs[c & 1], Hs[c & 1], wr, wc, lane);
                if (c + 1 < total) stage_store<false, 4>(R, Gs[(c + 1) & 1], Hs[(c + 1) & 1], tid);
                if (c + 2 < total) {
                    const int op = 1 + (c + 2) / (T / KC), kc = ((c + 2) % (T / KC)) * KC;
                    stage_load<false, 4>(R, P.G[op] + go, pld, P.H[op] + ho, pld, kc, tid);
                }
                chunk_mma<4, 4, KC>(acc, Gs[c & 1], Hs[c & 1], wr, wc, lane);
                __syncthreads();
            }
        } else {
            UpdateChunks<0, false>::run(acc, R, cp, g1, h1, nullptr, nullptr, pld, Gs, Hs, At, ld);
        }
    }
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

// Host driver of the block sweep.  With a side stream (s1 != nullptr) it runs with look-ahead, by default in
// PAIRS of pivots: the big trailing update of pivots (2p, 2p+1) is ONE launch with K = 256 on the side stream
// (half the C-tile traffic per flop of two K = 128 launches -- the trailing update is HBM-bound otherwise), while
// the main (high-priority) stream runs the chain of the next pair: the slices of the pair update that touch
// blocks 2p+2 and 2p+3, pivot/panel of 2p+2, its rank-128 update of column 2p+3, pivot/panel of 2p+3 and that
// pivot's update of column 2p+2.  Panels are double-buffered by pair parity.  The single-pivot look-ahead
// (GDCA_PAIRS=0) and the serial schedule (s1 == nullptr) are kept for small matrices and for comparison.
void gdca_launch_spd_inverse(hipStream_t s0, hipStream_t s1, double *A, int n_pad, const gdca_inverse_ws &ws,
                             gdca_dev_scalars *sc, int n_real, hipEvent_t *sync_ev, hipEvent_t *upd_ev, int max_upd_ev,
                             int *n_upd_launch, double *upd_flops)
{
    const int nblk = n_pad / T;
    const size_t ld = (size_t)n_pad;
    const double tile_flops = 2.0 * T * T * T;
    int nl = 0;
    double fl = 0.0;
    // big launch over all blocks outside [skip_lo, skip_lo + skip_n)
    auto timed_update = [&](hipStream_t st, int skip_lo, int skip_n, const double *G, const double *H, const double *G2,
                            const double *H2) {
        const int m = nblk - skip_n;
        if (m <= 0) return;
        const unsigned ntile = (unsigned)((long long)m * (m + 1) / 2);
        const bool tm = upd_ev && 2 * nl + 1 < max_upd_ev;
        if (tm) (void)hipEventRecord(upd_ev[2 * nl], st);
        if (G2)
            hipLaunchKernelGGL((k_sweep_update<true, false>), dim3(ntile), dim3(256), 0, st, A, ld, skip_lo, skip_n, -1, 0, G, H,
                               G2, H2, ld, -1, (const double *)nullptr, -1, 0);
        else
            hipLaunchKernelGGL((k_sweep_update<false, false>), dim3(ntile), dim3(256), 0, st, A, ld, skip_lo, skip_n, -1, 0, G,
                               H, G2, H2, ld, -1, (const double *)nullptr, -1, 0);
        if (tm) (void)hipEventRecord(upd_ev[2 * nl + 1], st);
        ++nl;
        fl += tile_flops * (double)ntile * (G2 ? 2.0 : 1.0);
    };
    // slice: the tiles that involve block col (other index outside the skip range), plus, when wb_col >= 0, the
    // write-back of pivot column wb_col from wbH
    auto slice = [&](hipStream_t st, int col, int skip_lo, int skip_n, const double *G, const double *H, const double *G2,
                     const double *H2, int wb_col, const double *wbH, int col2 = -1) {
        const int ns1 = nblk - skip_n;
        const int ns = ns1 + (col2 >= 0 ? nblk - skip_n - 1 : 0);
        const dim3 grid(ns + (wb_col >= 0 ? nblk - 1 : 0));
        if (G2)
            hipLaunchKernelGGL((k_sweep_update<true, true>), grid, dim3(256), 0, st, A, ld, skip_lo, skip_n, col, ns, G, H, G2,
                               H2, ld, wb_col, wbH, col2, ns1);
        else
            hipLaunchKernelGGL((k_sweep_update<false, true>), grid, dim3(256), 0, st, A, ld, skip_lo, skip_n, col, ns, G, H, G2,
                               H2, ld, wb_col, wbH, col2, ns1);
    };
    auto pivot = [&](int k) {
        hipLaunchKernelGGL(k_pivot, dim3(1), dim3(PIVOT_THREADS), 0, s0, A + (size_t)k * T + (size_t)k * T * ld, ld, ws.P, (size_t)T,
                           sc, k * T, n_real);
    };
    auto panel = [&](int k, double *G, double *H) {
        hipLaunchKernelGGL(k_panel, dim3(nblk - 1, 2), dim3(256), 0, s0, A, ld, k, ws.P, G, H, ld);
    };
    // pairs pay off once the trailing update dominates (measured crossover on MI355X at 66 blocks = n ~ 8400;
    // below that the longer chain per pivot costs more than the halved C-tile traffic saves); GDCA_PAIRS=0/1 forces
    static const int pairs_env = getenv("GDCA_PAIRS") ? atoi(getenv("GDCA_PAIRS")) : -1;
    const bool pairs_on = pairs_env < 0 ? nblk >= 66 : pairs_env != 0;
    // groups of 3-4 pivots per trailing update (k_group_update) once the matrix is large enough for the longer chain of
    // a group to stay hidden (measured: 3 from 66 blocks, 4 from 90); GDCA_GROUP=g forces (0..2: pairs / single pivots)
    static const int group_env = getenv("GDCA_GROUP") ? atoi(getenv("GDCA_GROUP")) : -1;
    const int group_g = group_env >= 0 ? std::min(group_env, 4) : (nblk >= 90 ? 4 : (nblk >= 66 ? 3 : 0));

    pivot(0);
    if (nblk > 1) {
        if (!s1) {
            // serial schedule: pivot -> panel -> write-back -> full update, one stream
            for (int k = 0; k < nblk; ++k) {
                if (k > 0) pivot(k);
                panel(k, ws.G[0], ws.H[0]);
                hipLaunchKernelGGL(k_panel_writeback, dim3(nblk - 1), dim3(256), 0, s0, A, ld, k, ws.H[0], ld);
                timed_update(s0, k, 1, ws.G[0], ws.H[0], nullptr, nullptr);
            }
        } else if (group_g >= 3 && nblk >= 2 * group_g && ws.G[4]) {
            // ---- groups of group_g pivots (large matrices) ----
            hipEvent_t *Ep = sync_ev, *Eb = sync_ev + nblk;
            const int g = group_g, ng = (nblk + g - 1) / g;
            auto base = [&](int p) { return p * g; };
            auto size = [&](int p) { return std::min(g, nblk - p * g); };
            auto GG = [&](int p, int w) { return ws.G[4 * (p & 1) + w]; };
            auto HH = [&](int p, int w) { return ws.H[4 * (p & 1) + w]; };
            auto launch_group = [&](hipStream_t st, bool slice_mode, unsigned grid, const GroupUpd &P) {
                if (slice_mode) {
                    if (P.nop > 1)
                        hipLaunchKernelGGL((k_group_update<true, true>), dim3(grid), dim3(256), 0, st, A, ld, ld, P);
                    else
                        hipLaunchKernelGGL((k_group_update<true, false>), dim3(grid), dim3(256), 0, st, A, ld, ld, P);
                } else {
                    if (P.nop > 1)
                        hipLaunchKernelGGL((k_group_update<false, true>), dim3(grid), dim3(256), 0, st, A, ld, ld, P);
                    else
                        hipLaunchKernelGGL((k_group_update<false, false>), dim3(grid), dim3(256), 0, st, A, ld, ld, P);
                }
            };
            auto set_ops = [&](GroupUpd &P, int p, int first_op, int nop) {
                for (int w = 0; w < 4; ++w) {
                    P.G[w] = GG(p, std::min(first_op + w, 3));
                    P.H[w] = HH(p, std::min(first_op + w, 3));
                }
                P.nop = nop;
            };
            // the chain of group p: pivots one after the other, each applied at once to the other columns of the group
            auto chain = [&](int p) {
                const int b0 = base(p), sz = size(p);
                for (int i = 0; i < sz; ++i) {
                    const int k = b0 + i;
                    if (k > 0) pivot(k);  // pivot 0 was launched above, before the schedules branch
                    panel(k, GG(p, i), HH(p, i));
                    GroupUpd P{};
                    set_ops(P, p, i, 1);
                    P.wb_col = k;
                    P.wbH = HH(p, i);
                    int m = 0, first = 0;
                    for (int c = k + 1; c < b0 + sz; ++c, ++m) {  // later columns of the group: skip [k, c)
                        P.col[m] = c;
                        P.cskip_lo[m] = k;
                        P.cskip_n[m] = c - k;
                        P.first[m] = first;
                        first += nblk - P.cskip_n[m];
                    }
                    for (int c = k - 1; c >= b0; --c, ++m) {  // earlier columns: skip (c, b0 + sz)
                        P.col[m] = c;
                        P.cskip_lo[m] = c + 1;
                        P.cskip_n[m] = b0 + sz - (c + 1);
                        P.first[m] = first;
                        first += nblk - P.cskip_n[m];
                    }
                    P.ncol = m;
                    P.first[m] = first;
                    launch_group(s0, true, (unsigned)(first + nblk - 1), P);
                }
            };
            chain(0);
            (void)hipEventRecord(Ep[0], s0);
            for (int p = 0; p < ng; ++p) {
                const int b0 = base(p), sz = size(p);
                const bool has_next = p + 1 < ng;
                const int nsz = has_next ? size(p + 1) : 0;
                // side stream: the group's update of everything outside its own and the next group's blocks
                (void)hipStreamWaitEvent(s1, Ep[p], 0);
                {
                    GroupUpd P{};
                    set_ops(P, p, 0, sz);
                    P.skip_lo = b0;
                    P.skip_n = sz + nsz;
                    const int m = nblk - P.skip_n;
                    if (m > 0) {
                        const unsigned ntile = (unsigned)((long long)m * (m + 1) / 2);
                        const bool tm = upd_ev && 2 * nl + 1 < max_upd_ev;
                        if (tm) (void)hipEventRecord(upd_ev[2 * nl], s1);
                        launch_group(s1, false, ntile, P);
                        if (tm) (void)hipEventRecord(upd_ev[2 * nl + 1], s1);
                        ++nl;
                        fl += tile_flops * (double)ntile * (double)sz;
                    }
                }
                (void)hipEventRecord(Eb[p], s1);
                if (!has_next) break;
                if (p >= 1) (void)hipStreamWaitEvent(s0, Eb[p - 1], 0);  // the next group's columns carry update p-1
                {
                    // the tiles of group p's update that involve the next group's blocks, one launch
                    GroupUpd P{};
                    set_ops(P, p, 0, sz);
                    P.wb_col = -1;
                    int first = 0;
                    for (int m = 0; m < nsz; ++m) {
                        P.col[m] = b0 + sz + m;
                        P.cskip_lo[m] = b0;
                        P.cskip_n[m] = sz + m;
                        P.first[m] = first;
                        first += nblk - P.cskip_n[m];
                    }
                    P.ncol = nsz;
                    P.first[nsz] = first;
                    launch_group(s0, true, (unsigned)first, P);
                }
                chain(p + 1);
                (void)hipEventRecord(Ep[p + 1], s0);
            }
            (void)hipStreamWaitEvent(s0, Eb[ng - 1], 0);
        } else if (pairs_on && nblk >= 6 && ws.G[2]) {
            hipEvent_t *Ep = sync_ev, *Eb = sync_ev + nblk;
            // panels of pair p: G[2 (p & 1) + {0, 1}]
            auto PG = [&](int p, int w) { return ws.G[2 * (p & 1) + w]; };
            auto PH = [&](int p, int w) { return ws.H[2 * (p & 1) + w]; };
            const int np = nblk / 2;
            // chain of pair 0
            panel(0, PG(0, 0), PH(0, 0));
            slice(s0, 1, 0, 1, PG(0, 0), PH(0, 0), nullptr, nullptr, 0, PH(0, 0));
            pivot(1);
            panel(1, PG(0, 1), PH(0, 1));
            slice(s0, 0, 1, 1, PG(0, 1), PH(0, 1), nullptr, nullptr, 1, PH(0, 1));
            (void)hipEventRecord(Ep[0], s0);
            for (int p = 0; p < np; ++p) {
                const int k1 = 2 * p, k3 = k1 + 2, k4 = k1 + 3;
                const bool has3 = k3 < nblk, has4 = k4 < nblk;
                // side stream: the pair's update of everything outside the pair and outside the next chain's columns
                (void)hipStreamWaitEvent(s1, Ep[p], 0);
                timed_update(s1, k1, 2 + (has3 ? 1 : 0) + (has4 ? 1 : 0), PG(p, 0), PH(p, 0), PG(p, 1), PH(p, 1));
                (void)hipEventRecord(Eb[p], s1);
                if (!has3) break;
                if (p >= 1) (void)hipStreamWaitEvent(s0, Eb[p - 1], 0);  // columns k3, k4 carry update p-1
                // one launch for the tiles of pair p's update that involve block k3 and (if any) block k4
                slice(s0, k3, k1, 2, PG(p, 0), PH(p, 0), PG(p, 1), PH(p, 1), -1, nullptr, has4 ? k4 : -1);
                pivot(k3);
                panel(k3, PG(p + 1, 0), PH(p + 1, 0));
                if (has4) {
                    slice(s0, k4, k3, 1, PG(p + 1, 0), PH(p + 1, 0), nullptr, nullptr, k3, PH(p + 1, 0));
                    pivot(k4);
                    panel(k4, PG(p + 1, 1), PH(p + 1, 1));
                    slice(s0, k3, k4, 1, PG(p + 1, 1), PH(p + 1, 1), nullptr, nullptr, k4, PH(p + 1, 1));
                    (void)hipEventRecord(Ep[p + 1], s0);
                } else {
                    // odd block count: the last pivot stands alone
                    hipLaunchKernelGGL(k_panel_writeback, dim3(nblk - 1), dim3(256), 0, s0, A, ld, k3, PH(p + 1, 0), ld);
                    (void)hipStreamWaitEvent(s0, Eb[p], 0);
                    timed_update(s0, k3, 1, PG(p + 1, 0), PH(p + 1, 0), nullptr, nullptr);
                }
            }
            (void)hipStreamWaitEvent(s0, Eb[np - 1], 0);
        } else {
            hipEvent_t *Ep = sync_ev, *Eb = sync_ev + nblk;
            panel(0, ws.G[0], ws.H[0]);
            (void)hipEventRecord(Ep[0], s0);
            for (int k = 0; k < nblk; ++k) {
                const bool has_next = k + 1 < nblk;
                const double *G = ws.G[k & 1], *H = ws.H[k & 1];
                // side stream: everything of update k that does not touch block k+1
                (void)hipStreamWaitEvent(s1, Ep[k], 0);
                timed_update(s1, k, has_next ? 2 : 1, G, H, nullptr, nullptr);
                (void)hipEventRecord(Eb[k], s1);
                if (has_next) {
                    if (k >= 1) (void)hipStreamWaitEvent(s0, Eb[k - 1], 0);
                    // look-ahead slice: the nblk-1 tiles in row/column k+1 (+ write-back of column k), then the next
                    // pivot and panel
                    slice(s0, k + 1, k, 1, G, H, nullptr, nullptr, k, H);
                    pivot(k + 1);
                    panel(k + 1, ws.G[(k + 1) & 1], ws.H[(k + 1) & 1]);
                    (void)hipEventRecord(Ep[k + 1], s0);
                } else {
                    // last pivot: no look-ahead launch to carry its column write-back
                    hipLaunchKernelGGL(k_panel_writeback, dim3(nblk - 1), dim3(256), 0, s0, A, ld, k, H, ld);
                }
            }
            (void)hipStreamWaitEvent(s0, Eb[nblk - 1], 0);
        }
    }
    if (n_upd_launch) *n_upd_launch = nl;
    if (upd_flops) *upd_flops = fl;
}

// -------------------------------------------------------------------------------------------------
// f64 MFMA issue-rate probe (register-resident, 8 independent accumulators per wave).
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_probe_mfma_f64(double *out, int iters)
{
    // 16 independent accumulators per wave (4 A x 4 B fragments, as in the tile kernels): with 8 the loop is bound by
    // the accumulator dependency, not by the matrix pipe (46-49 instead of 76-77 TFLOP/s chip-wide)
    double4_t acc[4][4];
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = 1.0 + 1e-9 * (threadIdx.x + 64 * i);
        b[i] = 1.0 - 1e-9 * (threadIdx.x + 64 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    }
    for (int it = 0; it < iters; it += 2) {  // 16 MFMAs per trip = two of the former 8-MFMA iterations
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    double sacc = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = sacc;
}

void gdca_launch_probe_mfma_f64(hipStream_t s, double *out, int iters, int blocks)
{
    hipLaunchKernelGGL(k_probe_mfma_f64, dim3(blocks), dim3(256), 0, s, out, iters);
}
