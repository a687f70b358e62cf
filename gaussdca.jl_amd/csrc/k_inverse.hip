// Dense SPD inverse  mJ = inv(cholesky(C))  (reference call site src/GaussDCA.jl:34; there:
// LAPACK dpotrf + dpotri, n^3 flops).
//
// MI355X design: a BLOCK SYMMETRIC SWEEP (block Gauss-Jordan on an SPD matrix).  For a pivot block (group) K with
// D = A_KK (the Schur complement at that point -- the same matrix Cholesky would factor, so the
// positive-definiteness test and the `info` index are the same as dpotrf's), P = D^-1, G = A_{.,K}:
//       A_ij <- A_ij - G_i P G_j^T   (i, j != K),   A_{.,K} <- G P,   A_KK <- -P.
// After all pivots A = -C^-1.  Same n^3 flop count as dpotrf+dpotri, but every step is ONE launch of
// ~(n/128)^2/2 identical 128x128 tile products over the whole lower triangle: no shrinking trailing matrix, no
// trtri/lauum dependency chains, no tail of tiny launches -- the shape a 256-CU chip wants.  The matrix stays
// symmetric throughout, only the lower triangle (with full diagonal tiles) is touched.  Pivots are taken in groups
// of up to four 128-blocks (one update launch of depth K = 128 g per group); the serial part of a group -- the
// inverse of its 128 g x 128 g diagonal super-block -- runs on a small dense scratch copy while the bulk of the
// look-ahead work proceeds beside it (see the driver at the end of this file).
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

// threadIdx.x behind a volatile asm: inside the persistent loop of the update kernel the compiler would otherwise treat
// every per-thread address offset of every path (hundreds of values) as loop-invariant, hoist them all and spill them
__device__ __forceinline__ int opaque_tid()
{
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

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
// sc->info.  Writes P (full symmetric, ld = pld) and -P (full tile) to the output tile.
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

// Ain (ld = ldin) is read, Aout (ld = ldout) receives -P; the two may be the same tile.
__global__ __launch_bounds__(PIVOT_THREADS) void k_pivot(const double *Ain, size_t ldin, double *Aout, size_t ldout,
                                                          double *__restrict__ P, size_t pld, gdca_dev_scalars *sc, int index0,
                                                          int n_real)
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
            acc[t][reg] = Ain[(size_t)r + (size_t)c * ldin];
        }
    __syncthreads();  // badj initialised

    // K = -1 is the prologue: only the micro-pivot of micro-block 0 (no update precedes it); the same code as the
    // look-ahead micro-pivots of the loop, so that the kernel carries ONE copy of the unrolled 16-step sweep (the pivot
    // is launched once per block with other kernels in between: its instructions are fetched cold every time)
#pragma unroll 1
    for (int K = -1; K < NMB; ++K) {
      if (K >= 0) {
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
      }
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
        if (K >= 0) update_tile(0);
        if (K + 1 < NMB && wv == K + 1) {
            // look-ahead: the next micro-pivot, beside the other waves' updates
            double v[4] = {acc[0][0], acc[0][1], acc[0][2], acc[0][3]};
            micro_pivot(v, lane, MB * (K + 1), &badj);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) Pms[(K + 1) & 1][lq + 4 * reg][l15] = v[reg];  // [j][i] = -Pm(i, j)
        }
        if (K >= 0) {
            update_tile(1);
            update_tile(2);
            __syncthreads();
        }
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
                Aout[(size_t)r + (size_t)c * ldout] = v;
                if (r > c) {
                    P[(size_t)c + (size_t)r * pld] = -v;
                    Aout[(size_t)c + (size_t)r * ldout] = v;
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
    const int tid = opaque_tid(), lane = tid & 63, wv = tid >> 6;
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

// DUAL: more pivots follow in the same pass -- the k loop runs over (G, H) of the first pivot (these unrolled chunks,
// which also bring in the C tile) and then over the panels of the other pivots of the group (a plain rolled loop in the
// kernel): the C tile is read and written once per group of rank-128 updates.
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
        const int tid = opaque_tid(), lane = tid & 63, wv = tid >> 6;
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

// Write-back of the new column block: A[i,k] = G_i P = -H_i  (A[k,i] = (G_i P)^T for i < k).  Not done by
// the panel kernel because the two panel workgroups of a row block both read the OLD A[i,k] as their G
// operand.  Runs as extra workgroups of the look-ahead update launch (or as a launch of its own at the
// last step): every thread first loads all of its 64 values, then stores them.
__device__ __forceinline__ void panel_writeback_tile(double *__restrict__ A, size_t ld, int kblk, int i,
                                                     const double *__restrict__ Hbuf, size_t pld, double (*Ts)[LDS_LD])
{
    const int tid = opaque_tid();
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

// ---- trailing update: up to four pivots per launch -------------------------------------------------------------
// One launch applies the rank-128 updates of the nop <= 4 pivots of a group (K = 128 nop): lower-triangle tiles
//       A_IJ += sum_w G_w[I] H_w[J]^T,     H_w = -(G Pg)_w,
// so the C tile is read and written once per nop updates and a workgroup's launch / first-chunk / store-drain
// overhead is paid once per nop times the work (with one pivot per launch the update is HBM-bound).
//
// The launch covers every tile outside the group's own blocks, in an ORDER that serves the look-ahead: workgroups are
// dispatched by ascending block index, and the HEAD of the grid holds what the pivot chain of the NEXT group waits for,
//   1. the tiles inside the next group's diagonal super-block              -> counter cnt[0] (one tick per workgroup)
//   2. the write-back of this group's new columns (A[., g] <- -H)           -> counter cnt[1]
//   3. the other tiles in the next group's columns (all remaining rows)     -> counter cnt[1]
// followed by the big remainder (every tile over the blocks outside [skip_lo, skip_lo + skip_n) = this group and the
// next).  A kind-1 workgroup makes its stores visible device-wide (agent-scope release) and ticks cnt[0]; the main
// stream waits on the counter VALUE (hipStreamWaitValue32): the next group's pivot chain starts ~one tile time after
// the launch does, while the remainder keeps the chip busy -- no separate look-ahead launches competing with the
// update for compute units, no kernel boundary between look-ahead and bulk.
//
// In FRONT of all tiles the same launch carries the group's PANEL: workgroup (row block i, 64 of the 128 nop columns)
// forms G_i = the group's columns of row block i (from the lower triangle: A[i, k] below the group, A[k, i]^T above it)
// and H_i = -G_i Pg (K = 128 nop), stores them in the panel buffers and ticks row block i's counter (2 nop ticks = the
// row block is ready).  A tile (I, J) waits for row blocks I and J before it touches its operands, the write-back of
// row block i waits for row block i (its panel workgroups are the ones that read the OLD columns).  Workgroups are
// dispatched in index order and panel workgroups never wait, so the waits cannot deadlock.  Kind 2 and kind 3 need no
// counter any more: the next group's update is a later launch on the same stream.
// Head part: up to 8 pseudo-columns; entry m covers the tiles (b, col[m]) for n1[m] blocks b from lo1[m] and then
// blocks from lo2[m] on; first[m] .. first[m+1]-1 are its workgroups; entries below ndiag_cols are the diagonal
// super-block.
struct GroupUpd {
    const double *G[4];
    const double *H[4];
    int nop;
    int skip_lo, skip_n;                           // remainder
    int ncol, ndiag;                               // head: entries; workgroups of kind 1
    int col[8], first[9], lo1[8], n1[8], lo2[8];
    int wb_first, wb_rows, wb_b0, wb_sz, nhead;    // kind 2: workgroups [wb_first, nhead): (row block, column w of the group)
    int npanel;                                    // panel workgroups in front of everything: (row block, 64 columns of H)
    double *G0, *H0;                               // the group's panels (panel w at G0 / H0 + w * pstride)
    size_t pstride;
    const double *Pg;                              // inverse of the group's diagonal super-block (ld = 128 wb_sz)
    unsigned *cnt;                                 // cnt[0]: ticks of the kind-1 workgroups; cnt[1 + i]: panel ticks of row block i
    unsigned *next;                                // work counter of the launch: the next work item to hand out
    int total;                                     // work items: npanel + nhead + remainder
};

// One work item (panel workgroup, head tile, write-back tile or remainder tile) of a group's update.
template <bool MULTI>
__device__ __forceinline__ void group_update_item(double *__restrict__ A, size_t ld, size_t pld, const GroupUpd &P, int item,
                                                  double (*Gs)[KC][LDS_LD], double (*Hs)[KC][LDS_LD])
{
    const int tid = opaque_tid(), lane = tid & 63, wv = tid >> 6;
    if (item < P.npanel) {
        // ---- panel workgroup ----
        const int wr = wv & 1, wc = wv >> 1, l15 = lane & 15, lq = lane >> 4;
        const int sz = P.wb_sz, b0 = P.wb_b0;
        const int y = item % (2 * sz), w = y >> 1, ch = y & 1;
        int i = item / (2 * sz);
        const int irow = i;
        if (i >= b0) i += sz;
        const size_t pgld = (size_t)sz * T;
        double4_t acc[2][4];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = (double4_t){0.0, 0.0, 0.0, 0.0};
        // H operand of pivot block v: Pg(c, k) for c = w 128 + ch 64 + .., k = v 128 + ..  (Pg is symmetric)
        const double *hsrc0 = P.Pg + (size_t)w * T + (size_t)ch * 64;
        double *gcopy0 = (y == 0) ? P.G0 + (size_t)i * T : nullptr;
        if (i > b0) {  // below the group: G_i = A[i, k]
#pragma unroll 1
            for (int v = 0; v < sz; ++v)
                tile_product<false, 2>(acc, A + (size_t)i * T + (size_t)(b0 + v) * T * ld, ld, hsrc0 + (size_t)v * T * pgld, pgld, Gs[0],
                                       Hs[0], gcopy0 ? gcopy0 + (size_t)v * P.pstride : nullptr, pld);
        } else {       // above the group: G_i = A[k, i]^T
#pragma unroll 1
            for (int v = 0; v < sz; ++v)
                tile_product<true, 2>(acc, A + (size_t)(b0 + v) * T + (size_t)i * T * ld, ld, hsrc0 + (size_t)v * T * pgld, pgld, Gs[0],
                                      Hs[0], gcopy0 ? gcopy0 + (size_t)v * P.pstride : nullptr, pld);
        }
        double *Hw = P.H0 + (size_t)w * P.pstride + (size_t)i * T;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int r = wr * 64 + tn * 16 + l15;
                    const int c = ch * 64 + wc * 32 + tm * 16 + lq + 4 * reg;
                    Hw[(size_t)r + (size_t)c * pld] = -acc[tm][tn][reg];
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(P.cnt + 1 + irow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    const int t = item - P.npanel;
    const bool head = t < P.nhead;
    // wait until the panels of row blocks ra and rb (indices among the blocks outside the group) are complete
    auto wait_rows = [&](int blk_a, int blk_b) {
        if (tid == 0) {
            const unsigned need = 2u * (unsigned)P.wb_sz;
            const int ra = blk_a >= P.wb_b0 ? blk_a - P.wb_sz : blk_a, rb = blk_b >= P.wb_b0 ? blk_b - P.wb_sz : blk_b;
            while (__hip_atomic_load(P.cnt + 1 + ra, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need ||
                   __hip_atomic_load(P.cnt + 1 + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need)
                __builtin_amdgcn_s_sleep(8);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    };
    int I, J;
    if (head && t >= P.wb_first) {
        // write-back of one tile of the group's new columns
        const int e = t - P.wb_first;
        int i = e % P.wb_rows;
        const int w = e / P.wb_rows;
        if (i >= P.wb_b0) i += P.wb_sz;
        wait_rows(i, i);
        panel_writeback_tile(A, ld, P.wb_b0 + w, i, P.H0 + (size_t)w * P.pstride, pld, Gs[0]);
        I = J = -1;
    } else if (head) {
        int m = 0;
        while (m + 1 < P.ncol && t >= P.first[m + 1]) ++m;
        const int local = t - P.first[m];
        const int b = local < P.n1[m] ? P.lo1[m] + local : P.lo2[m] + (local - P.n1[m]);
        const int cb = P.col[m];
        I = b > cb ? b : cb;
        J = b > cb ? cb : b;
    } else {
        const int tt = t - P.nhead;
        int ii = (int)((sqrt(8.0 * (double)tt + 1.0) - 1.0) * 0.5);
        while ((long long)ii * (ii + 1) / 2 > tt) --ii;
        while ((long long)(ii + 1) * (ii + 2) / 2 <= tt) ++ii;
        int jj = tt - (int)((long long)ii * (ii + 1) / 2);
        if (ii >= P.skip_lo) ii += P.skip_n;
        if (jj >= P.skip_lo) jj += P.skip_n;
        I = ii;
        J = jj;
    }
    if (I >= 0) wait_rows(I, J);
    if (I >= 0) {
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
    if (t < P.ndiag) {
        // publish: every wave's stores have left the CU, then one agent-scope release and the tick
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(P.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// The launch is PERSISTENT: two workgroups per compute unit of the stream's CU mask, each taking work items off one
// device-wide counter until none are left.  Items are handed out in index order (so "panel first, head next" holds as
// it would for dispatch order, and a waiting item can only wait for items that are already running); a compute unit that
// is slower for any reason (a shader engine that lost a CU to the mask and still gets an equal share of a static grid,
// the tail round of a static grid) simply takes fewer items; and the next item's operand loads are issued while the
// previous item's stores drain, with no workgroup launch in between.
template <bool MULTI>
__global__ __launch_bounds__(256, 2) void k_group_update(const GroupUpd Parg, double *__restrict__ A, size_t ld, size_t pld)
{
    __shared__ __attribute__((aligned(16))) double Gs[2][KC][LDS_LD];
    __shared__ __attribute__((aligned(16))) double Hs[2][KC][LDS_LD];
    __shared__ int s_item;
    // the descriptor is read where it lies, in the kernel-argument segment (first argument = offset 0): indexing its
    // arrays with run-time indices through a by-value copy would put the copy into scratch memory
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const GroupUpd __attribute__((address_space(4))) *kernarg_desc_t;
    const GroupUpd &P = *(const GroupUpd *)(kernarg_desc_t)__builtin_amdgcn_kernarg_segment_ptr();
    (void)Parg;
#else
    const GroupUpd &P = Parg;
#endif
    for (;;) {
        if (threadIdx.x == 0) s_item = (int)__hip_atomic_fetch_add(P.next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int item = s_item;
        if (item >= P.total) break;
        // the leading dimensions are made opaque per iteration: otherwise every per-thread address offset of every path
        // (hundreds of 64-bit values) is hoisted out of this loop as loop-invariant and spilled to scratch
        size_t ld_i = ld, pld_i = pld;
        asm volatile("" : "+s"(ld_i), "+s"(pld_i));
        group_update_item<MULTI>(A, ld_i, pld_i, P, item, Gs, Hs);
        __syncthreads();  // LDS and s_item are free again
    }
}

// ---- the look-ahead chain of a pivot GROUP ------------------------------------------------------------------------
// A group of sz <= 4 consecutive pivot blocks (m = 128 sz columns) is swept as ONE pivot of width m:
//       Pg = (A_gg)^-1 (m x m),   G = A_{.,g},   H = -G Pg,   A_ij += H_i G_j^T  (the launches above),   A_{.,g} <- -H,   A_gg <- -Pg.
// Only Pg is serial work, and it is small: the m x m diagonal super-block is copied to a dense scratch matrix
// (k_gather_diag) and swept there block by block -- k_pivot for the 128 x 128 pivot, then two tiny launches of tile
// products (k_tile_jobs) for the other blocks of the scratch matrix -- while the bulk of the group's look-ahead work
// (the previous group's update applied to this group's columns, all rows) runs beside it inside the previous group's
// update launch.  The group's own update launch then forms G and H for every row block in its front part (K = m) and
// writes the new columns back.  Per group: no per-pivot panel / slice launches over the whole matrix any more.

// Sg (m x m, ld = m, full storage) <- the diagonal super-block of A at block b0 (lower triangle authoritative)
__global__ __launch_bounds__(256) void k_gather_diag(const double *__restrict__ A, size_t ld, int b0, int m,
                                                      double *__restrict__ Sg)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= m * m) return;
    const int r = e % m, c = e / m;
    const double *Agg = A + (size_t)b0 * T + (size_t)b0 * T * ld;
    Sg[e] = r >= c ? Agg[(size_t)r + (size_t)c * ld] : Agg[(size_t)c + (size_t)r * ld];
}

// Sg holds -Pg: A_gg <- Sg (lower-triangle tiles, diagonal tiles in full), Pg <- -Sg, exactly symmetric (from the lower
// triangle)
__global__ __launch_bounds__(256) void k_scatter_diag(double *__restrict__ A, size_t ld, int b0, int m,
                                                       const double *__restrict__ Sg, double *__restrict__ Pg)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= m * m) return;
    const int r = e % m, c = e / m;
    if (r < c) return;
    double *Agg = A + (size_t)b0 * T + (size_t)b0 * T * ld;
    const double v = Sg[e];
    Agg[(size_t)r + (size_t)c * ld] = v;
    Pg[(size_t)r + (size_t)c * m] = -v;
    if (r > c) {
        Pg[(size_t)c + (size_t)r * m] = -v;
        if (r / T == c / T) Agg[(size_t)c + (size_t)r * ld] = v;
    }
}

// A handful of independent 128 x 128 x 128 tile products on the scratch matrix, two workgroups (64 columns each) per
// job:   X = g h^T;   out = cin ? cin - X : X;   outT (optional) = out^T.
struct TileJob {
    const double *g, *h, *cin;
    double *out, *outT;
};
struct TileJobs {
    TileJob j[9];
    size_t gld, hld, cld;
};

// The jobs run beside the big update, when a dependent global load takes several microseconds: all of a job's operands
// (K = 128: 8 chunks) are requested up front and the chunks then pass through a double-buffered LDS stage, one barrier
// each -- one memory round trip per job instead of one per chunk.
__global__ __launch_bounds__(256, 1) void k_tile_jobs(const TileJobs J)
{
    __shared__ __attribute__((aligned(16))) double Gs[2][KC][LDS_LD];
    __shared__ __attribute__((aligned(16))) double Hs[2][KC][LDS_LD];
    const TileJob job = J.j[blockIdx.x >> 1];
    const int ch = blockIdx.x & 1;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv & 1, wc = wv >> 1, l15 = lane & 15, lq = lane >> 4;
    constexpr int NCH = T / KC;
    StageRegs<2> R[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) stage_load<false, 2>(R[c], job.g, J.gld, job.h + (size_t)ch * 64, J.hld, c * KC, tid);
    // the tile of cin this thread will combine with, requested now as well
    double cin[2][4][4];
    if (job.cin) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int r = wr * 64 + tn * 16 + l15;
                    const int c = ch * 64 + wc * 32 + tm * 16 + lq + 4 * reg;
                    cin[tm][tn][reg] = job.cin[(size_t)r + (size_t)c * J.cld];
                }
    }
    double4_t acc[2][4];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = (double4_t){0.0, 0.0, 0.0, 0.0};
    stage_store<false, 2>(R[0], Gs[0], Hs[0], tid);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) stage_store<false, 2>(R[c + 1], Gs[(c + 1) & 1], Hs[(c + 1) & 1], tid);
        chunk_mma<2>(acc, Gs[c & 1], Hs[c & 1], wr, wc, lane);
        __syncthreads();
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = wr * 64 + tn * 16 + l15;
                const int c = ch * 64 + wc * 32 + tm * 16 + lq + 4 * reg;
                double v = acc[tm][tn][reg];
                if (job.cin) v = cin[tm][tn][reg] - v;
                job.out[(size_t)r + (size_t)c * J.cld] = v;
                if (job.outT) job.outT[(size_t)c + (size_t)r * J.cld] = v;
            }
}

// Host driver of the block sweep: pivot groups of g blocks (g = 1 .. 4 by matrix size; GDCA_GROUP=g forces).  Per
// group p two things happen:
//     M(p)  Pg = inverse of the group's diagonal super-block: gather, sz x (k_pivot, two tile-job launches), scatter
//     U(p)  ONE launch (k_group_update): panel, look-ahead head, write-back, remainder -- everything else of the group
// With a side stream (s1) they overlap: U(p) runs on the side stream; the main (high-priority) stream waits for U(p)'s
// head counter (the next diagonal super-block carries update p) and runs M(p+1) on the few CUs that are kept free of
// U's workgroups; U(p+1) follows U(p) on the side stream as soon as M(p+1) is done.  Panels are double-buffered by
// group parity.  Without a side stream everything runs on s0 in order.
void gdca_launch_spd_inverse(hipStream_t s0, hipStream_t s1, double *A, int n_pad, const gdca_inverse_ws &ws,
                             gdca_dev_scalars *sc, int n_real, hipEvent_t *sync_ev, hipEvent_t *upd_ev, int max_upd_ev,
                             int *n_upd_launch, double *upd_flops)
{
    const int nblk = n_pad / T;
    const size_t ld = (size_t)n_pad;
    const double tile_flops = 2.0 * T * T * T;
    int nl = 0;
    double fl = 0.0;
    const bool la = s1 != nullptr && nblk >= 3;
    // pivots per group: more pivots per launch raise the update's arithmetic intensity (K = 128 g) and amortise its
    // per-tile overheads; the group's chain grows with g and must stay shorter than the update it hides behind
    static const int group_env = getenv("GDCA_GROUP") ? atoi(getenv("GDCA_GROUP")) : -1;
    int g = group_env >= 1 ? std::min(group_env, 4) : (nblk >= 90 ? 4 : (nblk >= 48 ? 3 : (nblk >= 24 ? 2 : 1)));
    if (nblk < 2 * g) g = 1;
    const int ng = (nblk + g - 1) / g;
    hipEvent_t *Ep = sync_ev, *Eb = sync_ev + ng;
    const size_t pstride = (size_t)(ws.G[1] - ws.G[0]);
    const int cstride = nblk + 2;  // counters per group: [0] head ticks, [1 + i] panel ticks of row block i, [nblk + 1] work counter
    auto base = [&](int p) { return p * g; };
    auto size = [&](int p) { return std::min(g, nblk - p * g); };
    // M: Pg of group p (and A_gg <- -Pg)
    auto super_pivot = [&](int p) {
        const int b0 = base(p), sz = size(p), m = sz * T;
        if (sz == 1) {
            double *Akk = A + (size_t)b0 * T + (size_t)b0 * T * ld;
            hipLaunchKernelGGL(k_pivot, dim3(1), dim3(PIVOT_THREADS), 0, s0, (const double *)Akk, ld, Akk, ld, ws.Pg, (size_t)T, sc,
                               b0 * T, n_real);
            return;
        }
        const unsigned eg = (unsigned)((m * m + 255) / 256);
        hipLaunchKernelGGL(k_gather_diag, dim3(eg), dim3(256), 0, s0, (const double *)A, ld, b0, m, ws.Sg[0]);
        int cur = 0;
        for (int w = 0; w < sz; ++w) {
            const double *Sin = ws.Sg[cur];
            double *Sout = ws.Sg[cur ^ 1];
            const size_t dd = (size_t)w * T + (size_t)w * T * m;
            hipLaunchKernelGGL(k_pivot, dim3(1), dim3(PIVOT_THREADS), 0, s0, Sin + dd, (size_t)m, Sout + dd, (size_t)m, ws.P,
                               (size_t)T, sc, (b0 + w) * T, n_real);
            // the other blocks of the scratch matrix:  S_iw <- S_iw Pw (and its mirror S_wi),  S_ij <- S_ij - (S_iw Pw) S_jw^T
            TileJobs J1{}, J2{};
            int n1 = 0, n2 = 0;
            for (int i = 0; i < sz; ++i) {
                if (i == w) continue;
                TileJob &a = J1.j[n1++];
                a.g = Sin + (size_t)i * T + (size_t)w * T * m;
                a.h = ws.P;
                a.cin = nullptr;
                a.out = Sout + (size_t)i * T + (size_t)w * T * m;
                a.outT = Sout + (size_t)w * T + (size_t)i * T * m;
                for (int j = 0; j < sz; ++j) {
                    if (j == w) continue;
                    TileJob &b = J2.j[n2++];
                    b.g = Sout + (size_t)i * T + (size_t)w * T * m;
                    b.h = Sin + (size_t)j * T + (size_t)w * T * m;
                    b.cin = Sin + (size_t)i * T + (size_t)j * T * m;
                    b.out = Sout + (size_t)i * T + (size_t)j * T * m;
                    b.outT = nullptr;
                }
            }
            J1.gld = (size_t)m;
            J1.hld = (size_t)T;
            J1.cld = (size_t)m;
            J2.gld = J2.hld = J2.cld = (size_t)m;
            hipLaunchKernelGGL(k_tile_jobs, dim3(2 * n1), dim3(256), 0, s0, J1);
            hipLaunchKernelGGL(k_tile_jobs, dim3(2 * n2), dim3(256), 0, s0, J2);
            cur ^= 1;
        }
        hipLaunchKernelGGL(k_scatter_diag, dim3(eg), dim3(256), 0, s0, A, ld, b0, m, (const double *)ws.Sg[cur], ws.Pg);
    };
    // U(p): panel, head (diagonal super-block of group p+1, the other rows of group p+1's columns, write-back of group
    // p's columns), remainder.  Returns the head-counter target.
    auto update = [&](hipStream_t st, int p) -> unsigned {
        const int b0 = base(p), sz = size(p);
        const int nsz = p + 1 < ng ? size(p + 1) : 0, c0 = b0 + sz;
        GroupUpd P{};
        for (int w = 0; w < 4; ++w) {
            P.G[w] = ws.G[4 * (p & 1) + std::min(w, sz - 1)];
            P.H[w] = ws.H[4 * (p & 1) + std::min(w, sz - 1)];
        }
        P.nop = sz;
        int first = 0;
        for (int mm = 0; mm < nsz; ++mm) {  // diagonal super-block of the next group: rows mm .. nsz-1 of column mm
            P.col[mm] = c0 + mm;
            P.first[mm] = first;
            P.lo1[mm] = c0 + mm;
            P.n1[mm] = nsz - mm;
            P.lo2[mm] = 0;
            first += nsz - mm;
        }
        P.ndiag = first;
        const int nrest = nblk - sz - nsz;
        const int ncol_rest = nrest > 0 ? nsz : 0;
        for (int mm = 0; mm < ncol_rest; ++mm) {  // the other rows: above group p, then below group p+1
            const int e = nsz + mm;
            P.col[e] = c0 + mm;
            P.first[e] = first;
            P.lo1[e] = 0;
            P.n1[e] = b0;
            P.lo2[e] = c0 + nsz;
            first += nrest;
        }
        P.ncol = nsz + ncol_rest;
        P.first[P.ncol] = first;
        P.wb_first = first;
        P.wb_rows = nblk - sz;
        P.wb_b0 = b0;
        P.wb_sz = sz;
        P.nhead = first + (nblk - sz) * sz;
        P.npanel = (nblk - sz) * 2 * sz;
        P.G0 = ws.G[4 * (p & 1)];
        P.H0 = ws.H[4 * (p & 1)];
        P.pstride = pstride;
        P.Pg = ws.Pg;
        P.cnt = ws.cnt + (size_t)p * cstride;
        P.next = P.cnt + nblk + 1;
        P.skip_lo = b0;
        P.skip_n = sz + nsz;
        const int mrem = nblk - P.skip_n;
        const long long nbig = mrem > 0 ? (long long)mrem * (mrem + 1) / 2 : 0;
        P.total = (int)(P.npanel + P.nhead + nbig);
        if (P.total == 0) return 0u;
        const unsigned grid = (unsigned)std::min<long long>(P.total, 2 * ws.update_cus);
        const bool tm = upd_ev && 2 * nl + 1 < max_upd_ev;
        if (tm) (void)hipEventRecord(upd_ev[2 * nl], st);
        if (sz > 1)
            hipLaunchKernelGGL((k_group_update<true>), dim3(grid), dim3(256), 0, st, P, A, ld, ld);
        else
            hipLaunchKernelGGL((k_group_update<false>), dim3(grid), dim3(256), 0, st, P, A, ld, ld);
        if (tm) (void)hipEventRecord(upd_ev[2 * nl + 1], st);
        ++nl;
        fl += tile_flops * ((double)(first + nbig) * (double)sz + (double)(nblk - sz) * (double)(sz * sz));
        return (unsigned)P.ndiag;
    };

    (void)hipMemsetAsync(ws.cnt, 0, (size_t)ng * cstride * sizeof(unsigned), s0);
    super_pivot(0);
    if (nblk == size(0)) {  // a single group: nothing to update
        if (n_upd_launch) *n_upd_launch = 0;
        if (upd_flops) *upd_flops = 0.0;
        return;
    }
    if (!la) {
        // serial schedule on s0
        for (int p = 0; p < ng; ++p) {
            if (p > 0) super_pivot(p);
            (void)update(s0, p);
        }
    } else {
        (void)hipEventRecord(Ep[0], s0);
        for (int p = 0; p < ng; ++p) {
            (void)hipStreamWaitEvent(s1, Ep[p], 0);
            const unsigned nA = update(s1, p);
            if (p + 1 < ng) {
                if (nA) (void)hipStreamWaitValue32(s0, ws.cnt + (size_t)p * cstride, nA, hipStreamWaitValueGte, 0xFFFFFFFFu);
                super_pivot(p + 1);
                (void)hipEventRecord(Ep[p + 1], s0);
            }
        }
        (void)hipEventRecord(Eb[0], s1);
        (void)hipStreamWaitEvent(s0, Eb[0], 0);
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
