// Dense SPD inverse  mJ = inv(cholesky(C))  (reference call site src/GaussDCA.jl:34; there:
// LAPACK dpotrf + dpotri, n^3 flops).
//
// MI355X design: a BLOCK SYMMETRIC SWEEP (block Gauss-Jordan on an SPD matrix).  For a pivot block (group) K with
// D = A_KK (the Schur complement at that point -- the same matrix Cholesky would factor, so the
// positive-definiteness test and the `info` index are the same as dpotrf's), P = D^-1, G = A_{.,K}:
//       A_ij <- A_ij - G_i P G_j^T   (i, j != K),   A_{.,K} <- G P,   A_KK <- -P.
// After all pivots A = -C^-1.  Same n^3 flop count as dpotrf+dpotri, but every step is ~(n/128)^2/2 identical 128x128
// tile products over the whole lower triangle: no shrinking trailing matrix, no trtri/lauum dependency chains, no tail
// of tiny steps -- the shape a 256-CU chip wants.  The matrix stays symmetric throughout, only the lower triangle (with
// full diagonal tiles) is touched.  Pivots are taken in groups of up to four 128-blocks (tile products of depth
// K = 128 g per group); the serial part of a group -- the inverse of its 128 g x 128 g diagonal super-block -- runs on a
// small dense scratch copy beside the tile products of the previous group.  The WHOLE inverse is one persistent launch
// (k_sweep, second half of this file): work items off a device-wide counter, dependencies as flags in HBM.
//
// Tile product: 256 threads = 4 waves in 2 x 2, each wave a 64 x 64 sub-tile as 4 x 4
// v_mfma_f64_16x16x4_f64 accumulators (128 VGPRs).  The f64 C/D fragment is
// col = lane & 15, row = (lane >> 4) + 4 * reg; the MFMA "column" index is mapped to the
// matrix ROW (the contiguous direction of the column-major storage), so every accumulator
// load/store instruction moves 4 columns x 128 contiguous bytes.  Operands go through LDS as
// [k][row] with a 128-byte pad per k-row, which puts the four k-rows a ds_read_b64 touches on
// disjoint bank halves (conflict-free).
#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <vector>

#include "gdca_internal.h"
#include "gdca_launch.h"

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
// Pivot: P = inverse of a 128 x 128 SPD block by a BLOCKED symmetric sweep, one 256-thread workgroup, the block's lower
// triangle resident in MFMA accumulators from the first load to the last store.
//
// The block is cut into 8 x 8 micro-blocks of 16 x 16 (36 lower-triangular tiles).  Sweeping micro-block K
//       D_ij <- D_ij - G_i Pm G_j^T   (i, j != K),   D_iK <- G_i Pm,   D_KK <- -Pm,      G = D_{.,K},  Pm = D_KK^-1
// is, for EVERY tile, four v_mfma_f64_16x16x4_f64 on two operand images in LDS:
//       Gs = the old column block K (128 x 16), with the rows of micro-block K replaced by -I
//       Ns = -(G Pm)               (128 x 16), with the rows of micro-block K replaced by +Pm
//       tile(rb, cb) <- [rb == K or cb == K ? 0 : tile] + Gs[rb] Ns[cb]^T
// (the same -1 / -p device the scalar sweep uses for its pivot row and column), so the three kinds of tiles need
// no special code.  The serial part -- Pm = (16 x 16 diagonal tile)^-1, sixteen dependent steps, and the two products that
// bring the NEXT diagonal tile up to date -- runs on a wave of its own (pivot_chain below); the other three waves do every
// other update beside it.
//
// Operand images are [kk][row] with a swizzled row offset (pv_off): the MFMA operand reads (16 consecutive rows of
// two adjacent kk per 32 lanes) and the transposed stores of the tiles left of the diagonal (16 different kk, one
// row) are both (nearly) conflict-free.
// 1/d is v_rcp_f64 + two Newton steps.  A non-positive pivot (the same test dpotrf makes) is reported through
// sc->info.  Writes P (full symmetric, ld = pld) and -P (full tile) to the output tile.
// -------------------------------------------------------------------------------------------------
#define MB 16                      // micro-block edge
#define NMB (T / MB)               // micro-blocks per side
#define PV_ROW 160                 // doubles per kk-row of an operand image
#define SLAB_ITEMS 8               // 16-row slabs of a 128-row block (the chain's row-slab items)
#define SLAB_PD 4                  // their load pipeline: rounds of 16 k in flight (a round's MFMAs take 0.4 us, a load from L2 / HBM one to two)

__device__ __forceinline__ int pv_off(int kk)
{
    return kk * PV_ROW + 16 * (kk & 1) + 2 * (kk >> 1);
}

template <int L>
__device__ __forceinline__ double read_lane_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), L);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), L);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

#ifdef GDCA_PIVOT_STAMPS
// tools/test_pivot.hip only: shader-clock stamps of the phases of pivot_chain, [micro-block + 1][phase][wave]
__device__ long long g_pivot_stamps[9 * 8 * 12];
#define PV_STAMP(K, ph)                                                                      \
    do {                                                                                      \
        if (lane == 0) g_pivot_stamps[(((K) + 1) * 8 + (ph)) * 12 + wv] = (long long)clock64(); \
    } while (0)
#else
#define PV_STAMP(K, ph) do { } while (0)
#endif

// ---- the form the persistent sweep kernel runs: ONE 256-thread workgroup, the serial chain on a wave of its own ----
// The only serial part of the blocked sweep is  micro-pivot(K) -> N_{K+1} = -(G_{K+1} Pm) -> tile (K+1, K+1) += G_{K+1} N_{K+1}^T ->
// micro-pivot(K+1),  and all of it stays inside ONE wave's registers: -Pm in accumulator layout IS the MFMA operand of the first
// product, its result IS the operand of the second (out(l15, lq + 4 reg) = sum_kk b(l15, kk) a(lq + 4 reg, kk) with operand element
// (l15, kk = lq + 4 t4) in register t4: operand layout = accumulator layout).  So wave 3 (the chain wave) does nothing else, and
// waves 0-2 (the workers, 12 of the 36 lower-triangular micro-tiles each) do everything that is not on the chain beside it:
//      window 1 (after barrier 1):  chain: the two products for tile (K+1, K+1)      workers: Ns = -(G Pm), rows K := +Pm; Gs rows K := -I
//      window 2 (after barrier 2):  chain: micro-pivot(K+1), -Pm(K+1) -> Pms         workers: 12 tile updates each, then column block
//                                                                                     K+1 -> the OTHER Gs image, tile (K+2, K+2) -> Dt
// Two barriers per micro-block; per micro-block the longer of (micro-pivot, 12 tile updates) instead of their sum plus the rest.
// LDS (doubles): bufA = Gs image 0 | Ns rows kk < 8 | Pms | Dt | flag,  bufB = Gs image 1 | Ns rows kk >= 8  (two buffers of
// 2 KC LDS_LD doubles: the staging arrays of the tile paths).
#define PVC_NS_OFF (MB * PV_ROW)
#define PVC_PMS_OFF (PVC_NS_OFF + (MB / 2) * PV_ROW)
#define PVC_DT_OFF (PVC_PMS_OFF + MB * MB)
#define PVC_FLAG_OFF (PVC_DT_OFF + MB * MB)
#define PVC_NT 12

// One step of the 16 x 16 sweep as ONE MFMA, bad pivots remembered in a scalar (first one wins).
template <int JJ>
struct ChainStep {
    static __device__ __forceinline__ void run(double4_t &v, int l15, int lq, int index_base, int &bad)
    {
        // lanes with lq == JJ & 3 hold column JJ, D[l15][JJ] (= D[JJ][l15] up to rounding), in register JJ >> 2: they are the
        // k = JJ & 3 slice of both operands, the other slices are zero:  v <- [row or column JJ ? 0 : v] - u w^T,
        // u_i = (i == JJ ? -1 : D[i][JJ]),  w_j = (j == JJ ? -p : D[j][JJ] p)
        const double col = v[JJ >> 2];
        const double d = read_lane_f64<(JJ & 3) * 16 + JJ>(col);
        bad = (bad == 0 && !(d > 0.0)) ? index_base + JJ + 1 : bad;
        double p = __builtin_amdgcn_rcp(d);
        p = fma(p, fma(-d, p, 1.0), p);
        p = fma(p, fma(-d, p, 1.0), p);
        const bool sel = lq == (JJ & 3);
        const double cm = (l15 == JJ) ? -1.0 : col;
        const double b = sel ? -cm : 0.0;      // -u_i, i = l15
        const double a = sel ? cm * p : 0.0;   // w_j, j = l15
        double4_t c;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) c[reg] = (l15 == JJ || (reg == (JJ >> 2) && sel)) ? 0.0 : v[reg];
        v = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        if constexpr (JJ + 1 < MB) ChainStep<JJ + 1>::run(v, l15, lq, index_base, bad);
    }
};

// tile e of the lower triangle <-> (rb, cb), rb >= cb, e = rb (rb + 1) / 2 + cb; worker W owns e = 3 t + W, t = 0 .. 11 (compile-time
// per worker: the LDS addresses of its operand reads are immediates, and nothing per tile stays in registers but the tile)
__device__ __forceinline__ constexpr int pvc_rb(int e)
{
    int r = 0;
    while ((r + 1) * (r + 2) / 2 <= e) ++r;
    return r;
}
__device__ __forceinline__ constexpr int pvc_cb(int e)
{
    return e - pvc_rb(e) * (pvc_rb(e) + 1) / 2;
}

// bufA = base, bufB = base + 2 KC LDS_LD: ONE LDS array (so that every access stays a ds_ instruction off one base register)
#define PVC_B_OFF (2 * KC * LDS_LD)
struct PivotBufs {
    double *A;
    __device__ __forceinline__ double *gs(int K) const { return A + (K & 1) * PVC_B_OFF; }
    // Ns(., kk): the half (kk >= 8) is known at compile time at every use
    __device__ __forceinline__ double *ns(int hi, int kk) const
    {
        return A + hi * PVC_B_OFF + PVC_NS_OFF + (kk & 7) * PV_ROW + 16 * (kk & 1) + 2 * (kk >> 1);
    }
    __device__ __forceinline__ double *pms() const { return A + PVC_PMS_OFF; }
    __device__ __forceinline__ double *dt() const { return A + PVC_DT_OFF; }
};

__device__ __forceinline__ double4_t pvc_load_tile(const double *Ain, size_t ldin, int rb, int cb, int l15, int lq)
{
    double4_t x;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        int r = MB * rb + l15, c = MB * cb + lq + 4 * reg;
        if (r < c) {
            const int y = r;
            r = c;
            c = y;
        }
        x[reg] = Ain[(size_t)r + (size_t)c * ldin];
    }
    return x;
}

template <class Mid>
__device__ __forceinline__ void pivot_chain_wave(const double *Ain, size_t ldin, const PivotBufs L, int lane, int *bad_out, Mid &mid)
{
    const int l15 = lane & 15, lq = lane >> 4;
    [[maybe_unused]] const int wv = 3;
    double *Pms = L.pms();
    const double *Dt = L.dt();
    double4_t v = (double4_t){0.0, 0.0, 0.0, 0.0};
    double4_t dt = pvc_load_tile(Ain, ldin, 0, 0, l15, lq);
    int bad = 0;
#pragma unroll 1
    for (int K = -1; K < NMB; ++K) {
        PV_STAMP(K, 0);
        if (K >= 0) {
            __syncthreads();  // barrier 1
            PV_STAMP(K, 1);
            if (K + 1 < NMB) {
                const double *G = L.gs(K);
                double bb[4];
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) bb[t4] = G[pv_off(4 * t4 + lq) + MB * (K + 1) + l15];
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) dt[reg] = Dt[64 * reg + lane];
                double4_t g = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) g = __builtin_amdgcn_mfma_f64_16x16x4f64(v[t4], bb[t4], g, 0, 0, 0);
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) dt = __builtin_amdgcn_mfma_f64_16x16x4f64(g[t4], bb[t4], dt, 0, 0, 0);
            }
            PV_STAMP(K, 2);
            __syncthreads();  // barrier 2
            PV_STAMP(K, 3);
        }
        if (K + 1 < NMB) {
            ChainStep<0>::run(dt, l15, lq, MB * (K + 1), bad);
            v = dt;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) Pms[(lq + 4 * reg) * MB + l15] = v[reg];  // Pms[j][i] = -Pm(i, j)
        }
        PV_STAMP(K, 4);
    }
    if (lane == 0 && bad != 0) *bad_out = bad;
    __syncthreads();  // the workers' "LDS is free" barrier
    mid();
}

__device__ __forceinline__ void store_wt(double *p, double v);

template <int W, class Mid>
__device__ __forceinline__ void pivot_worker(const double *Ain, size_t ldin, double *Aout, size_t ldout, double *__restrict__ P,
                                             size_t pld, const PivotBufs L, int lane, Mid &mid, unsigned long long *ph = nullptr)
{
    const int l15 = lane & 15, lq = lane >> 4;
    [[maybe_unused]] const int wv = W;
    if (W == 0 && ph && lane == 0) ph[0] = wall_clock64();
    const double *Pms = L.pms();
    double *Dt = L.dt();
    double4_t acc[PVC_NT];
#pragma unroll
    for (int t = 0; t < PVC_NT; ++t) acc[t] = pvc_load_tile(Ain, ldin, pvc_rb(3 * t + W), pvc_cb(3 * t + W), l15, lq);
    // tile t, final for column block Kn, into image Kn ([kk][row]: tiles below the diagonal as they are, tiles of row Kn
    // transposed), and tile (Kn + 1, Kn + 1) -- what the chain wave completes and inverts in the next round -- into Dt
    auto stage_column = [&](const int Kn) {  // Kn: a literal at every call (the conditions fold, only the stores remain)
        double *Gn = L.gs(Kn);
#pragma unroll
        for (int t = 0; t < PVC_NT; ++t) {
            const int rb = pvc_rb(3 * t + W), cb = pvc_cb(3 * t + W);
            if (cb == Kn && rb > Kn) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) Gn[pv_off(lq + 4 * reg) + MB * rb + l15] = acc[t][reg];
            } else if (rb == Kn && cb < Kn) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) Gn[pv_off(l15) + MB * cb + lq + 4 * reg] = acc[t][reg];
            } else if (rb == Kn + 1 && cb == Kn + 1) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) Dt[64 * reg + lane] = acc[t][reg];
            }
        }
    };
#pragma unroll 1
    for (int K = -1; K < NMB; ++K) {
        PV_STAMP(K, 0);
        if (K >= 0) {
            double *G = L.gs(K);
            __syncthreads();  // barrier 1: Pms(K), image K and Dt are in LDS
            PV_STAMP(K, 1);
            if (W == 0 && ph && lane == 0 && K == 0) ph[1] = wall_clock64();
            // Ns = -(G Pm) for the row blocks W, W + 3, W + 6 (independent MFMA chains, interleaved); the one of block K itself is
            // computed on whatever the image holds there and dropped
            {
                constexpr int NR = (NMB - W + 2) / 3;
                double4_t g[NR];
#pragma unroll
                for (int i = 0; i < NR; ++i) g[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    const int kk = 4 * t4 + lq;
                    const double a = Pms[kk * MB + l15];
#pragma unroll
                    for (int i = 0; i < NR; ++i) g[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, G[pv_off(kk) + MB * (W + 3 * i) + l15], g[i], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < NR; ++i)
                    if (W + 3 * i != K) {
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) L.ns(reg >> 1, lq + 4 * reg)[MB * (W + 3 * i) + l15] = g[i][reg];
                    }
            }
            if (W == 2) {
                // the rows of micro-block K: -I in the G image, +Pm in Ns
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int kk = lq + 4 * reg;
                    G[pv_off(kk) + MB * K + l15] = (l15 == kk) ? -1.0 : 0.0;
                    L.ns(reg >> 1, kk)[MB * K + l15] = -Pms[kk * MB + l15];
                }
            }
            PV_STAMP(K, 2);
            __syncthreads();  // barrier 2: Ns(K) complete
            PV_STAMP(K, 3);
            // tile <- [in row or column K ? 0 : tile] + Gs[rb] Ns[cb]^T, two tiles at a time (independent MFMA chains)
#pragma unroll
            for (int t = 0; t < PVC_NT; t += 2) {
                const int rb0 = pvc_rb(3 * t + W), cb0 = pvc_cb(3 * t + W), rb1 = pvc_rb(3 * t + 3 + W), cb1 = pvc_cb(3 * t + 3 + W);
                const bool f0 = rb0 == K || cb0 == K, f1 = rb1 == K || cb1 == K;
                double4_t c0, c1;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    c0[reg] = f0 ? 0.0 : acc[t][reg];
                    c1[reg] = f1 ? 0.0 : acc[t + 1][reg];
                }
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    const int kk = 4 * t4 + lq;
                    const double *N = L.ns(t4 >> 1, kk);
                    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(N[MB * cb0 + l15], G[pv_off(kk) + MB * rb0 + l15], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(N[MB * cb1 + l15], G[pv_off(kk) + MB * rb1 + l15], c1, 0, 0, 0);
                }
                acc[t] = c0;
                acc[t + 1] = c1;
            }
        }
        // column block K + 1 is final: into the other image, one straight-line store sequence per value of K
        switch (K + 1) {
        case 0: stage_column(0); break;
        case 1: stage_column(1); break;
        case 2: stage_column(2); break;
        case 3: stage_column(3); break;
        case 4: stage_column(4); break;
        case 5: stage_column(5); break;
        case 6: stage_column(6); break;
        case 7: stage_column(7); break;
        default: break;
        }
        PV_STAMP(K, 5);
    }
    if (W == 0 && ph && lane == 0) ph[2] = wall_clock64();
    __syncthreads();  // the last update's operand reads are done (the chain wave joins): LDS is free
    // D = -inverse (lower-triangular tiles).  P = -D and the output tile = D, both as full symmetric matrices; write-through stores
    // (store_wt): the caller publishes them by draining its stores, without an L2 write-back.  The mirror image of a tile goes through
    // a 16 x 17 scratch of this wave's in LDS, so that its stores are 128-byte row segments too (element by element they were 64
    // different cache lines per instruction and a third of the whole pivot item)
    double *X = L.A + W * (MB * (MB + 1));
    // first P, then mid() -- the caller may publish P there: the next pivot row's items wait for nothing else --, then the output tile
    auto emit = [&](double *O, size_t ldo, const double sgn) {
#pragma unroll
        for (int t = 0; t < PVC_NT; ++t) {
            const int rb = pvc_rb(3 * t + W), cb = pvc_cb(3 * t + W);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int j = lq + 4 * reg;
                const double x = sgn * acc[t][reg];
                X[j * (MB + 1) + l15] = x;
                if (rb > cb || l15 >= j) store_wt(&O[(size_t)(MB * rb + l15) + (size_t)(MB * cb + j) * ldo], x);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int j = lq + 4 * reg;
                const double y = X[l15 * (MB + 1) + j];  // tile element (row j, column l15): to (16 cb + l15, 16 rb + j)
                if (rb > cb || l15 < j) store_wt(&O[(size_t)(MB * cb + l15) + (size_t)(MB * rb + j) * ldo], y);
            }
        }
    };
    emit(P, pld, -1.0);
    mid();
    emit(Aout, ldout, 1.0);
}

// Ain (ld = ldin) is read, Aout (ld = ldout) receives -P, P (ld = pld) the inverse; Ain and Aout may be the same tile.  buf:
// 4 KC LDS_LD doubles of LDS; *bad_out (zeroed by the caller before a barrier) receives the 1-based local index of the first
// non-positive pivot.  The caller puts a barrier behind the call before it reads *bad_out or reuses the buffers.  mid(): called once by
// every wave, between the stores of P and those of the output tile (it may hold a barrier).
template <class Mid>
__device__ __forceinline__ void pivot_chain(const double *Ain, size_t ldin, double *Aout, size_t ldout, double *__restrict__ P,
                                            size_t pld, double *buf, int *bad_out, Mid mid, unsigned long long *ph = nullptr)
{
    static_assert(PVC_FLAG_OFF < 2 * KC * LDS_LD && PVC_NS_OFF + (MB / 2) * PV_ROW <= 2 * KC * LDS_LD, "pivot images fit the staging buffers");
    const int tid = opaque_tid(), lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const PivotBufs L{buf};
    if (wv == 3)
        pivot_chain_wave(Ain, ldin, L, lane, bad_out, mid);
    else if (wv == 0)
        pivot_worker<0>(Ain, ldin, Aout, ldout, P, pld, L, lane, mid, ph);
    else if (wv == 1)
        pivot_worker<1>(Ain, ldin, Aout, ldout, P, pld, L, lane, mid);
    else
        pivot_worker<2>(Ain, ldin, Aout, ldout, P, pld, L, lane, mid);
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
// Addresses of a thread's accumulator elements inside a 128 x 128 tile of a column-major matrix: element (tm, tn, reg) of the wave
// at (wr, wc) is row wr 64 + 16 tn + l15, column wc 64 + 16 tm + lq + 4 reg.  Kept as FOUR 32-bit byte offsets (one per reg; tm
// moves a wave-uniform base, tn is an immediate): the C pieces' loads and the tile's final stores take the scalar-base form, and
// what has to survive the chunk loop is four registers -- as 64-bit addresses these were sixteen register pairs, twelve of
// which hipcc parked in scratch across the loop.
struct TileOff {
    unsigned o[4];
};
__device__ __forceinline__ TileOff tile_off(size_t ld, int wr, int wc, int l15, int lq)
{
    TileOff t;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) t.o[reg] = ((unsigned)(wr * 64 + l15) + (unsigned)(wc * 64 + lq + 4 * reg) * (unsigned)ld) * 8u;
    return t;
}
// the 16-column strip tm of the tile at At (wave-uniform)
__device__ __forceinline__ const char *tile_strip(const double *At, size_t ld, int tm)
{
    return reinterpret_cast<const char *>(At + (size_t)(16 * tm) * ld);
}

template <int CI>
__device__ __forceinline__ void cpiece_load(double (&cp)[8], const double *__restrict__ At, size_t ld, const TileOff &to)
{
    constexpr int tm = CI / 2;
    const char *base = tile_strip(At, ld, tm);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            constexpr int tn0 = 2 * (CI % 2);
            cp[h * 4 + reg] = reinterpret_cast<const double *>(base + to.o[reg])[16 * (tn0 + h)];
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

typedef double (*lds_chunk_t)[LDS_LD];

// ---- the tile item's k loop, scheduled by hand --------------------------------------------------------------------------------
// One k-chunk (16 deep) of a 128 x 128 tile product is, per wave, 64 MFMAs of 64 matrix-pipe clocks each.  Everything else a
// chunk needs -- 16 LDS fragment reads, the 8 register -> LDS staging stores of the NEXT chunk, the 8 global loads of the chunk
// after it, a piece of the C tile -- has to be issued in the shadow of those MFMAs: any stretch of more than ~64 clocks of
// other instructions between two MFMAs of a wave leaves the pipe to the co-resident workgroup's wave alone, and when that one is
// in the same phase the pipe idles.  hipcc's own schedule put the 8 stores, the address arithmetic and the 8 loads in ONE
// block after the first 16 MFMAs of a chunk and started every chunk with `s_waitcnt lgkmcnt(0)` on four fresh LDS reads.  Here:
//   * the fragments of a chunk's first k4 step are read right after the barrier that completes the chunk's LDS buffer,
//     underneath the last 16 MFMAs of the chunk before (carried across the loop edge in `f`);
//   * the staging goes unit by unit between PAIRS of MFMAs of the second k4 step: store unit u to LDS, then at once load unit u
//     of the chunk after next into the same registers (prefetch distance = one whole chunk per unit);
//   * `sched_barrier(0)` pins that order (nothing moves across it; the waitcnt pass runs later and counts exactly).
struct Frag {
    double a[4], b[4];
};

// Where a chunk's operands come from.  G(r, k), r = 0..127 operand rows, k = the chunk's 16 columns:
//   GT = false: G(r, k) = g[r + k * gld]            (panels; tiles of A below the pivot group)
//   GT = true:  G(r, k) = g[k + r * gld]            (tiles of A above the pivot group: the stored tile is the transpose)
// H(c, k) = h[c + k * hld] always.  goff / hoff: this thread's byte offset inside a staging unit.
struct ChunkIO {
    size_t gld, hld;
    unsigned goff, hoff;
    size_t gcopy_ld;  // GCOPY: leading dimension of the untransposed copy of G that the panel items keep
};

template <bool GT>
__device__ __forceinline__ ChunkIO chunk_io(size_t gld, size_t hld, int tid, size_t gcopy_ld = 0)
{
    ChunkIO io;
    io.gld = gld;
    io.hld = hld;
    io.goff = GT ? ((unsigned)((tid & 7) * 2) + (unsigned)(tid >> 3) * (unsigned)gld) * 8u
                 : ((unsigned)((tid & 63) * 2) + (unsigned)(tid >> 6) * (unsigned)gld) * 8u;
    io.hoff = ((unsigned)((tid & 63) * 2) + (unsigned)(tid >> 6) * (unsigned)hld) * 8u;
    io.gcopy_ld = gcopy_ld;
    return io;
}

// The LDS image of a G chunk: [k][row] with LDS_LD doubles per k-row (GT = false), or, when the source is transposed, [row][k]
// with GT_LD doubles per row -- 16-byte global loads of two k values go to LDS as they are, and the fragment reads of a half-wave
// (16 rows x 2 k values) fall on 32 distinct bank pairs (18 doubles = 36 words per row).  Same 18 KB either way.
#define GT_LD 18
static_assert(T * GT_LD <= KC * LDS_LD, "transposed G image fits the chunk buffer");

typedef const volatile double __attribute__((address_space(3))) *lds_cvd_t;  // (an LDS address, said so: a volatile access through a generic pointer is a flat one)

template <bool GT>
__device__ __forceinline__ void frag_read(Frag &f, const double (*Gs)[LDS_LD], const double (*Hs)[LDS_LD], int k4, int wr, int wc,
                                          int l15, int lq)
{
    // (volatile: single ds_read_b64 with 16-bit immediate offsets off ONE address register per operand.  Left to itself hipcc pairs
    // them into ds_read2_b64, whose 8-bit offsets reach 2 KB: it then keeps an address register per LDS buffer and k4 step -- a dozen
    // registers of a kernel that has none to spare -- and parks staging registers in scratch inside the chunk loop instead)
#pragma unroll
    for (int t = 0; t < 4; ++t) f.a[t] = *(lds_cvd_t)&Hs[k4 + lq][wc * 64 + t * 16 + l15];
    if constexpr (GT) {
        const double *Gt = &Gs[0][0];
#pragma unroll
        for (int t = 0; t < 4; ++t) f.b[t] = *(lds_cvd_t)&Gt[(wr * 64 + t * 16 + l15) * GT_LD + k4 + lq];
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) f.b[t] = *(lds_cvd_t)&Gs[k4 + lq][wr * 64 + t * 16 + l15];
    }
}

// MFMA number I of a k4 step, in an order that starts with accumulator row TM0 (the C piece added after this step belongs to
// that row: its MFMAs are long finished when the VALU adds read the accumulators)
template <int I, int TM0>
__device__ __forceinline__ void mma_one(double4_t (&acc)[4][4], const Frag &f)
{
    constexpr int tm = (TM0 + I / 4) & 3, tn = I & 3;
    acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a[tm], f.b[tn], acc[tm][tn], 0, 0, 0);
}

template <int TM0>
__device__ __forceinline__ void mma_all(double4_t (&acc)[4][4], const Frag &f)
{
    mma_one<0, TM0>(acc, f);
    mma_one<1, TM0>(acc, f);
    mma_one<2, TM0>(acc, f);
    mma_one<3, TM0>(acc, f);
    mma_one<4, TM0>(acc, f);
    mma_one<5, TM0>(acc, f);
    mma_one<6, TM0>(acc, f);
    mma_one<7, TM0>(acc, f);
    mma_one<8, TM0>(acc, f);
    mma_one<9, TM0>(acc, f);
    mma_one<10, TM0>(acc, f);
    mma_one<11, TM0>(acc, f);
    mma_one<12, TM0>(acc, f);
    mma_one<13, TM0>(acc, f);
    mma_one<14, TM0>(acc, f);
    mma_one<15, TM0>(acc, f);
}

// staging unit U of a chunk: one 16-byte load / one ds_write_b128 per thread.  U = 4..7: k-rows 4 (U-4) .. of the H chunk (a
// wave moves one contiguous k-row of 128 operand rows).  U = 0..3: the same for the G chunk, or (GT) operand rows 32 U .. 32 U + 31
// of the G chunk, two k values per thread.  gc != nullptr: the G unit also goes to gc[r + k * gcopy_ld] (k relative to the chunk).
template <int U, bool GT>
__device__ __forceinline__ void stage_unit_store(const StageRegs<4> &R, double (*Gs)[LDS_LD], double (*Hs)[LDS_LD], int tid,
                                                 double *__restrict__ gc, size_t gcopy_ld)
{
    if constexpr (U >= 4) {
        const int kk = (tid >> 6) + 4 * (U - 4), r2 = tid & 63;
        *reinterpret_cast<double2 *>(&Hs[kk][r2 * 2]) = make_double2(R.h[2 * (U - 4)], R.h[2 * (U - 4) + 1]);
    } else if constexpr (GT) {
        const int r = (tid >> 3) + 32 * U, k2 = tid & 7;
        *reinterpret_cast<double2 *>(&Gs[0][0] + r * GT_LD + 2 * k2) = make_double2(R.g[2 * U], R.g[2 * U + 1]);
        if (gc) {
            gc[(size_t)r + (size_t)(2 * k2) * gcopy_ld] = R.g[2 * U];
            gc[(size_t)r + (size_t)(2 * k2 + 1) * gcopy_ld] = R.g[2 * U + 1];
        }
    } else {
        const int kk = (tid >> 6) + 4 * U, r2 = tid & 63;
        *reinterpret_cast<double2 *>(&Gs[kk][r2 * 2]) = make_double2(R.g[2 * U], R.g[2 * U + 1]);
        if (gc)
            *reinterpret_cast<double2 *>(gc + (size_t)(r2 * 2) + (size_t)kk * gcopy_ld) = make_double2(R.g[2 * U], R.g[2 * U + 1]);
    }
}

// g, h: the sources at the chunk's first k column
template <int U, bool GT>
__device__ __forceinline__ void stage_unit_load(StageRegs<4> &R, const double *__restrict__ g, const double *__restrict__ h,
                                                const ChunkIO &io)
{
    if constexpr (U >= 4) {
        const char *base = reinterpret_cast<const char *>(h + (size_t)(4 * (U - 4)) * io.hld);
        const double2 v = *reinterpret_cast<const double2 *>(base + io.hoff);
        R.h[2 * (U - 4)] = v.x;
        R.h[2 * (U - 4) + 1] = v.y;
    } else {
        const char *base = reinterpret_cast<const char *>(g + (size_t)((GT ? 32 : 4) * U) * io.gld);
        const double2 v = *reinterpret_cast<const double2 *>(base + io.goff);
        R.g[2 * U] = v.x;
        R.g[2 * U + 1] = v.y;
    }
}

// a whole chunk at once (item prologues)
template <bool GT>
__device__ __forceinline__ void stage_chunk_load(StageRegs<4> &R, const double *__restrict__ g, const double *__restrict__ h,
                                                 const ChunkIO &io)
{
    stage_unit_load<0, GT>(R, g, h, io);
    stage_unit_load<1, GT>(R, g, h, io);
    stage_unit_load<2, GT>(R, g, h, io);
    stage_unit_load<3, GT>(R, g, h, io);
    stage_unit_load<4, GT>(R, g, h, io);
    stage_unit_load<5, GT>(R, g, h, io);
    stage_unit_load<6, GT>(R, g, h, io);
    stage_unit_load<7, GT>(R, g, h, io);
}

template <bool GT>
__device__ __forceinline__ void stage_chunk_store(const StageRegs<4> &R, double (*Gs)[LDS_LD], double (*Hs)[LDS_LD], int tid,
                                                  double *__restrict__ gc, size_t gcopy_ld)
{
    stage_unit_store<0, GT>(R, Gs, Hs, tid, gc, gcopy_ld);
    stage_unit_store<1, GT>(R, Gs, Hs, tid, gc, gcopy_ld);
    stage_unit_store<2, GT>(R, Gs, Hs, tid, gc, gcopy_ld);
    stage_unit_store<3, GT>(R, Gs, Hs, tid, gc, gcopy_ld);
    stage_unit_store<4, GT>(R, Gs, Hs, tid, gc, gcopy_ld);
    stage_unit_store<5, GT>(R, Gs, Hs, tid, gc, gcopy_ld);
    stage_unit_store<6, GT>(R, Gs, Hs, tid, gc, gcopy_ld);
    stage_unit_store<7, GT>(R, Gs, Hs, tid, gc, gcopy_ld);
}

template <int U, bool STORE, bool LOAD, bool GT>
__device__ __forceinline__ void stage_unit(double4_t (&acc)[4][4], const Frag &f, StageRegs<4> &R, double (*Gn)[LDS_LD],
                                           double (*Hn)[LDS_LD], const double *__restrict__ g, const double *__restrict__ h,
                                           const ChunkIO &io, int tid, double *__restrict__ gc)
{
    mma_one<2 * U, 0>(acc, f);
    mma_one<2 * U + 1, 0>(acc, f);
    if constexpr (STORE) stage_unit_store<U, GT>(R, Gn, Hn, tid, gc, io.gcopy_ld);
    if constexpr (LOAD) stage_unit_load<U, GT>(R, g, h, io);
    __builtin_amdgcn_sched_barrier(0);
}

// One chunk.  On entry: `f` = fragments of (Gc, Hc) at k4 = 0; R = the next chunk (loads in flight or landed).
// STORE: R -> (Gn, Hn) (gc != nullptr: its G part also to gc);  LOAD: R <- the chunk at (g, h);  NEXT: on exit `f` = fragments of
// (Gn, Hn) at k4 = 0.
// CI: 0..7 = this chunk also adds C piece CI (requested one chunk earlier; piece 0 in the item's prologue) and requests piece
// CI + 1; -1 = no C traffic.
template <bool STORE, bool LOAD, bool NEXT, int CI, bool GT = false>
__device__ __forceinline__ void tile_chunk(double4_t (&acc)[4][4], Frag &f, StageRegs<4> &R, double (&cp)[8], double (*Gc)[LDS_LD],
                                           double (*Hc)[LDS_LD], double (*Gn)[LDS_LD], double (*Hn)[LDS_LD],
                                           const double *__restrict__ g, const double *__restrict__ h, const ChunkIO &io,
                                           const double *__restrict__ At, size_t ld, int tid, double *__restrict__ gc = nullptr,
                                           const TileOff &to = TileOff{})
{
    const int lane = tid & 63, wv = tid >> 6, wr = wv & 1, wc = wv >> 1, l15 = lane & 15, lq = lane >> 4;
    Frag f1;
    frag_read<GT>(f1, Gc, Hc, 4, wr, wc, l15, lq);
    constexpr int TM0 = CI >= 0 ? (CI / 2) & 3 : 0;
    mma_all<TM0>(acc, f);  // k4 = 0
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (CI >= 0) cpiece_add<(CI >= 0 ? CI : 0)>(acc, cp);
    if constexpr (CI >= 0 && CI < 7) cpiece_load<(CI >= 0 && CI < 7 ? CI + 1 : 0)>(cp, At, ld, to);
    frag_read<GT>(f, Gc, Hc, 8, wr, wc, l15, lq);
    __builtin_amdgcn_sched_barrier(0);
    stage_unit<0, STORE, LOAD, GT>(acc, f1, R, Gn, Hn, g, h, io, tid, gc);  // k4 = 4, two MFMAs per unit
    stage_unit<1, STORE, LOAD, GT>(acc, f1, R, Gn, Hn, g, h, io, tid, gc);
    stage_unit<2, STORE, LOAD, GT>(acc, f1, R, Gn, Hn, g, h, io, tid, gc);
    stage_unit<3, STORE, LOAD, GT>(acc, f1, R, Gn, Hn, g, h, io, tid, gc);
    stage_unit<4, STORE, LOAD, GT>(acc, f1, R, Gn, Hn, g, h, io, tid, gc);
    stage_unit<5, STORE, LOAD, GT>(acc, f1, R, Gn, Hn, g, h, io, tid, gc);
    stage_unit<6, STORE, LOAD, GT>(acc, f1, R, Gn, Hn, g, h, io, tid, gc);
    stage_unit<7, STORE, LOAD, GT>(acc, f1, R, Gn, Hn, g, h, io, tid, gc);
    frag_read<GT>(f1, Gc, Hc, 12, wr, wc, l15, lq);
    __builtin_amdgcn_sched_barrier(0);
    mma_all<0>(acc, f);  // k4 = 8; the last fragment reads of this buffer land underneath
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();     // (Gc, Hc) is free, (Gn, Hn) is complete
    if constexpr (NEXT) frag_read<GT>(f, Gn, Hn, 0, wr, wc, l15, lq);
    __builtin_amdgcn_sched_barrier(0);
    mma_all<0>(acc, f1);  // k4 = 12
    __builtin_amdgcn_sched_barrier(0);
}

// The panel product of a row block for ONE pivot block's 128 columns of H, K = 128 sz, on the tile items' chunk loop:
//   acc(r, c) = sum_v sum_k G_v(r, k) Pg(c, 128 v + k)
// gA: the row block's tile of the group's FIRST pivot block in A -- G_v(r, k) = gA[r + (128 v + k) ld] below the group, and
// (GT) gA[128 v + k + r ld] above it, where the stored tile is the transpose.  hP: Pg at (row c = 0 of the wanted columns, k = 0).
// gcopy: the item also leaves the untransposed G_v in the panel buffers (gcopy + v pstride, ld = ld).
template <bool GT>
__device__ __forceinline__ void panel_chunks(double4_t (&acc)[4][4], const double *__restrict__ gA, size_t ld,
                                             const double *__restrict__ hP, size_t pgld, int sz, double (*Gs)[KC][LDS_LD],
                                             double (*Hs)[KC][LDS_LD], double *__restrict__ gcopy, size_t pstride)
{
    const int tid = opaque_tid(), lane = tid & 63, wv = tid >> 6;
    const int wr = wv & 1, wc = wv >> 1, l15 = lane & 15, lq = lane >> 4;
    const ChunkIO io = chunk_io<GT>(ld, pgld, tid, ld);
    const int nch = (T / KC) * sz;  // chunk c: k = 16 c .. 16 c + 15 of the group's 128 sz columns
    auto gsrc = [&](int c) { return GT ? gA + (size_t)c * KC : gA + (size_t)c * KC * ld; };
    auto hsrc = [&](int c) { return hP + (size_t)c * KC * pgld; };
    auto gcp = [&](int c) { return gcopy ? gcopy + (size_t)(c >> 3) * pstride + (size_t)((c & 7) * KC) * ld : nullptr; };
    // (a second staging register set -- two chunks of prefetch distance -- was tried for these items, which often run alone on
    // their compute unit with chunks shorter than a load's round trip: no gain, and the transposed-G loop then spills)
    StageRegs<4> R;
    double cp[8];
    Frag f;
    stage_chunk_load<GT>(R, gsrc(0), hsrc(0), io);
    stage_chunk_store<GT>(R, Gs[0], Hs[0], tid, gcp(0), ld);
    stage_chunk_load<GT>(R, gsrc(1), hsrc(1), io);
    __syncthreads();
    frag_read<GT>(f, Gs[0], Hs[0], 0, wr, wc, l15, lq);
#pragma unroll 1
    for (int c = 0; c < nch - 2; c += 2) {
        tile_chunk<true, true, true, -1, GT>(acc, f, R, cp, Gs[0], Hs[0], Gs[1], Hs[1], gsrc(c + 2), hsrc(c + 2), io, nullptr, 0, tid, gcp(c + 1));
        tile_chunk<true, true, true, -1, GT>(acc, f, R, cp, Gs[1], Hs[1], Gs[0], Hs[0], gsrc(c + 3), hsrc(c + 3), io, nullptr, 0, tid, gcp(c + 2));
    }
    tile_chunk<true, false, true, -1, GT>(acc, f, R, cp, Gs[0], Hs[0], Gs[1], Hs[1], gA, hP, io, nullptr, 0, tid, gcp(nch - 1));
    tile_chunk<false, false, false, -1, GT>(acc, f, R, cp, Gs[1], Hs[1], Gs[0], Hs[0], gA, hP, io, nullptr, 0, tid);
}

// Write-back of the new column block: A[i,k] = G_i P = -H_i  (A[k,i] = (G_i P)^T for i < k).  Not done by
// the panel kernel because the two panel workgroups of a row block both read the OLD A[i,k] as their G
// operand.  A work item of its own in the sweep kernel (wb): every thread first loads all of its 64 values, then
// stores them.
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
        __builtin_amdgcn_sched_barrier(0);  // (all 64 loads in flight before the first store: left to itself hipcc interleaves them)
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            const int idx = tid + 256 * u;
            dst[(size_t)(idx & 127) + (size_t)(idx >> 7) * ld] = -v[u];
        }
    } else {
        double *dst = A + (size_t)kblk * T + (size_t)i * T * ld;  // dst(c, r) = -H(r, c)
        // 16 columns of H at a time through LDS: Ts[c][r], then rows of dst are read across c
        // (the next 16 columns are requested before this round's barrier and stores: as one load / one LDS store per loop trip the
        // function form of this item waited out 64 round trips one after the other: 44 us an item instead of 26)
        double nx[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = tid + 256 * u;  // r = idx & 127, c = idx >> 7 (0..15)
            nx[u] = H[(size_t)(idx & 127) + (size_t)(idx >> 7) * pld];
        }
        for (int cb = 0; cb < T; cb += KC) {
            double cur[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) cur[u] = nx[u];
            if (cb + KC < T) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = tid + 256 * u;
                    nx[u] = H[(size_t)(idx & 127) + (size_t)(cb + KC + (idx >> 7)) * pld];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = tid + 256 * u;
                Ts[idx >> 7][idx & 127] = -cur[u];
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

// =====================================================================================================================
// The sweep as ONE persistent launch
// =====================================================================================================================
// Pivots are taken in GROUPS of sz <= 4 consecutive 128-blocks (m = 128 sz columns), each swept as one pivot of width m:
//       Pg = (A_gg)^-1 (m x m),   G = A_{.,g},   H = -G Pg,   A_ij += H_i G_j^T (K = m),   A_{.,g} <- -H,   A_gg <- -Pg,
// so the C tile of a trailing-update product is read and written once per sz rank-128 updates (with one pivot per pass
// the update is HBM-bound) and a tile's per-item overheads are paid once per sz times the work.
//
// The whole inverse is a list of WORK ITEMS, cut so that every item is one workgroup's job of tens of microseconds, and
// ordered so that every item depends only on items EARLIER in the list:
//   MAIN list, for group p = 0, 1, ...
//     panel(p)     (row block i, 64 of the m columns):  G_i, H_i = -G_i Pg            needs Pg(p), the group's columns of row i
//     diag2(p+2)   tiles of update p inside the diagonal super-block of group p+2      (early: the chain of that group needs them)
//     rest(p+1)    the tiles of update p in the next group's columns (rows outside both groups)
//     wb(p)        write the group's new columns back (A[., g] <- -H)
//     rem(p)       every other tile of update p
//   M list (the serial chain), for group q = 0, 1, ...
//     M(q)         Pg(q): gather the diagonal super-block into a dense scratch matrix, sweep it block by block (128 x 128
//                  pivot, then one level of row-slab jobs for the other blocks of the scratch matrix), scatter -Pg back
//     mpanel(q)    the panel items of the NEXT group's row blocks
//     diag(q+1)    the tiles of update q inside the next group's diagonal super-block  -> M(q+1) can start
//     (a schedule of single blocks has, instead of the last two: slab(q), xslab(q), slab2(q) -- the next pivot row's panel and
//      diagonal tile, the tile below it and the row after that as fused 16-row slabs, see sweep_slab_item)
//   The chain never waits for the bulk of an update, only for the early diag2 / rest items of the previous group.
// ONE launch of 2 workgroups per compute unit runs the list: a workgroup takes the next item off a device-wide counter,
// waits (one lane polling, s_sleep) until the few flags the item depends on are set, does it, publishes its result
// (agent-scope release) and sets the item's own flag.  Items are handed out in list order, so an item can only wait for
// items that somebody already holds: no deadlock.  The serial part of a group -- M(p+1), some dozen small items on a
// handful of workgroups -- runs while the ~3000 remainder tiles of update p keep every other workgroup busy; the tail of
// update p overlaps the head of update p+1; nothing is a kernel boundary, a stream, an event or a priority.
//
// Flags (device memory, zeroed by the host before the launch):
//   gen[I][J]   how many groups have been applied to tile (I, J), I >= J: every group touches every tile exactly once
//               (update, write-back or scatter), so "group p may touch it" is gen == p, and it leaves gen = p + 1
//   rb[p][i]    panel items of row block i done in group p (2 sz = complete)
//   mc[q]       M items of group q done (levels are thresholds of this one counter)
//   done[p]     tile and write-back items of group p done (guards the reuse of the double-buffered panels / Pg)
struct SweepDesc {
    double *A;
    size_t ld;
    int nblk, g, ng;
    double *G0, *H0;       // panels: parity q, column w of the group at G0 + (4 q + w) * pstride
    size_t pstride;
    double *Sg0, *Sg1;     // dense scratch copies of a group's diagonal super-block (ping-pong), ld = 128 sz
    double *Pg0, *Pg1;     // Pg by group parity, ld = 128 sz
    double *Pw;            // 128 x 128: inverse of the current pivot block
    unsigned *gen, *rb, *mc, *done, *next, *next_m, *mcu;
    unsigned *mxcc;        // XCC the chain's compute units are elected on (0 = not chosen yet, else id + 1): the first workgroup to arrive decides
    unsigned *arrived;     // workgroups of the launch that have shown up so far (the first member's counter serves a merged launch)
    unsigned *abort;       // set by a workgroup whose dependency wait ran out of time: everybody leaves (watchdog)
    unsigned long long timeout_ticks;  // bound of a single dependency wait, 100 MHz ticks
    unsigned long long hole_timeout_ticks;  // ... of what is left of a wait once its wave has been off the hardware (spin_until)
    int debug;             // GDCA_SWEEP_DEBUG bits (tests): 1 = workgroups on XCC 0 stay out of the election; 2 = nobody is elected
    const int *gs;         // [ng + 1]: first block of group p (groups need not have the same size: a short ramp 1, 2, .. opens
                           // the sweep so that the first chains are short while there is little update work to hide them behind)
    const int *item0;      // [ng + 1]: first item of group p's sequence in the main list
    const int *mitem0;     // [ng + 1]: first item of M(q) in the M list
    int total, total_m;
    int pro;               // main list: the first `pro` items are panel(0); group p's sequence follows at pro + item0[p]
    int rem_tail;          // remainder tiles of update p that are listed AFTER panel(p+1)
    int ppb;               // main-list panel items per pivot block and row: 2 (128 x 64 each) or 1 (128 x 128)
    int n_mcu;             // compute units to elect for the M list (<= 16; <= 32 with mcu_solo)
    int mcu_solo;          // one chain worker per elected compute unit: its second workgroup leaves the launch at once (GDCA_MCU_SOLO)
    int ring;              // Pg / panel buffers per kind: 2 (group parity) or 8 (single-block groups), see ring_panel
    int slab;              // single-block groups: the next pivot row's panel and diagonal tile as SLAB_ITEMS fused row-slab items (0: off)
    unsigned *sl;          // [3 ng] slab items of group p done (row b0 + 1), its xslab items, its slab items of row b0 + 2
    int n_real;
    int rl;                // real (non-padding) rows of the last block, rounded up to 16: 16 .. 128
    gdca_dev_scalars *sc;
    unsigned long long *dbg;  // optional (GDCA_SWEEP_TRACE): 100 MHz wall-clock stamps, (start, end) per M-list item
    unsigned long long *dbg_main;  // optional: [0] ticks tile items waited, [1] panel items waited, [2..] ticks / counts by kind
    unsigned long long *dbg_items; // optional: per main-list item (kind, p, a, b, workgroup), taken, end of its wait, done (100 MHz stamps)
};

// LDS of the sweep kernels, at file scope so that the out-of-line item functions (below) address it as LDS without having it
// handed to them as generic pointers: the staging buffers of the tile paths -- ONE array: the pivot's images (pivot_chain) span
// both staging buffers off one base -- and the few words a workgroup's threads share about its current and next item
__shared__ __attribute__((aligned(16))) double sw_lds[4][KC][LDS_LD];
// (ONE object: its members are immediate offsets off one address.  As separate variables each had an address register of its own,
// hoisted out of the persistent loop and parked in scratch across the items)
struct SweepShared {
    unsigned long long kbase;     // address of the kernel-argument segment (launch_desc)
    unsigned long long probe[2];  // the workgroup's start: shader clock, wall clock (the launch's clock measurement)
    unsigned long long stamp[2];  // trace: start of the current main-list item (wall clock, shader clock)
    int item, next, ready, fam, live, fnext, pnext;
    int p[8];                     // merged launch: per family, the group this workgroup's last item belonged to
    int cur;                      // trace: the main-list item this workgroup is working on (-1: an item of the chain)
    unsigned hw_home;             // HW_ID of thread 0's wave when it last looked (spin_until: a wave that has moved was saved and restored)
};
__shared__ SweepShared sw;
#define sw_item sw.item
#define sw_next sw.next
#define sw_ready sw.ready
#define sw_fam sw.fam
#define sw_live sw.live
#define sw_fnext sw.fnext
#define sw_pnext sw.pnext
#define sw_probe sw.probe
#define sw_stamp sw.stamp
#define sw_kbase sw.kbase
#define sw_p sw.p

// The descriptor of family f of the running launch, read where it lies: in the kernel-argument segment (constant address space:
// scalar loads).  k_sweep's argument is one SweepDesc (f = 0), k_sweep_merged's a SweepBatch, whose first member is fam[].
// launch_desc_k: inside the kernels themselves.  launch_desc: anywhere -- the out-of-line item functions take the family's index,
// not a reference to its descriptor, and `llvm.amdgcn.kernarg.segment.ptr` is NULL outside a kernel: the kernels leave the
// segment's address in LDS (sw_kbase, thread 0, before their first barrier).
typedef const SweepDesc __attribute__((address_space(4))) *kernarg_desc_t;

__device__ __forceinline__ const SweepDesc &launch_desc_k(int f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return ((const SweepDesc *)(kernarg_desc_t)__builtin_amdgcn_kernarg_segment_ptr())[f];
#else
    (void)f;
    static const SweepDesc host_pass_dummy{};  // (host pass: never executed)
    return host_pass_dummy;
#endif
}

__device__ __forceinline__ void launch_desc_init()  // thread 0 of a kernel, before the workgroup's first barrier
{
#if defined(__HIP_DEVICE_COMPILE__)
    sw_kbase = (unsigned long long)(const SweepDesc *)(kernarg_desc_t)__builtin_amdgcn_kernarg_segment_ptr();
#endif
}

__device__ __forceinline__ const SweepDesc &launch_desc(int f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long b = sw_kbase;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    // (the generic address of the segment, back in the constant address space: the loads stay scalar)
    return ((const SweepDesc *)(kernarg_desc_t)(((unsigned long long)hi << 32) | lo))[f];
#else
    return launch_desc_k(f);
#endif
}

__device__ __forceinline__ int g_start(const SweepDesc &D, int p)
{
    return D.gs[p];
}
__device__ __forceinline__ int g_size(const SweepDesc &D, int p)
{
    return D.gs[p + 1] - D.gs[p];
}
// k-chunks (16 deep) of a tile item of update p: 8 per pivot block; the block at the ragged end of the matrix counts only the
// chunks that hold real columns (rl of its 128; the padding columns of G and H are zero), rounded up to an even number
__device__ __forceinline__ int g_chunks(const SweepDesc &D, int p)
{
    const int sz = g_size(D, p);
    if (D.rl < T && D.gs[p + 1] == D.nblk && sz > 1) return (T / KC) * (sz - 1) + 2 * ((D.rl + 2 * KC - 1) / (2 * KC));
    return (T / KC) * sz;
}
// tile and write-back items of group p (what done[p] counts up to)
__device__ __forceinline__ unsigned g_done_total(const SweepDesc &D, int p)
{
    const int sz = g_size(D, p), pn = D.nblk - sz;
    return (unsigned)(pn * sz + (long long)pn * (pn + 1) / 2);
}

// Pg and the G / H panels of a group live in a ring of D.ring buffers indexed by the group number: 2 (parity; multi-block groups fill
// the four panels of a slot) or, for single-block groups, 8 one-panel slots in the same memory -- the chain may then run that many
// groups ahead of the update instead of two (on a chain-bound matrix the wait for "group p - 2 complete" was what paced it)
__device__ __forceinline__ size_t ring_panel(const SweepDesc &D, int p)
{
    return D.ring == 2 ? (size_t)4 * (p & 1) : (size_t)(p % D.ring);
}
__device__ __forceinline__ double *ring_pg(const SweepDesc &D, int p)
{
    return D.ring == 2 ? ((p & 1) ? D.Pg1 : D.Pg0) : D.Pg0 + (size_t)(p % D.ring) * T * T;
}

__device__ __forceinline__ unsigned flag_load(const unsigned *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// every wave's stores have left the CU, then ONE agent-scope release; the caller then sets its flag(s) from thread 0
__device__ __forceinline__ void publish_begin()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// The write-through form: every store of the published bytes was an agent-scope (sc1) store -- it goes to memory, not
// into this XCD's write-back L2 -- so draining the stores (each wave's vmcnt(0)) is all there is to do before the flag: no
// L2 write-back of the whole XCD per item (measured cost of that release beside streaming tiles: several microseconds of
// a ~100 us tile item)
__device__ __forceinline__ void store_wt(double *p, double v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void publish_wt_begin()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// ---- waiting for flags, with a way out ---------------------------------------------------------------------------------------
// Every dependency wait of the kernel is thread 0 polling a few flags (s_sleep between polls).  A wait that never ends -- a
// bug in the item order, a workgroup of the launch that was never scheduled, a device fault elsewhere -- would hang the whole
// GPU behind a C-ABI that promises to return a status.  So a wait is bounded (D.timeout_ticks of the 100 MHz wall clock, seconds:
// a healthy wait is microseconds): the workgroup that runs out of time sets the device-wide abort word and sc->info = INT_MIN,
// every poll loop also looks at that word, and everybody leaves; the host runs the inverse once more (gdca_run_collect) and maps
// a second failure to GDCA_EHIP.  The fast path (flags already set) costs nothing; a poll iteration costs one more load.
//
// The one way a healthy launch gets there, found in round 6 (gdca_api.hip, warm_stream): the driver takes the queues of the device
// off the hardware and back whenever it maps a new one -- a stream's first command in this process, another process starting --
// and saves and restores the waves of every running kernel for that.  Measured from in here: all waves of the launch miss ~1.7 ms
// and wake up on other compute units.  Alone on the device the launch then carries on.  Beside kernels of other queues it may
// get only part of its workgroups back (416 of 512 in every case seen: the second workgroup of three compute units per shader
// engine stays out, frozen in the middle of whatever item it was working on), the restored ones poll for the items the others hold
// and keep the compute units those need: a standstill to the end of the bound, although nothing is wrong with the launch.  A
// polling wave SEES the event -- two polls are microseconds apart, a hole of SWEEP_HOLE_TICKS between them means the wave was off
// the hardware; a wave that finds itself on another compute unit than at its last wait was off it inside an item (2 of 11 disturbed
// launches of tools/rounds/r06/gpu_r6s.sh had no workgroup polling at that moment and ran into the full bound) -- and gives its wait only
// D.hole_timeout_ticks more: the launch ends after tens of milliseconds instead of seconds,
// and the second attempt has the device to itself like any other launch.
#define SWEEP_HOLE_TICKS 30000ull  // 0.3 ms of the 100 MHz clock
__device__ __forceinline__ int *abort_lds()
{
    __shared__ int w;  // this workgroup has seen the abort (written by thread 0 before a barrier, read by all after it)
    return &w;
}

// `ready` loads the flags of the wait and says whether they are all set; the abort word travels in the same round trip.
// Returns false when the wait was abandoned.  (Thread 0 only.)
// where thread 0's wave runs: compute unit, SIMD, wave slot (a wave never moves -- unless it is saved and restored)
__device__ __forceinline__ unsigned sweep_hw_id()
{
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    return hw & 0x000fff3fu;  // wave slot, SIMD, compute unit, shader array, shader engine, workgroup slot (not the queue / state fields)
}

template <class F>
__device__ __forceinline__ bool spin_until(const SweepDesc &D, F ready)
{
    unsigned long long t0 = 0ull, prev = 0ull, limit = D.timeout_ticks;
    for (unsigned it = 0;; ++it) {
        const unsigned ab = flag_load(D.abort);
        const bool r = ready();
        if (ab != 0u) return false;
        if (r) return true;
        const unsigned long long now = wall_clock64();
        if (it == 0) {
            t0 = now;
            // ... or it sits on another compute unit than the last time it looked: the hole was in an item, not in a wait
            const unsigned hw = sweep_hw_id();
            if (hw != sw.hw_home) {
                sw.hw_home = hw;
                if (D.hole_timeout_ticks < limit) limit = D.hole_timeout_ticks;
            }
        } else {
            // a hole between two polls: this wave was off the hardware (see above) -- from here on the wait gets the short bound
            if (now - prev > SWEEP_HOLE_TICKS && now - t0 + D.hole_timeout_ticks < limit) limit = now - t0 + D.hole_timeout_ticks;
            if (now - t0 > limit) {
                __hip_atomic_store(D.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&D.sc->info, (int)0x80000000, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        prev = now;
        __builtin_amdgcn_s_sleep(8);
    }
}

// after thread 0 has seen all the flags it polled (ok = none of its waits was abandoned): agent-scope acquire, then the
// workgroup may load the data.  Returns false when a wait was abandoned (abort): the caller drops the item.
__device__ __forceinline__ bool acquire_end(bool ok)
{
    if (threadIdx.x == 0) {
        *abort_lds() = ok ? 0 : 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    return *abort_lds() == 0;
}

// M(q): sz (sz+1) / 2 gather items (one lower tile each), per block w a pivot + 8 (sz-1) row-slab jobs, then
// sz (sz+1) / 2 scatter items
// (a group of ONE block needs no scratch copy: a single item, the pivot in place)
__device__ __forceinline__ int m_items(int sz)
{
    return sz == 1 ? 1 : sz * (sz + 1) + sz * (1 + SLAB_ITEMS * (sz - 1));
}

// ---- the main list: item number -> what to do --------------------------------------------------------------------------
// The first D.pro items are panel(0); then, group after group,
//     diag2(p+2) | rest(p+1) | wb(p) | rem(p) but for its tail | panel(p+1) | the tail of rem(p)
// (panel(p+1) covers the rows outside groups p+1 and p+2: Pg(p+1) comes from the chain, which runs a group ahead, and its
// other inputs were produced early in this sequence -- so the panels are complete when the tile items of update p+1 are
// handed out, instead of holding all of them up at the start of every group).
// kind 0: panel item (group p, row block a, part b); 1: tile item (group p, tile (a, b)); 4: the same, one of the early ones the
// chain of a later group waits for (diag2, rest); 2: write-back item (group p, index a); 3: nothing (a remainder slot that
// belongs to diag2).  `p` is the caller's running group (items ascend).
struct MainItem {
    int kind, p, a, b;
};
__device__ __forceinline__ MainItem main_decode(const SweepDesc &D, int &p, int item)
{
    if (item < D.pro) {  // panel(0): nobody is ahead of it
        const int sz0 = g_size(D, 0), nsz0 = D.ng > 1 ? g_size(D, 1) : 0;
        const int i0 = sz0 + nsz0 + item / (D.ppb * sz0);
        if (D.slab && i0 == 2) return MainItem{3, 0, 0, 0};  // row block b0 + 2: the chain's (sweep_slab_item, rowoff 2)
        return MainItem{0, 0, i0, item % (D.ppb * sz0)};
    }
    while (item - D.pro >= D.item0[p + 1]) ++p;
    int e = item - D.pro - D.item0[p];
    const int b0 = g_start(D, p), sz = g_size(D, p), c0 = b0 + sz;
    const int nsz = p + 1 < D.ng ? g_size(D, p + 1) : 0, d0 = c0 + nsz;
    const int n2 = p + 2 < D.ng ? g_size(D, p + 2) : 0;
    const int nrest = D.nblk - sz - nsz;  // blocks outside this group and the next
    // (between single blocks the chain itself does tiles (b0+1, b0+1), (b0+2, b0+1), (b0+2, b0+2) and the panels of row blocks
    // b0 + 1 and b0 + 2 -- sweep_slab_item, sweep_xslab_item -- and what it waits for a step later is row block b0 + 3: its tiles
    // (b0+3, b0+2), (b0+3, b0+3) take the early slots here, (b0+3, b0+1) is among rest(p+1))
    // ... and the tiles (i, b0+2), i > b0+3, follow them: column b0 + 2 is the next step's column b0 + 1, whose tiles must carry
    // this update before they can take the next one early -- as ordinary remainder tiles they were what the chain ended up waiting for
    const int n_diag2 = D.slab ? (b0 + 3 < D.nblk ? 2 : 0) + max(0, D.nblk - b0 - 4) : n2 * (n2 + 1) / 2;
    if (e < n_diag2) {
        if (D.slab) return e < 2 ? MainItem{4, p, b0 + 3, b0 + 2 + e} : MainItem{4, p, b0 + 2 + e, b0 + 2};
        int mm = 0, first = 0;
        while (e >= first + (n2 - mm)) {
            first += n2 - mm;
            ++mm;
        }
        return MainItem{4, p, d0 + mm + (e - first), d0 + mm};
    }
    e -= n_diag2;
    if (e < nsz * nrest) {
        const int mm = e / nrest, local = e % nrest;
        const int b = local < b0 ? local : local - b0 + d0;
        const int cb = c0 + mm;
        if (D.slab && b == d0) return MainItem{3, p, 0, 0};  // tile (c0 + 1, c0): done on the chain (sweep_xslab_item)
        return MainItem{4, p, b > cb ? b : cb, b > cb ? cb : b};
    }
    e -= nsz * nrest;
    const int n_wb = (D.nblk - sz) * sz;
    if (e < n_wb) return MainItem{2, p, e, 0};
    e -= n_wb;
    const int n_rem = nrest > 0 ? nrest * (nrest + 1) / 2 : 0;
    const int n_head = n_rem - min(n_rem, D.rem_tail);
    if (e >= n_head) {
        const int n_pan = nsz > 0 ? (D.nblk - nsz - n2) * D.ppb * nsz : 0;
        if (e - n_head < n_pan) {
            const int ep = e - n_head;
            int i = ep / (D.ppb * nsz);
            if (i >= c0) i += nsz + n2;
            if (D.slab && i == c0 + 2) return MainItem{3, p, 0, 0};  // row block b0' + 2 of group p + 1 (b0' = c0): the chain's
            return MainItem{0, p + 1, i, ep % (D.ppb * nsz)};
        }
        e -= n_pan;
    }
    int ii = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
    while ((long long)ii * (ii + 1) / 2 > e) --ii;
    while ((long long)(ii + 1) * (ii + 2) / 2 <= e) ++ii;
    int jj = e - (int)((long long)ii * (ii + 1) / 2);
    if (ii >= b0) ii += sz + nsz;
    if (jj >= b0) jj += sz + nsz;
    if (D.slab) {
        // (b0+2, b0+2): the chain's; (b0+3, b0+3) and column b0 + 2 below the diagonal: early slots
        if ((jj >= d0 && ii <= d0 + 1) || jj == d0) return MainItem{3, p, 0, 0};
    } else if (jj >= d0 && ii < d0 + n2)
        return MainItem{3, p, 0, 0};  // inside the diagonal super-block of group p+2: done as diag2
    return MainItem{1, p, ii, jj};
}

// One 128 x 128 pivot by the calling 256-thread workgroup (LDS of the tile paths reused: Gs and Hs are the two halves of ONE array,
// see k_sweep).  ONE call site in the kernel: the unrolled 16-step micro-sweep is long.
template <class Mid>
__device__ __forceinline__ void sweep_pivot(const double *Ain, size_t ldin, double *Aout, size_t ldout, double *P,
                                                      double (*Gs)[KC][LDS_LD], double (*Hs)[KC][LDS_LD], int index0, int n_real,
                                                      gdca_dev_scalars *sc, Mid mid, unsigned long long *ph = nullptr)
{
    double *buf = &Gs[0][0][0];
    (void)Hs;
    int *badj = reinterpret_cast<int *>(buf + PVC_FLAG_OFF);
    if (threadIdx.x == 0) *badj = 0;
    __syncthreads();
    pivot_chain(Ain, ldin, Aout, ldout, P, (size_t)T, buf, badj, mid, ph);
    __syncthreads();
    if (threadIdx.x == 0 && *badj != 0) {
        // pivots run one after the other (each waits for the previous one's items): the first report is the smallest index
        const int idx = index0 + *badj;
        if (idx <= n_real && sc->info == 0) sc->info = idx;
    }
}

// ---- row-slab products (the chain's items between single blocks and inside a multi-block group) --------------------------------

// x0, x1 (two 16 x 16 blocks of one 16-row slab) += sum_k b(r, k) a(c, k): operand elements straight from memory in MFMA operand
// layout -- b(l15, 4 k4 + lq) = bp[(4 k4) ldb], a(l15, 4 k4 + lq) = ap[(4 k4) lda] and ap[(4 k4) lda + 16] with the lane's (l15, lq)
// part already in bp / ap -- SLAB_PD rounds of 16 k ahead of the MFMAs.  B_LDS: b comes from LDS and is read where it is used.
template <bool B_LDS>
struct SlabPipe {
    double b[SLAB_PD][4], a0[SLAB_PD][4], a1[SLAB_PD][4];
    const double *bp, *ap;
    size_t ldb, lda;
    __device__ __forceinline__ void load(int kb, int st)
    {
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const size_t k = (size_t)(16 * kb + 4 * t4);
            if (!B_LDS) b[st][t4] = bp[k * ldb];
            a0[st][t4] = ap[k * lda];
            a1[st][t4] = ap[k * lda + 16];
        }
    }
    __device__ __forceinline__ void prologue()
    {
#pragma unroll
        for (int st = 0; st < SLAB_PD - 1; ++st) load(st, st);
    }
    // COPY: the b elements also go to cp[(4 k4) ldc] (write-through), the 16 k of round kb by wave kb & 3
    template <bool COPY>
    __device__ __forceinline__ void run(double4_t &x0, double4_t &x1, double *cp, size_t ldc, int wv)
    {
#pragma unroll
        for (int kb = 0; kb < NMB; ++kb) {
            if (kb + SLAB_PD - 1 < NMB) load(kb + SLAB_PD - 1, (kb + SLAB_PD - 1) % SLAB_PD);
            const int st = kb % SLAB_PD;
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                const double bv = B_LDS ? bp[(size_t)(16 * kb + 4 * t4) * ldb] : b[st][t4];
                x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[st][t4], bv, x0, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[st][t4], bv, x1, 0, 0, 0);
            }
            if (COPY && (kb & 3) == wv) {
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) store_wt(&cp[(size_t)(16 * kb + 4 * t4) * ldc], b[st][t4]);
            }
        }
    }
};

// ---- M(q): one item of the super-block inverse of group q -------------------------------------------------------------
__device__ __forceinline__ void sweep_m_item(const SweepDesc &D, int q, int e, double (*Gs)[KC][LDS_LD], double (*Hs)[KC][LDS_LD],
                                             unsigned long long *dbg_ready = nullptr)
{
    const int tid = opaque_tid();
    const int b0 = g_start(D, q), sz = g_size(D, q), m = sz * T;
    const int per_w = 1 + SLAB_ITEMS * (sz - 1), nt = sz * (sz + 1) / 2;
    const int nm = 2 * nt + sz * per_w;
    double *Agg = D.A + (size_t)b0 * T + (size_t)b0 * T * D.ld;
    unsigned *mc = D.mc + q;
    // The FIRST pivot of a multi-block group is item 0 and reads its tile straight from A: it needs only the group's first
    // diagonal tile (the first thing the previous group's chain produces), not the gathered copy of the whole super-block, so
    // it runs beside the other diagonal tiles and the gathers instead of after them.  Its old slot (e == nt) only counts.
    const bool first_pivot = sz > 1 && e == 0;
    if (sz > 1 && e == nt) {
        if (tid == 0) __hip_atomic_fetch_add(mc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (sz > 1 && !first_pivot && (e < nt || e >= nm - nt)) {
        // gather / scatter, one lower-triangle tile (ib >= jb) of the super-block per item
        const bool gather = e < nt;
        int x = gather ? e : e - (nm - nt), jb = 0;
        while (x >= sz - jb) {
            x -= sz - jb;
            ++jb;
        }
        const int ib = jb + x;
        unsigned *genp = D.gen + (size_t)(b0 + ib) * D.nblk + (b0 + jb);
        bool ok = true;
        if (tid == 0) {
            if (gather) {
                ok = spin_until(D, [&] { return flag_load(genp) >= (unsigned)q; });   // the tile carries all earlier groups
            } else {
                ok = spin_until(D, [&] { return flag_load(mc) >= (unsigned)(nm - nt); });
                if (ok && q >= D.ring) {
                    const unsigned want = g_done_total(D, q - D.ring);  // the group that used this Pg slot before is complete
                    ok = spin_until(D, [&] { return flag_load(D.done + (q - D.ring)) >= want; });
                }
            }
        }
        if (!acquire_end(ok)) return;
        double *At = Agg + (size_t)ib * T + (size_t)jb * T * D.ld;
        double(*Ts)[LDS_LD] = Gs[0];
        // the tile goes 16 columns at a time: straight copies stay contiguous along columns, and the mirror images
        // (transposes) pass through LDS so that their stores are 128-byte row segments instead of single 8-byte words
        if (gather) {
            // Sg0 (m x m, full storage) <- the tile and its mirror (lower triangle authoritative inside a diagonal tile)
            double *S = D.Sg0;
            double *Sd = S + (size_t)ib * T + (size_t)jb * T * m;   // (r, c)
            double *Sm = S + (size_t)jb * T + (size_t)ib * T * m;   // mirror: (c, r)
            for (int cb = 0; cb < T; cb += KC) {
                __syncthreads();
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = tid + 256 * u, r = idx & 127, c = cb + (idx >> 7);
                    const double v = At[(size_t)r + (size_t)c * D.ld];
                    if (ib != jb || r >= c) Sd[(size_t)r + (size_t)c * m] = v;
                    Ts[idx >> 7][r] = v;
                }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = tid + 256 * u, cc = idx & 15, r = idx >> 4;   // element (r, cb + cc) -> mirror position
                    if (ib != jb || r > cb + cc) Sm[(size_t)(cb + cc) + (size_t)r * m] = Ts[cc][r];
                }
            }
        } else {
            // A_gg <- -Pg (lower-triangle tiles, diagonal tiles in full), Pg <- exactly symmetric from the lower triangle
            const double *Sf = ((sz & 1) ? D.Sg1 : D.Sg0) + (size_t)ib * T + (size_t)jb * T * m;
            double *Pg = ring_pg(D, q);
            double *Pd = Pg + (size_t)ib * T + (size_t)jb * T * m, *Pm = Pg + (size_t)jb * T + (size_t)ib * T * m;
            for (int cb = 0; cb < T; cb += KC) {
                __syncthreads();
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = tid + 256 * u, r = idx & 127, c = cb + (idx >> 7);
                    const double v = Sf[(size_t)r + (size_t)c * m];
                    if (ib != jb || r >= c) {
                        At[(size_t)r + (size_t)c * D.ld] = v;
                        Pd[(size_t)r + (size_t)c * m] = -v;
                    }
                    Ts[idx >> 7][r] = v;
                }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = tid + 256 * u, cc = idx & 15, r = idx >> 4;
                    if (ib != jb || r > cb + cc) {
                        Pm[(size_t)(cb + cc) + (size_t)r * m] = -Ts[cc][r];
                        if (ib == jb) At[(size_t)(cb + cc) + (size_t)r * D.ld] = Ts[cc][r];
                    }
                }
            }
        }
        publish_begin();
        if (tid == 0) {
            if (!gather) __hip_atomic_store(genp, (unsigned)(q + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(mc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    const int w = (sz == 1 || first_pivot) ? 0 : (e - nt) / per_w, r = (sz == 1 || first_pivot) ? 0 : (e - nt) % per_w;
    const int base_w = nt + w * per_w;  // M items before the pivot of block w
    const double *Sin = (w & 1) ? D.Sg1 : D.Sg0;
    double *Sout = (w & 1) ? D.Sg0 : D.Sg1;
    if (r == 0) {
        // pivot of block w: on the scratch matrix, or -- a group of ONE block -- in place on A with Pg = its inverse
        unsigned *genp = D.gen + (size_t)b0 * D.nblk + b0;
        bool ok = true;
        if (tid == 0) {
            if (sz == 1) {
                ok = spin_until(D, [&] { return flag_load(genp) >= (unsigned)q; });
                if (ok && q >= D.ring) {
                    const unsigned want = g_done_total(D, q - D.ring);  // the group that used this Pg slot before is complete
                    ok = spin_until(D, [&] { return flag_load(D.done + (q - D.ring)) >= want; });
                }
            } else if (first_pivot) {
                // (the tile at generation q implies that M(q-1) is complete: the scratch matrices and Pw are free)
                ok = spin_until(D, [&] { return flag_load(genp) >= (unsigned)q; });
            } else {
                ok = spin_until(D, [&] { return flag_load(mc) >= (unsigned)base_w; });
            }
        }
        if (!acquire_end(ok)) return;
        if (dbg_ready && tid == 0) *dbg_ready = wall_clock64();  // trace: a pivot's start stamp becomes the end of its wait
        const size_t dd = (size_t)w * T + (size_t)w * T * m;
        const double *pin = (sz == 1 || first_pivot) ? (const double *)Agg : Sin + dd;
        double *pout = sz == 1 ? Agg : Sout + dd;
        const size_t pld_in = (sz == 1 || first_pivot) ? D.ld : (size_t)m, pld_out = sz == 1 ? D.ld : (size_t)m;
        // trace: where a pivot item's time goes (thread 0's stamps: entry, loads and first micro-pivot done, sweep done)
        __shared__ unsigned long long ph[3];
        // a single block's Pg is published as soon as ITS stores have drained (the slab / panel items of the next pivot row wait for
        // mc only), the block's own tile -Pg behind it; inside a multi-block group the scratch copy is read under mc too: one publication
        auto mid = [&] {
            if (sz == 1) {
                publish_wt_begin();
                if (tid == 0) __hip_atomic_fetch_add(mc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        };
        sweep_pivot(pin, pld_in, pout, pld_out, sz == 1 ? ring_pg(D, q) : D.Pw, Gs, Hs, (b0 + w) * T, D.n_real, D.sc, mid, D.dbg ? ph : nullptr);
        publish_wt_begin();  // the pivot's outputs are write-through stores
        if (tid == 0) {
            if (sz == 1)
                __hip_atomic_store(genp, (unsigned)(q + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                __hip_atomic_fetch_add(mc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (D.dbg) {
                atomicAdd(D.dbg_main + 8 + 600, ph[1] - ph[0]);
                atomicAdd(D.dbg_main + 8 + 601, ph[2] - ph[1]);
                atomicAdd(D.dbg_main + 8 + 602, wall_clock64() - ph[2]);
                atomicAdd(D.dbg_main + 8 + 603, 1ull);
            }
        }
        return;
    }
    // Jobs on the scratch matrix, one per OTHER row block i and 16-row slab (8 (sz-1) per block w, ONE level between two pivots):
    //      N = S_iw Pw  ->  S_iw (and its mirror S_wi);    S_ij <- S_ij - N S_jw^T for every j != w
    // Rows are independent in all of it, so a slab does the sz products of its 16 rows back to back, N changing hands through LDS
    // (SlabPipe: operands straight from L2 in MFMA operand layout).  As half-tile jobs -- 2 (sz-1) for S_iw, then 2 (sz-1)^2 for the
    // S_ij, two dependent levels of 128 x 64 x 128 products on one workgroup each -- a block took 30-45 us per level.
    const int k = r - 1, s = k % SLAB_ITEMS;
    int ii = k / SLAB_ITEMS;
    if (ii >= w) ++ii;
    bool ok = true;
    if (tid == 0) ok = spin_until(D, [&] { return flag_load(mc) >= (unsigned)(base_w + 1); });
    if (!acquire_end(ok)) return;
    const int lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int R0 = MB * s;
    constexpr int NL = MB + 1;  // N in LDS: Nl[c][r], 17 doubles per column (the mirror is read along c)
    double *Nl = &Gs[0][0][0];
    const size_t ms = (size_t)m;
    double4_t n0 = (double4_t){0.0, 0.0, 0.0, 0.0}, n1v = (double4_t){0.0, 0.0, 0.0, 0.0};
    {
        SlabPipe<false> pipe;
        pipe.bp = Sin + (size_t)(ii * T + R0 + l15) + (size_t)(w * T + lq) * ms;
        pipe.ldb = ms;
        pipe.ap = D.Pw + (size_t)(32 * wv + l15) + (size_t)lq * T;
        pipe.lda = T;
        pipe.prologue();
        pipe.run<false>(n0, n1v, nullptr, 0, wv);
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int c = 32 * wv + lq + 4 * reg;
        Nl[c * NL + l15] = n0[reg];
        Nl[(c + 16) * NL + l15] = n1v[reg];
        store_wt(&Sout[(size_t)(ii * T + R0 + l15) + (size_t)(w * T + c) * ms], n0[reg]);
        store_wt(&Sout[(size_t)(ii * T + R0 + l15) + (size_t)(w * T + c + 16) * ms], n1v[reg]);
    }
    __syncthreads();  // N(slab, :) of all four waves is in LDS
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int idx = tid + 256 * u, c = idx & 127, rr = idx >> 7;  // the mirror: row segments of S_wi
        store_wt(&Sout[(size_t)(w * T + c) + (size_t)(ii * T + R0 + rr) * ms], Nl[c * NL + rr]);
    }
#pragma unroll 1
    for (int jx = 0; jx < sz - 1; ++jx) {
        const int jj = jx >= w ? jx + 1 : jx;
        SlabPipe<true> pipe;
        pipe.bp = Nl + lq * NL + l15;
        pipe.ldb = NL;
        pipe.ap = Sin + (size_t)(jj * T + 32 * wv + l15) + (size_t)(w * T + lq) * ms;  // S_jw(c', k), c' = this wave's columns
        pipe.lda = ms;
        pipe.prologue();
        const double *cin = Sin + (size_t)(ii * T + R0 + l15) + (size_t)(jj * T + 32 * wv) * ms;
        double c0[4], c1[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            c0[reg] = cin[(size_t)(lq + 4 * reg) * ms];
            c1[reg] = cin[(size_t)(16 + lq + 4 * reg) * ms];
        }
        double4_t x0 = (double4_t){0.0, 0.0, 0.0, 0.0}, x1 = (double4_t){0.0, 0.0, 0.0, 0.0};
        pipe.run<false>(x0, x1, nullptr, 0, wv);
        double *out = Sout + (size_t)(ii * T + R0 + l15) + (size_t)(jj * T + 32 * wv) * ms;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            store_wt(&out[(size_t)(lq + 4 * reg) * ms], c0[reg] - x0[reg]);
            store_wt(&out[(size_t)(16 + lq + 4 * reg) * ms], c1[reg] - x1[reg]);
        }
    }
    publish_wt_begin();
    if (tid == 0) __hip_atomic_fetch_add(mc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- panel(p): G_i and 32 TM columns of H_i = -G_i Pg for one row block ---------------------------------------------------
// (i = the row block, y = which 32 TM of the m columns of H).  TM = 2: 64 columns, two items per pivot block -- the chain's
// form (M list), where the next group's rows are latency-critical; TM = 4: 128 columns, one item per pivot block -- the main
// list's form for multi-block groups, where the panels are throughput (a 128 x 64 product does half the MFMA work per staged
// operand byte: 73 us per item against 108 us for twice the flops).  The row's counter rb advances by TM / 2 per item.
template <int TM>
__device__ __forceinline__ void sweep_panel_item(const SweepDesc &D, int p, int i, int y, double (*Gs)[KC][LDS_LD],
                                                 double (*Hs)[KC][LDS_LD])
{
    const int tid = opaque_tid(), lane = tid & 63, wv = tid >> 6;
    const int wr = wv & 1, wc = wv >> 1, l15 = lane & 15, lq = lane >> 4;
    const int b0 = g_start(D, p), sz = g_size(D, p);
    const int w = TM == 2 ? y >> 1 : y, ch = TM == 2 ? (y & 1) : 0;
    bool ok = true;
    if (tid == 0) {
        const unsigned long long t0 = D.dbg ? wall_clock64() : 0ull;
        // Pg(p); the group's columns of row i at generation p; the panel buffers of parity p free (group p-2 complete): all
        // flags of a round are loaded together
        const unsigned want = p >= D.ring ? g_done_total(D, p - D.ring) : 0u;
        const unsigned *dn = D.done + (p >= D.ring ? p - D.ring : 0);
        const unsigned *gp[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int k = b0 + (v < sz ? v : 0), I = i > k ? i : k, J = i > k ? k : i;
            gp[v] = D.gen + (size_t)I * D.nblk + J;
        }
        const unsigned nmi = (unsigned)m_items(sz);
        ok = spin_until(D, [&] {
            const unsigned f0 = flag_load(D.mc + p), f1 = flag_load(gp[0]), f2 = flag_load(gp[1]), f3 = flag_load(gp[2]),
                           f4 = flag_load(gp[3]), f5 = flag_load(dn);
            return (f0 >= nmi) & (f1 >= (unsigned)p) & (f2 >= (unsigned)p) & (f3 >= (unsigned)p) & (f4 >= (unsigned)p) & (f5 >= want);
        });
        if (D.dbg) {
            if (sw.cur >= 0) D.dbg_items[4 * (size_t)sw.cur + 2] = wall_clock64();
            atomicAdd(D.dbg_main + 1, wall_clock64() - t0);
            if (D.slab && i == b0 + 3) atomicMax(D.dbg_main + 8 + 1024 + 3 * D.ng + 16 * p + 9, wall_clock64());  // trace: last half ready
        }
    }
    if (!acquire_end(ok)) return;
    const size_t ld = D.ld, pgld = (size_t)sz * T;
    const double *Pg = ring_pg(D, p);
    double *G0 = D.G0 + ring_panel(D, p) * D.pstride, *H0 = D.H0 + ring_panel(D, p) * D.pstride;
    double4_t acc[TM][4];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = (double4_t){0.0, 0.0, 0.0, 0.0};
    // H operand of pivot block v: Pg(c, k) for c = w 128 + ch 64 + .., k = v 128 + ..  (Pg is symmetric)
    const double *hsrc0 = Pg + (size_t)w * T + (size_t)ch * 64;
    double *gcopy0 = (y == 0) ? G0 + (size_t)i * T : nullptr;
    if constexpr (TM == 4) {
        // the hand-scheduled chunk loop of the tile items (tile_chunk), K = 128 sz
        if (i > b0)
            panel_chunks<false>(acc, D.A + (size_t)i * T + (size_t)b0 * T * ld, ld, hsrc0, pgld, sz, Gs, Hs, gcopy0, D.pstride);
        else
            panel_chunks<true>(acc, D.A + (size_t)b0 * T + (size_t)i * T * ld, ld, hsrc0, pgld, sz, Gs, Hs, gcopy0, D.pstride);
    } else if (i > b0) {  // below the group: G_i = A[i, k]
#pragma unroll 1
        for (int v = 0; v < sz; ++v)
            tile_product<false, TM>(acc, D.A + (size_t)i * T + (size_t)(b0 + v) * T * ld, ld, hsrc0 + (size_t)v * T * pgld, pgld, Gs[0],
                                   Hs[0], gcopy0 ? gcopy0 + (size_t)v * D.pstride : nullptr, ld);
    } else {       // above the group: G_i = A[k, i]^T
#pragma unroll 1
        for (int v = 0; v < sz; ++v)
            tile_product<true, TM>(acc, D.A + (size_t)(b0 + v) * T + (size_t)i * T * ld, ld, hsrc0 + (size_t)v * T * pgld, pgld, Gs[0],
                                  Hs[0], gcopy0 ? gcopy0 + (size_t)v * D.pstride : nullptr, ld);
    }
    double *Hw = H0 + (size_t)w * D.pstride + (size_t)i * T;
    if constexpr (TM == 4) {
        // (as in the tile item: four 32-bit offsets, computed after the chunk loop from a fresh thread index)
        const int t2 = opaque_tid(), lane2 = t2 & 63, wv2 = t2 >> 6;
        const TileOff to = tile_off(ld, wv2 & 1, wv2 >> 1, lane2 & 15, lane2 >> 4);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            char *base = const_cast<char *>(tile_strip(Hw, ld, tm));
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) (reinterpret_cast<double *>(base + to.o[reg]))[16 * tn] = -acc[tm][tn][reg];
        }
    } else {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int r = wr * 64 + tn * 16 + l15;
                    const int c = ch * 64 + wc * (16 * TM) + tm * 16 + lq + 4 * reg;
                    Hw[(size_t)r + (size_t)c * ld] = -acc[tm][tn][reg];
                }
    }
    publish_begin();
    if (tid == 0) __hip_atomic_fetch_add(D.rb + (size_t)p * D.nblk + i, (unsigned)(TM / 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- single-block groups: the chain between two pivots as row slabs ------------------------------------------------------------
// With one block per group the next pivot waits for  H = -G P  (panel of its row block c0 = b0 + 1, G = A[c0, b0], P = Pg) and then for
// its diagonal tile  C += G H^T = H G^T  (P is symmetric): two dependent 128 x 128 x 128 products, 14 us each on one compute unit at
// the full MFMA rate, which is what the panel and tile items took.  Rows are independent in both: slab s (16 rows) needs G's 16 rows
// and all of P for its rows of H, then those rows of H and all of G for its rows of C -- so SLAB_ITEMS workgroups do one slab each,
// side by side, with no hand-over between the two products but a barrier inside the workgroup.  Operands come straight from L2 in
// MFMA operand layout (8 bytes per lane: lane (l15, lq) holds element (16 blk + l15, 4 k4 + lq)); H changes hands through LDS.
// The last slab to finish sets the flags the panel items (rb += 2) and the tile item (gen, done) would have set.
// rowoff 1: the next pivot's row block; 2: the one after it (so that what the chain needs from the main list is a step further away)
__device__ __forceinline__ void sweep_slab_item(const SweepDesc &D, int p, int s, int rowoff, double *lds, unsigned long long *dbg_ready = nullptr)
{
    const int tid = opaque_tid(), lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b0 = g_start(D, p), c0 = b0 + rowoff;
    bool ok = true;
    if (tid == 0) {
        const unsigned want = p >= D.ring ? g_done_total(D, p - D.ring) : 0u;
        const unsigned *dn = D.done + (p >= D.ring ? p - D.ring : 0);
        const unsigned *g1 = D.gen + (size_t)c0 * D.nblk + b0, *g2 = D.gen + (size_t)c0 * D.nblk + c0;
        // (the slabs of the second row block have a whole step of float: they start behind those of the first, which the next
        // pivot waits for, instead of sharing the chain's compute units with them)
        const unsigned *r1 = D.rb + (size_t)p * D.nblk + b0 + 1;
        ok = spin_until(D, [&] {
            const unsigned f0 = flag_load(D.mc + p), f1 = flag_load(g1), f2 = flag_load(g2), f3 = flag_load(dn), f4 = flag_load(r1);
            return (f0 >= 1u) & (f1 >= (unsigned)p) & (f2 >= (unsigned)p) & (f3 >= want) & ((rowoff == 1) | (f4 >= 2u));
        });
    }
    if (!acquire_end(ok)) return;
    if (dbg_ready && tid == 0) *dbg_ready = wall_clock64();  // trace: the item's start stamp becomes the end of its wait
    const size_t ld = D.ld;
    const int R0 = MB * s;
    const double *Gt = D.A + (size_t)c0 * T + (size_t)b0 * T * ld;       // G(r, k) = Gt[r + k ld]
    const double *P = ring_pg(D, p);                                      // P(c, k) = P[c + k T]
    double *Ct = D.A + (size_t)c0 * T + (size_t)c0 * T * ld;
    double *Gc = D.G0 + ring_panel(D, p) * D.pstride + (size_t)c0 * T;  // the row block's G for the tile items: Gc[r + k ld]
    double *Hw = D.H0 + ring_panel(D, p) * D.pstride + (size_t)c0 * T;  // and its H: Hw[r + c ld]
    double *Hl = lds;                                                     // Hl[k][r]: 128 x 16
    // ---- H(slab, :) = -G(slab, :) P: this wave's two 16-column blocks ----
    double4_t h0 = (double4_t){0.0, 0.0, 0.0, 0.0}, h1 = (double4_t){0.0, 0.0, 0.0, 0.0};
    {
        SlabPipe<false> pipe;
        pipe.bp = Gt + (size_t)(R0 + l15) + (size_t)lq * ld;
        pipe.ldb = ld;
        pipe.ap = P + (size_t)(32 * wv + l15) + (size_t)lq * T;
        pipe.lda = T;
        pipe.prologue();
        pipe.run<true>(h0, h1, Gc + (size_t)(R0 + l15) + (size_t)lq * ld, ld, wv);
    }
    // ---- C(slab, :) += H(slab, :) G^T: the operand of G goes on its way before H changes hands ----
    SlabPipe<true> pipe2;
    pipe2.bp = Hl + lq * MB + l15;
    pipe2.ldb = MB;
    pipe2.ap = Gt + (size_t)(32 * wv + l15) + (size_t)lq * ld;  // G(c', k), c' = this wave's columns of C
    pipe2.lda = ld;
    pipe2.prologue();
    double4_t c0v, c1v;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        c0v[reg] = Ct[(size_t)(R0 + l15) + (size_t)(32 * wv + lq + 4 * reg) * ld];
        c1v[reg] = Ct[(size_t)(R0 + l15) + (size_t)(32 * wv + 16 + lq + 4 * reg) * ld];
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int c = 32 * wv + lq + 4 * reg;
        const double x0 = -h0[reg], x1 = -h1[reg];
        Hl[c * MB + l15] = x0;
        Hl[(c + 16) * MB + l15] = x1;
        store_wt(&Hw[(size_t)(R0 + l15) + (size_t)c * ld], x0);
        store_wt(&Hw[(size_t)(R0 + l15) + (size_t)(c + 16) * ld], x1);
    }
    __syncthreads();  // H(slab, :) of all four waves is in LDS
    pipe2.run<false>(c0v, c1v, nullptr, 0, wv);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        store_wt(&Ct[(size_t)(R0 + l15) + (size_t)(32 * wv + lq + 4 * reg) * ld], c0v[reg]);
        store_wt(&Ct[(size_t)(R0 + l15) + (size_t)(32 * wv + 16 + lq + 4 * reg) * ld], c1v[reg]);
    }
    publish_wt_begin();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(D.sl + (rowoff == 1 ? 0 : 2 * D.ng) + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == SLAB_ITEMS - 1) {
            // every slab's stores had been drained before its increment: the row block's panels and the tile are complete
            __hip_atomic_fetch_add(D.rb + (size_t)p * D.nblk + c0, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(D.gen + (size_t)c0 * D.nblk + c0, (unsigned)(p + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(D.done + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- single-block groups: the tile the slab items of the NEXT group wait for, one step early ---------------------------------------
// Slab item p + 1 reads G = A[c0 + 1, c0] with update p applied.  As a tile item of the main list that tile waits for the panel item
// of ITS row block (which copies G into the panel buffer), and that one for Pg(p): ~40 us behind the pivot -- the chain stood still
// for them every other group.  The tile needs neither: A[c0+1, c0] += G H^T with G = A[c0+1, b0] as it lies in A (generation p: an
// early item of update p - 1) and H = the rows the slab items of THIS group just wrote.  So it is done here, on the chain's idle
// workgroups, as SLAB_ITEMS row slabs like the slab items' second product, while the next pivot runs; the main list skips it (and its
// write-back item of that row block of column b0 waits for these items: they read it in place).
__device__ __forceinline__ void sweep_xslab_item(const SweepDesc &D, int p, int s, unsigned long long *dbg_ready = nullptr)
{
    const int tid = opaque_tid(), lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b0 = g_start(D, p), c0 = b0 + 1, I = c0 + 1;
    bool ok = true;
    if (tid == 0) {
        const unsigned *g1 = D.gen + (size_t)I * D.nblk + b0, *g2 = D.gen + (size_t)I * D.nblk + c0;
        const unsigned *rbp = D.rb + (size_t)p * D.nblk + c0;
        ok = spin_until(D, [&] {
            const unsigned f0 = flag_load(rbp), f1 = flag_load(g1), f2 = flag_load(g2);
            return (f0 >= 2u) & (f1 >= (unsigned)p) & (f2 >= (unsigned)p);
        });
    }
    if (!acquire_end(ok)) return;
    if (dbg_ready && tid == 0) *dbg_ready = wall_clock64();
    const size_t ld = D.ld;
    const int R0 = MB * s;
    const double *Gt = D.A + (size_t)I * T + (size_t)b0 * T * ld;                            // G(r, k) = Gt[r + k ld]
    const double *Hw = D.H0 + ring_panel(D, p) * D.pstride + (size_t)c0 * T;                  // H(c, k) = Hw[c + k ld]
    double *Xt = D.A + (size_t)I * T + (size_t)c0 * T * ld;
    SlabPipe<false> pipe;
    pipe.bp = Gt + (size_t)(R0 + l15) + (size_t)lq * ld;
    pipe.ldb = ld;
    pipe.ap = Hw + (size_t)(32 * wv + l15) + (size_t)lq * ld;
    pipe.lda = ld;
    pipe.prologue();
    double4_t x0, x1;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        x0[reg] = Xt[(size_t)(R0 + l15) + (size_t)(32 * wv + lq + 4 * reg) * ld];
        x1[reg] = Xt[(size_t)(R0 + l15) + (size_t)(32 * wv + 16 + lq + 4 * reg) * ld];
    }
    pipe.run<false>(x0, x1, nullptr, 0, wv);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        store_wt(&Xt[(size_t)(R0 + l15) + (size_t)(32 * wv + lq + 4 * reg) * ld], x0[reg]);
        store_wt(&Xt[(size_t)(R0 + l15) + (size_t)(32 * wv + 16 + lq + 4 * reg) * ld], x1[reg]);
    }
    publish_wt_begin();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(D.sl + D.ng + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == SLAB_ITEMS - 1) {
            __hip_atomic_store(D.gen + (size_t)I * D.nblk + c0, (unsigned)(p + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(D.done + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- one tile of update p:  A_IJ += sum_w G_w[I] H_w[J]^T ------------------------------------------------------------------
// `ready`: the flags were seen set already (by the previous item's look-ahead, see tile_item_finish).
// `nxt`, `s_next`, `s_ready` (main-list callers; thread 0's value / LDS words): the workgroup's NEXT item number, in flight as
// an atomic since the start of this item.  While this item's stores drain, thread 0 looks that item up and, if it is a tile
// item, reads its flags: the next trip then starts without the counter's and the flags' round trips (~3 us of a ~105 us item).
// `inplace` (between single blocks, early tiles below the pivot): the G operand is read where it lies, A[I, b0] at generation p, not
// from the copy the panel item of row block I makes -- the tile then does not wait for that panel item
__device__ __forceinline__ bool tile_item_wait(const SweepDesc &D, int p, int I, int J, int ready, bool inplace = false)
{
    bool ok = true;
    if (opaque_tid() == 0 && (!ready || inplace)) {
        const unsigned long long t0 = D.dbg ? wall_clock64() : 0ull;
        const unsigned need = 2u * (unsigned)g_size(D, p);
        const unsigned *genp = D.gen + (size_t)I * D.nblk + J;
        const unsigned *f1p = inplace ? D.gen + (size_t)I * D.nblk + g_start(D, p) : D.rb + (size_t)p * D.nblk + I;
        const unsigned need1 = inplace ? (unsigned)p : need;
        // the three flags are loaded TOGETHER (one L2 round trip, ~1.5 us under load, instead of three dependent ones)
        ok = spin_until(D, [&] {
            const unsigned f1 = flag_load(f1p), f2 = flag_load(D.rb + (size_t)p * D.nblk + J), f3 = flag_load(genp);
            return (f1 >= need1) & (f2 >= need) & (f3 >= (unsigned)p);
        });
        if (D.dbg) {
            if (sw.cur >= 0) D.dbg_items[4 * (size_t)sw.cur + 2] = wall_clock64();
            atomicAdd(D.dbg_main + 0, wall_clock64() - t0);
            const int r3 = g_start(D, p) + 3;
            if (D.slab && I == r3 && J >= r3 - 2) D.dbg_main[8 + 1024 + 3 * D.ng + 16 * p + 10 + (J - (r3 - 2))] = wall_clock64();  // trace: ready
        }
    }
    return acquire_end(ok);
}

// The workgroup's NEXT take: item number `nxt` (thread 0's value, in flight as an atomic since the start of this item) and where
// it goes.  main = 0: none (an M-list caller).  merged: the launch is k_sweep_merged, which takes the next item from another family
// -- that family's index and the group this workgroup's last item there belonged to wait in LDS (sw_fnext, sw_pnext: known when the
// atomic is issued; as registers of thread 0 they were live, and spilled, across the item).
struct NextTake {
    int nxt;
    int main;
    int merged;
};

// after the tile's (write-through) stores have been issued: look the next item up while they drain, then publish
__device__ __forceinline__ void tile_item_finish(const SweepDesc &D, int p, int I, int J, const NextTake &nt)
{
    const int tid = opaque_tid();
    if (nt.main && tid == 0) {
        sw_next = nt.nxt;
        const int fn = nt.merged ? sw_fnext : 0;
        if (nt.merged) sw_fam = fn;
        const SweepDesc &Dn = nt.merged ? launch_desc(fn) : D;
        int r = 0;
        if (nt.nxt < Dn.total) {
            int ph = nt.merged ? sw_pnext : p;
            const MainItem ni = main_decode(Dn, ph, nt.nxt);
            if (ni.kind == 1 || ni.kind == 4) {
                const int nsz2 = g_size(Dn, ni.p);
                const unsigned f1 = flag_load(Dn.rb + (size_t)ni.p * Dn.nblk + ni.a), f2 = flag_load(Dn.rb + (size_t)ni.p * Dn.nblk + ni.b),
                               f3 = flag_load(Dn.gen + (size_t)ni.a * Dn.nblk + ni.b);
                r = (f1 >= 2u * (unsigned)nsz2) & (f2 >= 2u * (unsigned)nsz2) & (f3 >= (unsigned)ni.p);
            }
        }
        sw_ready = r;
    }
    publish_wt_begin();
    if (tid == 0) {
        __hip_atomic_store(D.gen + (size_t)I * D.nblk + J, (unsigned)(p + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(D.done + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <bool MULTI>
__device__ __forceinline__ void sweep_tile_item(const SweepDesc &D, int p, int I, int J, double (*Gs)[KC][LDS_LD],
                                                double (*Hs)[KC][LDS_LD], int ready = 0, const NextTake &nt = NextTake{0, 0, 0}, bool inplace = false)
{
    const int tid = opaque_tid(), lane = tid & 63, wv = tid >> 6;
    const int wr = wv & 1, wc = wv >> 1, l15 = lane & 15, lq = lane >> 4;
    if (!tile_item_wait(D, p, I, J, ready, inplace)) return;
    const size_t ld = D.ld, pld = D.ld;
    // (in place: the group's column of A has the panel copy's layout, G(r, k) = A[r + k ld], the row block I rows down)
    const double *Gp = inplace ? D.A + (size_t)g_start(D, p) * T * D.ld : D.G0 + ring_panel(D, p) * D.pstride;
    const double *Hp = D.H0 + ring_panel(D, p) * D.pstride;
    double *At = D.A + (size_t)I * T + (size_t)J * T * ld;
    double4_t acc[4][4];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = (double4_t){0.0, 0.0, 0.0, 0.0};
    {
        StageRegs<4> R;
        double cp[8];
        Frag f;
        const TileOff to = tile_off(ld, wr, wc, l15, lq);
        const size_t go = (size_t)I * T, ho = (size_t)J * T;
        const ChunkIO io = chunk_io<false>(pld, pld, tid);
        const size_t cs = (size_t)KC * pld;  // one chunk further along k
        const double *g = Gp + go, *h = Hp + ho;
        // nch = chunks of the whole item (even, >= 10): 8 per pivot block, fewer for the block that holds the ragged end of the
        // matrix (its padding columns are zero in G and H: g_chunks).  Looked up here: no control flow between the first eight chunks
        const int nch = MULTI ? g_chunks(D, p) : 8;
        stage_load<false, 4>(R, g, pld, h, pld, 0, tid);
        cpiece_load<0>(cp, At, ld, to);
        stage_store<false, 4>(R, Gs[0], Hs[0], tid);
        stage_load<false, 4>(R, g, pld, h, pld, KC, tid);
        __syncthreads();
        frag_read<false>(f, Gs[0], Hs[0], 0, wr, wc, l15, lq);
        // chunk c multiplies LDS buffer c & 1, stores chunk c+1 into the other one and loads chunk c+2; the first eight chunks
        // (the group's first pivot block) also bring the C tile in
        tile_chunk<true, true, true, 0>(acc, f, R, cp, Gs[0], Hs[0], Gs[1], Hs[1], g + 2 * cs, h + 2 * cs, io, At, ld, tid, nullptr, to);
        tile_chunk<true, true, true, 1>(acc, f, R, cp, Gs[1], Hs[1], Gs[0], Hs[0], g + 3 * cs, h + 3 * cs, io, At, ld, tid, nullptr, to);
        tile_chunk<true, true, true, 2>(acc, f, R, cp, Gs[0], Hs[0], Gs[1], Hs[1], g + 4 * cs, h + 4 * cs, io, At, ld, tid, nullptr, to);
        tile_chunk<true, true, true, 3>(acc, f, R, cp, Gs[1], Hs[1], Gs[0], Hs[0], g + 5 * cs, h + 5 * cs, io, At, ld, tid, nullptr, to);
        tile_chunk<true, true, true, 4>(acc, f, R, cp, Gs[0], Hs[0], Gs[1], Hs[1], g + 6 * cs, h + 6 * cs, io, At, ld, tid, nullptr, to);
        tile_chunk<true, true, true, 5>(acc, f, R, cp, Gs[1], Hs[1], Gs[0], Hs[0], g + 7 * cs, h + 7 * cs, io, At, ld, tid, nullptr, to);
        if constexpr (MULTI) {
            // the other pivot blocks of the group: operand pair w at Gp / Hp + w * pstride
            g += D.pstride;
            h += D.pstride;
            tile_chunk<true, true, true, 6>(acc, f, R, cp, Gs[0], Hs[0], Gs[1], Hs[1], g, h, io, At, ld, tid, nullptr, to);
            tile_chunk<true, true, true, 7>(acc, f, R, cp, Gs[1], Hs[1], Gs[0], Hs[0], g + cs, h + cs, io, At, ld, tid, nullptr, to);
#pragma unroll 1
            for (int c = 8; c < nch - 2; c += 2) {
                // loads chunks c + 2 and c + 3 (the same operand pair: c is even and a pair holds 8 chunks)
                const double *gl = Gp + (size_t)((c + 2) >> 3) * D.pstride + go + (size_t)((c + 2) & 7) * cs;
                const double *hl = Hp + (size_t)((c + 2) >> 3) * D.pstride + ho + (size_t)((c + 2) & 7) * cs;
                tile_chunk<true, true, true, -1>(acc, f, R, cp, Gs[0], Hs[0], Gs[1], Hs[1], gl, hl, io, At, ld, tid);
                tile_chunk<true, true, true, -1>(acc, f, R, cp, Gs[1], Hs[1], Gs[0], Hs[0], gl + cs, hl + cs, io, At, ld, tid);
            }
            tile_chunk<true, false, true, -1>(acc, f, R, cp, Gs[0], Hs[0], Gs[1], Hs[1], g, h, io, At, ld, tid);
            tile_chunk<false, false, false, -1>(acc, f, R, cp, Gs[1], Hs[1], Gs[0], Hs[0], g, h, io, At, ld, tid);
        } else {
            tile_chunk<true, false, true, 6>(acc, f, R, cp, Gs[0], Hs[0], Gs[1], Hs[1], g, h, io, At, ld, tid, nullptr, to);
            tile_chunk<false, false, false, 7>(acc, f, R, cp, Gs[1], Hs[1], Gs[0], Hs[0], g, h, io, At, ld, tid, nullptr, to);
        }
    }
    {
        // (the offsets are computed again, from a thread index the compiler cannot connect with the one above: nothing of the
        // addressing stays live across the chunk loop)
        const int t2 = opaque_tid(), lane2 = t2 & 63, wv2 = t2 >> 6;
        const TileOff to = tile_off(ld, wv2 & 1, wv2 >> 1, lane2 & 15, lane2 >> 4);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            char *base = const_cast<char *>(tile_strip(At, ld, tm));
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) store_wt(reinterpret_cast<double *>(base + to.o[reg]) + 16 * tn, acc[tm][tn][reg]);
        }
    }
    tile_item_finish(D, p, I, J, nt);
}

// ---- a tile of the LAST block row when the matrix ends inside that block (n = 10 000: 78 blocks of 128 and 16 rows) ------------
// Only the first rl rows of the tile are real; the others are padding (zero, and they stay zero: the padding rows of H are
// zero).  The product is done for the 16-row blocks that hold real rows only (rl = 16: 8 of the 64 MFMAs of a k4 step); a
// plain loop (one LDS buffer, two barriers per chunk): these items are 1 / nblk of the tile items and bound by their operand
// traffic, not by the matrix pipe.
__device__ __forceinline__ void sweep_tile_item_ragged(const SweepDesc &D, int p, int I, int J, double (*Gs)[KC][LDS_LD],
                                                       double (*Hs)[KC][LDS_LD], int ready, const NextTake &nt)
{
    const int tid = opaque_tid(), lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv & 1, wc = wv >> 1, l15 = lane & 15, lq = lane >> 4;
    if (!tile_item_wait(D, p, I, J, ready)) return;
    const size_t ld = D.ld, pld = D.ld;
    const double *Gp = D.G0 + ring_panel(D, p) * D.pstride + (size_t)I * T, *Hp = D.H0 + ring_panel(D, p) * D.pstride + (size_t)J * T;
    double *At = D.A + (size_t)I * T + (size_t)J * T * ld;
    const int ntn = min(4, max(0, (D.rl - wr * 64) / 16));  // 16-row blocks of this wave's 64 rows that hold real rows
    const int nch = g_chunks(D, p);
    double4_t acc[4][4];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = (double4_t){0.0, 0.0, 0.0, 0.0};
    StageRegs<4> R;
    stage_load<false, 4>(R, Gp, pld, Hp, pld, 0, tid);
#pragma unroll 1
    for (int c = 0; c < nch; ++c) {
        __syncthreads();  // the previous chunk's LDS reads are done
        stage_store<false, 4>(R, Gs[0], Hs[0], tid);
        __syncthreads();
        if (c + 1 < nch)
            stage_load<false, 4>(R, Gp + (size_t)((c + 1) >> 3) * D.pstride, pld, Hp + (size_t)((c + 1) >> 3) * D.pstride, pld,
                                 ((c + 1) & 7) * KC, tid);
#pragma unroll
        for (int k4 = 0; k4 < KC; k4 += 4) {
            double a[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] = Hs[0][k4 + lq][wc * 64 + t * 16 + l15];
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
                if (tn < ntn) {
                    const double bq = Gs[0][k4 + lq][wr * 64 + tn * 16 + l15];
#pragma unroll
                    for (int tm = 0; tm < 4; ++tm) acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tm], bq, acc[tm][tn], 0, 0, 0);
                }
        }
    }
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
        if (tn < ntn) {
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    double *q = &At[(size_t)(wr * 64 + tn * 16 + l15) + (size_t)(wc * 64 + tm * 16 + lq + 4 * reg) * ld];
                    store_wt(q, *q + acc[tm][tn][reg]);
                }
        }
    tile_item_finish(D, p, I, J, nt);
}

// ---- wb(p): one tile of the group's new columns ----------------------------------------------------------------------------
__device__ __forceinline__ void sweep_wb_item(const SweepDesc &D, int p, int e, double (*Gs)[KC][LDS_LD])
{
    const int tid = opaque_tid();
    const int b0 = g_start(D, p), sz = g_size(D, p), rows = D.nblk - sz;
    int i = e % rows;
    const int w = e / rows;
    if (i >= b0) i += sz;
    const int k = b0 + w;
    bool ok = true;
    if (tid == 0) {
        // (between single blocks the chain's xslab items of this group read row block b0 + 2 of the column in place: not before they are done)
        // and so do the early tiles (i, b0 + 1) and (i, b0 + 2) of this update (sweep_tile_item, inplace)
        const bool after_x = D.slab && i >= b0 + 2, after_y = D.slab && i >= b0 + 3;
        const unsigned *gx = D.gen + (size_t)i * D.nblk + (after_x ? b0 + 1 : 0), *gy = D.gen + (size_t)i * D.nblk + (after_y ? b0 + 2 : 0);
        ok = spin_until(D, [&] {
            const unsigned f0 = flag_load(D.rb + (size_t)p * D.nblk + i), f1 = flag_load(gx), f2 = flag_load(gy);
            return (f0 >= 2u * (unsigned)sz) & (!after_x | (f1 >= (unsigned)(p + 1))) & (!after_y | (f2 >= (unsigned)(p + 1)));
        });
    }
    if (!acquire_end(ok)) return;
    if (D.dbg && tid == 0 && sw.cur >= 0) D.dbg_items[4 * (size_t)sw.cur + 2] = wall_clock64();
    panel_writeback_tile(D.A, D.ld, k, i, D.H0 + (ring_panel(D, p) + w) * D.pstride, D.ld, Gs[0]);
    publish_begin();
    if (tid == 0) {
        const int I = i > k ? i : k, J = i > k ? k : i;
        __hip_atomic_store(D.gen + (size_t)I * D.nblk + J, (unsigned)(p + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(D.done + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- item functions kept OUT OF LINE ---------------------------------------------------------------------------------------------
// The persistent loops below hold the hot items inline -- the tile item on the hand-scheduled chunk loop and the panel item that
// shares it -- and nothing else: every other kind of item is a function of its own, with registers of its own.  Inlined into one
// 256-VGPR kernel, the cold kinds' address arithmetic, descriptors and loop state were live (or spilled) across the chunk loops:
// 132 spilled VGPRs, scratch loads and stores inside the 512-MFMA blocks, and a kernel whose spill placement changed with every edit
// (a wrong-result incident that moved with it).  Arguments are the family's index in the launch and plain item
// numbers (made uniform again on entry: arguments arrive in vector registers); the callee finds the descriptor and the LDS itself.
// A call costs the callee's saves of the registers it uses (~1 us for a 100-register item): the items behind calls take 25 us and more.
// (GDCA_EXP_LATE_NEXT, experiments only: tile items leave the publication of the next take to the caller, behind the item -- the
// form of the merged loop that gave wrong inverses in round 4; tools/build_variant.sh late -DGDCA_EXP_LATE_NEXT)
#ifdef GDCA_EXP_LATE_NEXT
#define SWEEP_TILE_PUBLISHES 0
#else
#define SWEEP_TILE_PUBLISHES 1
#endif
#define SWEEP_OUTLINE __device__ __attribute__((noinline))
#define UNI(x) __builtin_amdgcn_readfirstlane(x)

// the M list of family f (the serial chain), run by the workgroups of the compute units elected for it: ONE call per workgroup
// (the 16-step micro-sweep of the pivot and the slab items stay inline in here, next to each other -- the chain's latency is theirs)
template <bool MULTI>
SWEEP_OUTLINE void sweep_chain_worker(int f_in)
{
    const SweepDesc &D = launch_desc(UNI(f_in));
    double(*const Gs)[KC][LDS_LD] = sw_lds;
    double(*const Hs)[KC][LDS_LD] = sw_lds + 2;
    int q = 0;
    for (;;) {
        if (threadIdx.x == 0) sw_item = (int)__hip_atomic_fetch_add(D.next_m, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int item = sw_item;
        __syncthreads();
        if (item >= D.total_m || *abort_lds()) break;
        while (item >= D.mitem0[q + 1]) ++q;
        int e = item - D.mitem0[q];
        if (D.dbg && threadIdx.x == 0) D.dbg[2 * item] = wall_clock64();
        const int b0 = g_start(D, q), sz = g_size(D, q), c0 = b0 + sz;
        const int nm = m_items(sz);
        if (e < nm) {
            sweep_m_item(D, q, e, Gs, Hs, D.dbg ? D.dbg + 2 * item : nullptr);
            if (D.dbg && threadIdx.x == 0) D.dbg[2 * item + 1] = wall_clock64();
            continue;
        }
        e -= nm;
        if (D.slab) {
            if (e < SLAB_ITEMS)
                sweep_slab_item(D, q, e, 1, &Gs[0][0][0], D.dbg ? D.dbg + 2 * item : nullptr);
            else if (e < 2 * SLAB_ITEMS)
                sweep_xslab_item(D, q, e - SLAB_ITEMS, D.dbg ? D.dbg + 2 * item : nullptr);
            else
                sweep_slab_item(D, q, e - 2 * SLAB_ITEMS, 2, &Gs[0][0][0], D.dbg ? D.dbg + 2 * item : nullptr);
            if (D.dbg && threadIdx.x == 0) D.dbg[2 * item + 1] = wall_clock64();
            continue;
        }
        // the next group's rows, one after the other: the 2 sz panel items of row rr, then the tiles (rr, 0 .. rr) of its
        // diagonal super-block -- its FIRST diagonal tile, which the next group's first pivot waits for, is complete after one
        // round of panel items instead of after all of them
        int rr = 0;
        while (e >= 2 * sz + rr + 1) {
            e -= 2 * sz + rr + 1;
            ++rr;
        }
        if (e < 2 * sz) {
            sweep_panel_item<2>(D, q, c0 + rr, e, Gs, Hs);
            if (D.dbg && threadIdx.x == 0) D.dbg[2 * item + 1] = wall_clock64();
            continue;
        }
        e -= 2 * sz;
        if (MULTI && sz > 1)
            sweep_tile_item<true>(D, q, c0 + rr, c0 + e, Gs, Hs);
        else
            sweep_tile_item<false>(D, q, c0 + rr, c0 + e, Gs, Hs);
        if (D.dbg && threadIdx.x == 0) D.dbg[2 * item + 1] = wall_clock64();
    }
}

SWEEP_OUTLINE void sweep_wb_item_ol(int f, int p, int e)
{
    sweep_wb_item(launch_desc(UNI(f)), UNI(p), UNI(e), sw_lds);
}

// (nxt is thread 0's value: it stays a vector argument)
SWEEP_OUTLINE void sweep_ragged_item_ol(int f, int p, int I, int J, int rdy, int nxt, int merged)
{
    const NextTake nt{nxt, SWEEP_TILE_PUBLISHES, UNI(merged)};
    sweep_tile_item_ragged(launch_desc(UNI(f)), UNI(p), UNI(I), UNI(J), sw_lds, sw_lds + 2, UNI(rdy), nt);
}

// a K = 128 tile item inside a launch of multi-block groups (the single blocks of its opening ramp) -- and every panel item of such
// a launch that comes as two 128 x 64 halves
SWEEP_OUTLINE void sweep_single_tile_item_ol(int f, int p, int I, int J, int rdy, int nxt, int merged, int inplace)
{
    const NextTake nt{nxt, SWEEP_TILE_PUBLISHES, UNI(merged)};
    sweep_tile_item<false>(launch_desc(UNI(f)), UNI(p), UNI(I), UNI(J), sw_lds, sw_lds + 2, UNI(rdy), nt, UNI(inplace) != 0);
}

SWEEP_OUTLINE void sweep_half_panel_item_ol(int f, int p, int i, int y)
{
    sweep_panel_item<2>(launch_desc(UNI(f)), UNI(p), UNI(i), UNI(y), sw_lds, sw_lds + 2);
}

SWEEP_OUTLINE void sweep_full_panel_item_ol(int f, int p, int i, int y)
{
    sweep_panel_item<4>(launch_desc(UNI(f)), UNI(p), UNI(i), UNI(y), sw_lds, sw_lds + 2);
}

// One item of the main list of family f, by the calling workgroup.  nt: where its next take goes (tile items publish it themselves,
// inside tile_item_finish; for the other kinds the caller does, after the item).  Returns true when the item published the take.
template <bool MULTI>
__device__ __forceinline__ bool sweep_main_item(const SweepDesc &D, int f, const MainItem &it, int rdy, const NextTake &nt_in)
{
    double(*const Gs)[KC][LDS_LD] = sw_lds;
    double(*const Hs)[KC][LDS_LD] = sw_lds + 2;
    const NextTake nt{nt_in.nxt, SWEEP_TILE_PUBLISHES, nt_in.merged};
    if (it.kind == 1 || it.kind == 4) {
        if (D.rl < T && it.a == D.nblk - 1)
            sweep_ragged_item_ol(f, it.p, it.a, it.b, rdy, nt.nxt, nt.merged);
        else if (MULTI && g_size(D, it.p) > 1)
            sweep_tile_item<true>(D, it.p, it.a, it.b, Gs, Hs, rdy, nt);
        else {
            const bool inplace = D.slab && it.kind == 4 && it.a > it.b && it.a > g_start(D, it.p) + 1;
            if (MULTI)
                sweep_single_tile_item_ol(f, it.p, it.a, it.b, rdy, nt.nxt, nt.merged, inplace ? 1 : 0);
            else
                sweep_tile_item<false>(D, it.p, it.a, it.b, Gs, Hs, rdy, nt, inplace);
        }
        return SWEEP_TILE_PUBLISHES != 0;
    }
    if (it.kind == 0) {
        // the form the item table was built for (D.ppb): 128 x 128 items inline where they are the rule (multi-block groups), 128 x 64
        // halves inline where THEY are (single blocks); the other form of either launch -- a forced PANEL_HALVES, the groups of
        // two -- out of line
        if (D.ppb == 1) {
            if (MULTI)
                sweep_panel_item<4>(D, it.p, it.a, it.b, Gs, Hs);
            else
                sweep_full_panel_item_ol(f, it.p, it.a, it.b);
        } else if (MULTI)
            sweep_half_panel_item_ol(f, it.p, it.a, it.b);
        else
            sweep_panel_item<2>(D, it.p, it.a, it.b, Gs, Hs);
    } else if (it.kind == 2)
        sweep_wb_item_ol(f, it.p, it.a);
    return false;
}

// trace (between single blocks): the slot of a main-list item of row block b0 + 3 of update it.p in the group's trace record, or -1
__device__ __forceinline__ int sweep_trace_xslot(const SweepDesc &D, const MainItem &it)
{
    const int r3 = g_start(D, it.p) + 3;
    if (it.kind == 0 && it.a == r3) return 0;
    if (it.kind == 4 && it.a == r3) return 3 + 2 * (it.b - (r3 - 2));
    return -1;
}

// the chain's compute units: a workgroup on XCC x offers itself to family (x + t) % K, t = 0, 1, ..: the first workgroup to reach a
// family decides its XCD, so with K <= 8 the chains sit on different XCDs (each hands its data on through one L2) whenever the
// workgroups of a launch are spread over them.  Thread 0; returns the family this workgroup's CU was elected for, or -1.
__device__ __forceinline__ int sweep_elect(int K)
{
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    xcc &= 15u;
    const unsigned key = 1u + (((hw >> 8) & 0xFFu));  // cu_id, sh_id, se_id
    const unsigned first = xcc;
    if (K > 1) {
        // A merged launch that shares the device with another persistent launch may have a few dozen workgroups resident for a long
        // time -- fewer than its members' chains have seats (up to 64).  Left to the first-come rule alone they ALL became chain
        // workers, nobody served the main lists the chains wait for, and the launch stood still until the watchdog ended it (seen in
        // round 4's batch driver).  The FIRST workgroup of the launch to show up therefore stays a main-list worker: with it the
        // lists move, however slowly, until the rest of the grid gets its compute units.  (Every fourth of the first 32 arrivals, tried
        // first, cost the merged launch of config B 10 %: chain workers then share compute units with tile items.)
        const unsigned a = __hip_atomic_fetch_add(launch_desc_k(0).arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a == 0u) return -1;
    }
    for (int t = 0; t < K; ++t) {
        const int f = (int)((first + (unsigned)t) % (unsigned)K);
        const SweepDesc &Df = launch_desc_k(f);
        // the XCD of a chain is whichever one the first workgroup to get there runs on (not a fixed id: under a CU mask, in a
        // partitioned mode or beside another tenant no workgroup of the launch may ever run on XCC 0)
        const bool candidate = !(Df.debug & 2) && !((Df.debug & 1) && xcc == 0u) && !((Df.debug & 4) && xcc != 0u);
        if (!candidate) continue;
        const unsigned old = atomicCAS(Df.mxcc, 0u, xcc + 1u);
        if (old == 0u || old == xcc + 1u) {
            for (int k = 0; k < Df.n_mcu; ++k) {
                const unsigned o2 = atomicCAS(Df.mcu + k, 0u, key);
                if (o2 == 0u) return f;
                // the OTHER workgroup of an elected compute unit: a chain worker too -- or (GDCA_MCU_SOLO) nobody: it leaves, and the
                // chain's items have the unit's SIMDs, LDS and L1 to themselves
                if (o2 == key) return Df.mcu_solo ? -2 - f : f;
            }
        }
    }
    return -1;
}

template <bool MULTI>
__global__ __launch_bounds__(256, 2) void k_sweep(const SweepDesc Darg)
{
    // the descriptor is read where it lies, in the kernel-argument segment
    const SweepDesc &D = launch_desc_k(0);
    (void)Darg;
    // ---- the M list runs on compute units of its own ----
    // The serial chain of a group (128 dependent steps per pivot block, each a handful of VALU / DPP / MFMA instructions) is
    // several times slower when its waves share their SIMDs with the MFMA stream of a tile item, and then IT sets the pace
    // of the whole inverse.  So a few compute units of ONE XCD (n_mcu, up to 16) are elected at run time -- the first ones that show up;
    // both workgroups of an elected CU become M workers -- and take only M items (their own counter); one XCD, so that the
    // items of a chain hand their data on through one L2.  Everybody else takes the main list.  The elected CUs are lost to
    // the tiles (2 n_mcu of 512 workgroups) and join them once the M list is exhausted.
    if (threadIdx.x == 0) {
        launch_desc_init();
        *abort_lds() = 0;
        sw.hw_home = sweep_hw_id();
        sw_item = sweep_elect(1);
    }
    __syncthreads();
    if (sw_item <= -2) return;  // (GDCA_MCU_SOLO: the second workgroup of a chain compute unit)
    const bool m_worker = sw_item >= 0;
    // the clock this launch really ran at (the governor moves it between 1.7 and 2.4 GHz, and not every XCD need run at the
    // same one): every workgroup times itself, the sums give the workgroup-time-weighted average over the chip
    // (thread 0's start stamps wait in LDS: in registers they would be live -- spilled -- across everything below)
    if (threadIdx.x == 0) {
        sw_probe[0] = (unsigned long long)clock64();
        sw_probe[1] = wall_clock64();
    }
    __syncthreads();
    if (threadIdx.x == 0) sw.cur = -1;
    if (m_worker) sweep_chain_worker<MULTI>(0);
    int p = 0;  // group whose sequence the last item belonged to (items come in ascending order)
    // the NEXT item is requested while the current one is being worked on (the returning atomic takes a microsecond or two
    // under load); its number stays in a register of thread 0 until the end of the item (so that nobody waits for it) and
    // reaches the others through sw_next; sw_ready = the next item is a tile item whose flags were already seen set
    if (threadIdx.x == 0) {
        sw_next = (int)__hip_atomic_fetch_add(D.next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sw_ready = 0;
    }
    for (;;) {
        __syncthreads();
        const int item = UNI(sw_next), rdy = UNI(sw_ready);  // (scalar: as vector registers they were live -- spilled -- across the item)
        __syncthreads();  // everybody has read them
        if (item >= D.total || *abort_lds()) break;
        int nxt = 0;
        if (threadIdx.x == 0) nxt = (int)__hip_atomic_fetch_add(D.next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (trace: the item's start stamps wait in LDS, not in registers of thread 0 that would be live across the item)
        if (D.dbg && threadIdx.x == 0) {
            sw_stamp[0] = wall_clock64();
            sw_stamp[1] = (unsigned long long)clock64();
        }
        const MainItem it = main_decode(D, p, item);
        // trace (between single blocks): when the main list's items of row block b0 + 3 of update it.p were taken and done -- the
        // inputs the chain waits for: [0..2] its panel halves (first taken, last done), [3..] tiles (b0+3, b0+1), (b0+3, b0+2), (b0+3, b0+3)
        if (D.dbg && D.slab && threadIdx.x == 0) {
            const int xslot = sweep_trace_xslot(D, it);
            if (xslot >= 0) {
                unsigned long long *xs = D.dbg_main + 8 + 1024 + 3 * D.ng + 16 * it.p;
                if (xslot == 0) atomicCAS(xs + 0, 0ull, wall_clock64()); else xs[xslot] = wall_clock64();
            }
        }
        const NextTake nt{nxt, 1, 0};
        if (D.dbg && threadIdx.x == 0) {
            sw.cur = item;
            D.dbg_items[4 * (size_t)item] = ((unsigned long long)(unsigned)it.kind << 56) | ((unsigned long long)(unsigned)it.p << 44) | ((unsigned long long)(unsigned)it.a << 32) |
                                            ((unsigned long long)(unsigned)it.b << 20) | (unsigned long long)blockIdx.x;
            D.dbg_items[4 * (size_t)item + 1] = sw_stamp[0];
        }
        const bool published = sweep_main_item<MULTI>(D, 0, it, rdy, nt);
        if (D.dbg && threadIdx.x == 0) D.dbg_items[4 * (size_t)item + 3] = wall_clock64();
        if (D.dbg && threadIdx.x == 0) {
            const unsigned long long t_item = sw_stamp[0], c_item = sw_stamp[1];
            const int xslot = D.slab ? sweep_trace_xslot(D, it) : -1;
            if (it.kind == 1 || it.kind == 4) {
                atomicAdd(D.dbg_main + 6, wall_clock64() - t_item);
                atomicAdd(D.dbg_main + 7, 1ull);
                const unsigned long long cyc = (unsigned long long)clock64() - c_item;
                atomicAdd(D.dbg_main + 8 + 1023, cyc);  // shader-clock cycles of the tile items
                atomicAdd(D.dbg_main + 8 + 1024 + 3 * it.p, cyc);  // ... and per group: cycles, ticks, items
                atomicAdd(D.dbg_main + 8 + 1024 + 3 * it.p + 1, wall_clock64() - t_item);
                atomicAdd(D.dbg_main + 8 + 1024 + 3 * it.p + 2, 1ull);
                if (xslot >= 3) D.dbg_main[8 + 1024 + 3 * D.ng + 16 * it.p + xslot + 1] = wall_clock64();
            } else if (it.kind == 0) {
                atomicAdd(D.dbg_main + 2, wall_clock64() - t_item);
                atomicAdd(D.dbg_main + 3, 1ull);
                if (xslot == 0) atomicMax(D.dbg_main + 8 + 1024 + 3 * D.ng + 16 * it.p + 1, wall_clock64());
            } else if (it.kind == 2) {
                atomicAdd(D.dbg_main + 4, wall_clock64() - t_item);
                atomicAdd(D.dbg_main + 5, 1ull);
            }
        }
        if (!published && threadIdx.x == 0) {
            sw_next = nxt;
            sw_ready = 0;
        }
    }
    if (threadIdx.x == 0) {
        atomicAdd(&D.sc->sweep_cycles, (unsigned long long)clock64() - sw_probe[0]);
        atomicAdd(&D.sc->sweep_ticks, wall_clock64() - sw_probe[1]);
    }
    if (D.dbg && threadIdx.x == 0) D.dbg_main[8 + blockIdx.x] = wall_clock64();  // when this workgroup ran out of work
}

// =====================================================================================================================
// K independent inverses in ONE persistent launch (gdca_run_dev_phased: families batched by phase)
// =====================================================================================================================
// A matrix of a few dozen blocks cannot fill the chip: between single blocks a step of the sweep is a serial chain of ~45-60 us
// (pivot, the next row's slabs, flag hops) with at most nblk^2 / 2 tile items of ~30 us beside it -- 200 at 20 blocks for 512
// workgroup slots -- and a second k_sweep launch cannot move in beside the first, which holds two 74-KB workgroups on every
// compute unit.  So the launch itself carries K families: one descriptor, item table and flag block per family (exactly what a
// launch of its own would get, so the arithmetic of a family -- every tile's update order is fixed by ITS flags -- is bit for bit
// that of a single run), and
//   * every family's chain gets compute units of its own (sweep_elect);
//   * everybody else takes main-list items of the families in turn (its j-th take goes to the next family that still has
//     items): the families advance side by side, and a workgroup parked on an item of family A that waits for A's chain keeps
//     nothing of family B from running.
// Items of one family are still handed out in that family's list order, so the no-deadlock argument of k_sweep holds family by
// family.  With the chains of K families side by side a member's chain no longer sets the pace -- it has the other members' tile
// items to hide behind -- so a member of a merged launch takes the multi-block groups (tile items of depth K = 256 .. 512: a half
// to a quarter of the C-tile traffic per flop) that pay only from 49 .. 58 blocks on when it runs alone (plan_sweep).  MULTI:
// some member has multi-block groups.
#define SWEEP_MAX_MERGE 8
struct SweepBatch {
    SweepDesc fam[SWEEP_MAX_MERGE];
    int K;
};
static_assert(offsetof(SweepBatch, fam) == 0, "launch_desc(f) finds member f at the start of the kernel-argument segment");
static_assert(SWEEP_MAX_MERGE <= 8, "SweepShared::p");

template <bool MULTI>
__global__ __launch_bounds__(256, 2) void k_sweep_merged(const SweepBatch Barg)
{
    (void)Barg;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const SweepBatch __attribute__((address_space(4))) *kernarg_batch_t;
    const int K = ((const SweepBatch *)(kernarg_batch_t)__builtin_amdgcn_kernarg_segment_ptr())->K;
#else
    const int K = Barg.K;
#endif
    if (threadIdx.x == 0) {
        launch_desc_init();
        *abort_lds() = 0;
        sw.hw_home = sweep_hw_id();
        sw_item = sweep_elect(K);
        for (int f = 0; f < SWEEP_MAX_MERGE; ++f) sw_p[f] = 0;
    }
    __syncthreads();
    const int wfam = __builtin_amdgcn_readfirstlane(sw_item);
    if (threadIdx.x == 0) {
        sw_probe[0] = (unsigned long long)clock64();
        sw_probe[1] = wall_clock64();
    }
    __syncthreads();
    if (wfam >= 0) sweep_chain_worker<MULTI>(wfam);
    // ---- the main lists, the families in turn ----
    // Which family the next take goes to is thread 0's business (sw_live: the families whose list this workgroup has not yet seen
    // exhausted); the workgroup learns the family of the current item from sw_fam, next to its number in sw_next.  As in k_sweep the
    // next item is requested while the current one is worked on, its number reaches LDS inside the item (tile_item_finish: with
    // the look-ahead at that item's flags), and per family the items of a workgroup ascend (sw_p: the group its last item there
    // belonged to).
    if (threadIdx.x == 0) {
        sw_live = (int)((1u << K) - 1u);
        const int f0 = (int)(blockIdx.x % (unsigned)K);
        sw_fam = f0;
        sw_next = (int)__hip_atomic_fetch_add(launch_desc_k(f0).next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sw_ready = 0;
    }
    for (;;) {
        __syncthreads();
        const int item = UNI(sw_next), rdy = UNI(sw_ready);
        const int f = __builtin_amdgcn_readfirstlane(sw_fam);
        __syncthreads();  // everybody has read them
        if (f < 0 || *abort_lds()) break;
        const SweepDesc &D = launch_desc_k(f);
        if (item >= D.total) {
            // this family's list is exhausted: strike it and take from the next one that is not (none left: f = -1 ends the loop)
            if (threadIdx.x == 0) {
                const unsigned live = (unsigned)sw_live & ~(1u << f);
                sw_live = (int)live;
                int fn = -1;
                if (live) {
                    fn = f;
                    do fn = fn + 1 == K ? 0 : fn + 1; while (!((live >> fn) & 1u));
                    sw_next = (int)__hip_atomic_fetch_add(launch_desc_k(fn).next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                sw_fam = fn;
                sw_ready = 0;
            }
            continue;
        }
        int nxt = 0;
        int p = sw_p[f];
        const MainItem it = main_decode(D, p, item);
        if (threadIdx.x == 0) {
            const unsigned live = (unsigned)sw_live;
            int fn = f;
            do fn = fn + 1 == K ? 0 : fn + 1; while (!((live >> fn) & 1u));
            nxt = (int)__hip_atomic_fetch_add(launch_desc_k(fn).next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sw_p[f] = p;  // (the others read it again only after the barriers at the top of the loop, or see this very value)
            sw_fnext = fn;
            sw_pnext = sw_p[fn];  // (fn == f: the value just written)
        }
        const NextTake nt{nxt, 1, 1};
        const bool published = sweep_main_item<MULTI>(D, f, it, rdy, nt);
        if (!published && threadIdx.x == 0) {
            sw_next = nxt;
            sw_fam = sw_fnext;
            sw_ready = 0;
        }
    }
    if (threadIdx.x == 0) {
        const unsigned long long cyc = (unsigned long long)clock64() - sw_probe[0], tk = wall_clock64() - sw_probe[1];
        const bool aborted = *abort_lds() != 0;
        for (int k = 0; k < K; ++k) {
            const SweepDesc &Dk = launch_desc_k(k);
            atomicAdd(&Dk.sc->sweep_cycles, cyc);
            atomicAdd(&Dk.sc->sweep_ticks, tk);
            // a wait of ANY member ran out of time: this workgroup has dropped its items, so no member's result can be trusted
            if (aborted) {
                __hip_atomic_store(Dk.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&Dk.sc->info, (int)0x80000000, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

size_t gdca_inverse_flag_bytes(int n_pad)
{
    const size_t nblk = (size_t)(n_pad / T);
    // gen, rb (ng <= nblk), mc, done, sl | next, next_m, mxcc, arrived, mcu[32] | the abort word on a 128-byte line of its own (every wait
    // of the kernel reads it; the line of the item counters is busy with atomics)
    return (nblk * nblk + nblk * nblk + 5 * nblk + 96) * sizeof(unsigned);
}

int gdca_inverse_max_merge(void)
{
    return SWEEP_MAX_MERGE;
}

// Host side: the schedule of one inverse -- group sizes, the item table (written to the job's pinned staging buffer and sent to
// the device on s0), the flags (zeroed on s0), the descriptor.  `merged`: the inverse is a member of a merged launch of
// `members` families (single-block groups, chain CUs from the merge rule, no trace).
// what a launch has to put in place for one inverse before its workgroups start: the flags zeroed, the item table on the device
struct SweepPrepOne {
    unsigned *flags;
    unsigned flag_words;
    const int *items_host;  // pinned host memory, as the device addresses it
    int *items_dev;
    int n_items;
};
struct SweepPrep {
    SweepPrepOne fam[SWEEP_MAX_MERGE];
    int K;
};
// ONE small launch for all members instead of a hipMemcpyAsync and a hipMemsetAsync of the runtime's per member (sixteen operations of
// ~12 us each in front of a merged launch of eight: 190 us of a 3.6-ms launch, profiles/r05_B_merged8_timeline.log)
__global__ __launch_bounds__(256) void k_sweep_prep(const SweepPrep P)
{
    const SweepPrepOne &f = P.fam[blockIdx.y];
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < f.flag_words; i += gridDim.x * 256u) f.flags[i] = 0u;
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < f.n_items; i += 256) f.items_dev[i] = f.items_host[i];
}

struct SweepPlan {
    SweepPrepOne prep;
    SweepDesc D;
    int g;
    long long mpos;
    double chunks;     // 128 x 128 x 16 MFMA chunk products issued (tile, panel and super-block items)
    bool slab;
    std::vector<int> mit;
};

static SweepPlan plan_sweep(hipStream_t s0, const gdca_inverse_job &job, bool merged, int members)
{
    const gdca_tuning &tu = *job.tune;
    const gdca_inverse_ws &ws = job.ws;
    const int n_pad = job.n_pad, n_real = job.n_real;
    const int nblk = n_pad / T;
    // pivot blocks per group: more blocks per pass raise the update's arithmetic intensity (K = 128 g) and amortise the per-item
    // costs; the serial chain of a group grows with g (and has the group's whole update to hide behind).  Small matrices
    // are bound by the chain itself, which is shortest with single blocks (no scratch copy, no tile jobs).  Measured on
    // MI355X with the round-3 tile loop and chain (tools/sweep_groups.py, profiles/r03_sweep_groups*.log: inverse time over
    // N x g x chain CUs): g = 1 is fastest up to 48 blocks (beyond, its K = 128 updates are bound by the traffic of the C tiles, not
    // by the chain), 2 to 54, 3 to 57, 4 from 58 on (with the super-block inverse as row-slab jobs; as half-tile jobs its chain hid only
    // from 71 blocks).
    // A member of a merged launch: its chain hides behind the other members' tile items, so what counts is the traffic of the C
    // tiles -- groups of four from 24 blocks, of two from 12 (option GDCA_MERGE_GROUP forces a size).
    const int g_merged = tu.merge_group >= 1 ? std::min(tu.merge_group, 4) : (nblk >= 24 ? 4 : (nblk >= 12 ? 2 : 1));
    // Round 5 measured the final kernel again, every block count from 44 to 77, the four sizes alternating inside one process, twice in
    // opposite order (tools/option_probe.py, profiles/r05_group_rule.log; the two passes agree to 1 %): 1 up to 46 blocks, 2 up to 53, 3 up to
    // 59, 4 from 60 (rounds 3-4: 2 from 49, 3 from 55, 4 from 58; 3-4 % at 47, 48, 54, 58 and 59 blocks) ...
    // ... and once more with one chain worker per compute unit for groups of two and three (below, profiles/r05_group_rule.log, second
    // part): 1 up to 44 blocks, 2 up to 52, 3 up to 60, 4 from 61.
    const int g_rule = nblk >= 61 ? 4 : (nblk >= 53 ? 3 : (nblk >= 45 ? 2 : 1));
    int g = merged ? g_merged : (tu.group >= 1 ? std::min(tu.group, 4) : g_rule);
    if (nblk < 2 * g) g = 1;
    // group sizes.  Before the first update there is nothing to hide the first chain behind: with full groups from the start
    // every workgroup waits ~400 us (2 % of the inverse at n = 10 000) for the first super-block inverse.  So the sweep opens
    // with a ramp 1, 2, .. g-1 (a single block is one in-place pivot item, ~70 us; each following chain is as long as the
    // update before it), and the remainder of nblk / g joins the ramp instead of ending the sweep as a lone short group
    // (measured at n = 10 000: 17.9-18.1 ms against 18.2-18.3 with the remainder last and 18.2 without the ramp).
    // GDCA_RAMP=0: uniform groups, the remainder last.
    std::vector<int> sizes;
    {
        const int ramp_sum = g * (g - 1) / 2;
        if (tu.ramp && g > 1 && nblk >= ramp_sum + 2 * g) {
            for (int k = 1; k < g; ++k) sizes.push_back(k);
            const int rest = nblk - ramp_sum, rem = rest % g;
            if (rem) sizes.insert(std::upper_bound(sizes.begin(), sizes.end(), rem), rem);
            for (int k = 0; k < rest / g; ++k) sizes.push_back(g);
        } else {
            for (int b = 0; b < nblk; b += g) sizes.push_back(std::min(g, nblk - b));
        }
    }
    const int ng = (int)sizes.size();
    // item table on the host, then to the device (pinned staging buffer of the workspace)
    int *it = ws.item0_host;
    auto size = [&](int p) { return sizes[(size_t)p]; };
    auto m_cnt = [&](int sz) { return sz == 1 ? 1 : sz * (sz + 1) + sz * (1 + SLAB_ITEMS * (sz - 1)); };
    int *mit = it + (ng + 1);
    int *gs = it + 2 * (ng + 1);
    gs[0] = 0;
    for (int p = 0; p < ng; ++p) gs[p + 1] = gs[p] + sizes[(size_t)p];
    long long pos = 0, mpos = 0;
    // real rows of the last block, in 16-row MFMA blocks: the sweep treats the matrix as n rounded up to 16, not to 128 -- the
    // tile items of the last block row and the k loop over the last pivot block skip the rest of the padding (GDCA_RAGGED=0: off)
    const int rl = tu.ragged ? std::min(T, ((n_real - (nblk - 1) * T + 15) / 16) * 16) : T;
    double chunks = 0.0;
    // remainder tiles of update p listed after panel(p+1): about one round of the workgroups, so that the panels are complete
    // when the first tile items of update p+1 are handed out and the chain has had most of update p to produce Pg(p+1).
    // Only where the update hides the chain (groups of three and four): on a chain-bound matrix Pg(p+1) is late anyway and
    // workgroups parked on panel items are missing from update p (measured: config E 59.5 instead of 63.4 families/s).
    const int rem_tail = tu.rem_tail >= 0 ? tu.rem_tail : (g >= 3 ? 2 * ws.update_cus : 0);
    // main-list panel items: one 128 x 128 item per pivot block and row where the panels are throughput (multi-block groups of
    // three and four), two 128 x 64 halves where their latency counts
    const int ppb = tu.panel_halves >= 0 ? (tu.panel_halves ? 2 : 1) : (g >= 3 ? 1 : 2);
    // single-block groups: the chain between two pivots as fused row-slab items (sweep_slab_item; GDCA_SLAB=0: panel and tile items)
    bool slab = tu.slab != 0;
    bool single = true;
    for (int p = 0; p < ng; ++p) single = single && size(p) == 1;
    slab = slab && single;
    // buffers per kind (ring_panel): eight one-panel slots between single blocks (GDCA_RING=2: two, as for multi-block groups)
    const int ring = single && tu.ring >= 3 ? std::min(tu.ring, 8) : 2;
    const int pro = ng > 0 ? (nblk - size(0) - (ng > 1 ? size(1) : 0)) * ppb * size(0) : 0;  // panel(0)
    for (int p = 0; p < ng; ++p) {
        it[p] = (int)pos;
        mit[p] = (int)mpos;
        const int sz = size(p), nsz = p + 1 < ng ? size(p + 1) : 0, n2 = p + 2 < ng ? size(p + 2) : 0;
        const int nrest = nblk - sz - nsz;
        // M(p), then the next group's panel rows and its diagonal tiles -- or, between single blocks, the fused slab items
        mpos += m_cnt(sz) + (slab ? (nsz ? SLAB_ITEMS : 0) + (n2 ? 2 * SLAB_ITEMS : 0) : nsz * 2 * sz + nsz * (nsz + 1) / 2);
        pos += slab ? (gs[p] + 3 < nblk ? 2 : 0) + std::max(0, nblk - gs[p] - 4) : n2 * (n2 + 1) / 2;  // diag2 (between single blocks: (b0+3, b0+3) and column b0 + 2)
        pos += (long long)nsz * nrest;                             // rest
        pos += (long long)(nblk - sz) * sz;                        // wb
        pos += nrest > 0 ? (long long)nrest * (nrest + 1) / 2 : 0; // rem (its diag2 tiles are empty items)
        pos += nsz > 0 ? (long long)(nblk - nsz - n2) * ppb * nsz : 0;  // panel(p+1), the rows outside groups p+1 and p+2
        const long long pn = nblk - sz;
        const bool last = gs[p + 1] == nblk;
        const int kch = (rl < T && last && sz > 1) ? 8 * (sz - 1) + 2 * ((rl + 31) / 32) : 8 * sz;  // = g_chunks()
        const double nrag = (rl < T && !last) ? (double)pn : 0.0;  // tile items of the ragged block row: rl / 128 of the MFMAs
        chunks += ((double)(pn * (pn + 1) / 2) - nrag + nrag * rl / T) * kch;  // tile items
        chunks += (double)pn * sz * 8 * sz;                                    // panel items (K = 128 sz, 128 sz columns)
        if (sz > 1) chunks += (double)sz * (sz - 1) * sz * 8;  // super-block inverse: per block and other row, sz products of 128 x 128 x 128
    }
    it[ng] = (int)pos;
    mit[ng] = (int)mpos;
    SweepPlan P{};
    P.prep = SweepPrepOne{ws.flags, (unsigned)(ws.flags_bytes / sizeof(unsigned)), ws.item0_host_dev, ws.item0_dev, 3 * (ng + 1)};
    if (!ws.item0_host_dev) {  // (no device address for the pinned table: the runtime's copy, and the prep kernel only clears)
        (void)hipMemcpyAsync(ws.item0_dev, it, (size_t)3 * (ng + 1) * sizeof(int), hipMemcpyHostToDevice, s0);
        P.prep.n_items = 0;
    }
    SweepDesc &D = P.D;
    D.A = job.A;
    D.ld = (size_t)n_pad;
    D.nblk = nblk;
    D.g = g;
    D.ng = ng;
    D.G0 = ws.G[0];
    D.H0 = ws.H[0];
    D.pstride = (size_t)(ws.G[1] - ws.G[0]);
    D.Sg0 = ws.Sg[0];
    D.Sg1 = ws.Sg[1];
    D.Pg0 = ws.Pg[0];
    D.Pg1 = ws.Pg[1];
    D.Pw = ws.P;
    unsigned *f = ws.flags;
    D.gen = f;
    f += (size_t)nblk * nblk;
    D.rb = f;
    f += (size_t)ng * nblk;
    D.mc = f;
    f += ng;
    D.done = f;
    f += ng;
    D.sl = f;
    f += 3 * ng;
    D.next = f;
    D.next_m = f + 1;
    D.mxcc = f + 18;
    D.arrived = f + 19;
    D.mcu = f + 20;  // 32 slots
    D.abort = f + 64;
    // bound of one dependency wait (GDCA_SWEEP_TIMEOUT_MS; a healthy wait is microseconds).  By default it grows with the work the
    // launch holds: the early workgroups of a launch that starts behind another one (two contexts in flight on one GPU, another
    // tenant) wait for most of that launch's run time, and a slow but correct run must not be turned into GDCA_EHIP -- 4 s or
    // eight times the modelled duration of everything this launch carries (n = 48 000: 1.8 s measured), whichever is longer
    const double model_ms = 8.0 * members * ((double)n_pad * n_pad * n_pad / 50e12 * 1e3);
    const long timeout_ms = tu.sweep_timeout_ms > 0 ? tu.sweep_timeout_ms : std::max(4000L, (long)model_ms);
    D.timeout_ticks = (unsigned long long)std::max(1L, timeout_ms) * 100000ull;
    // ... and once a polling wave has been off the hardware (spin_until): 50 ms or that modelled duration -- a launch that merely shares
    // the device goes on within that; one that did not get all its workgroups back never does
    D.hole_timeout_ticks = std::min(D.timeout_ticks, (unsigned long long)std::max(50L, (long)model_ms) * 100000ull);
    if (job.doomed) D.timeout_ticks = D.hole_timeout_ticks = 1ull;  // (tests: the first wait that has to poll twice gives up)
    D.debug = tu.sweep_debug;
    D.item0 = ws.item0_dev;
    D.mitem0 = ws.item0_dev + (ng + 1);
    D.gs = ws.item0_dev + 2 * (ng + 1);
    D.total = pro + (int)pos;
    D.total_m = (int)mpos;
    D.pro = pro;
    D.rem_tail = rem_tail;
    D.ppb = ppb;
    D.slab = slab ? SLAB_ITEMS : 0;
    D.ring = ring;
    // compute units for the M list (same measurement): a chain-bound inverse wants every parallel item of the chain served at
    // once -- the row-slab jobs of a level of a four-block group are 24 -- and once the update hides the chain the workers go
    // back to the tiles: for multi-block groups 16 CUs up to 63 blocks, 12 up to 68, 10 up to 74, 6 up to 90 (n = 10 000: 16.5 ms
    // with 6, 16.85 with 12, 20.8 with 4), 4 up to 120, 2 beyond (n = 20 000: 123.1 ms with 2, 123.75 with 4); between single blocks (a pivot and three times eight slab items per step) 12
    // CUs below 28 blocks, 8 above
    const int mcu_rule = g == 1 ? (nblk < 28 ? 12 : 8) : (nblk <= 63 ? 16 : (nblk <= 68 ? 12 : (nblk <= 74 ? 10 : (nblk <= 90 ? 6 : (nblk <= 120 ? 4 : 2)))));
    // a member of a merged launch: its own rule unless that would hand more than an eighth of the chip to the chains
    const int mcu_merged = tu.merge_mcus >= 1 ? tu.merge_mcus : std::min(mcu_rule, std::max(2, ws.update_cus / (8 * std::max(1, members))));
    // One chain worker per elected compute unit (its second workgroup leaves the launch at once) for groups of two and three: there a
    // pivot shared its SIMDs with a K = 256 / 384 panel or tile item of the M list on the unit's other workgroup.  Round 5, every block
    // count from 47 to 76 with alternating settings (profiles/r05_mcu_solo.log): 12 such units for groups of two (47-53 blocks: 2-12 %
    // faster than 16 units with two workers each), 16 for groups of three (54-56 blocks 3-8 %, 58-59 1 %); groups of four and the
    // fused slab items between single blocks gain nothing (and lose workers), a merged launch's members keep both.
    // (a merged launch's members: tried, no effect -- config B eight to a launch 0.444 against 0.448 ms per family)
    const bool solo = !merged && (tu.mcu_solo >= 0 ? tu.mcu_solo != 0 : (g == 2 || g == 3));
    D.mcu_solo = solo ? 1 : 0;
    const int mcu_solo_rule = g == 2 ? 12 : 16;
    D.n_mcu = merged ? std::min(mcu_merged, 16) : (tu.mcus >= 1 ? std::min(tu.mcus, solo ? 32 : 16) : (solo && tu.mcu_solo < 0 ? mcu_solo_rule : mcu_rule));
    D.n_real = n_real;
    D.rl = rl;
    D.sc = job.sc;
    D.dbg = nullptr;
    D.dbg_main = nullptr;
    D.dbg_items = nullptr;
    P.g = g;
    P.mpos = mpos;
    P.chunks = chunks;
    P.slab = slab;
    P.mit.assign(mit, mit + ng + 1);
    return P;
}

static void write_sweep_trace(const char *trace_path, const SweepPlan &P, unsigned grid, unsigned long long *dbg);

void gdca_launch_spd_inverse(hipStream_t s0, const gdca_inverse_job &job, hipEvent_t *upd_ev, int max_upd_ev, int *n_upd_launch,
                             double *upd_flops)
{
    SweepPlan P = plan_sweep(s0, job, false, 1);
    {
        SweepPrep prep{};
        prep.K = 1;
        prep.fam[0] = P.prep;
        GDCA_LAUNCH_DIRECT(k_sweep_prep, dim3(16, 1), dim3(256), 0, s0, prep);
    }
    SweepDesc &D = P.D;
    const int ng = D.ng;
    const long long mpos = P.mpos;
    // GDCA_SWEEP_TRACE=file: stamps of the M-list items of this inverse are written to `file` (debug aid; synchronises)
    const char *trace_path = job.tune->sweep_trace[0] ? job.tune->sweep_trace : nullptr;
    unsigned long long *dbg = nullptr;
    if (trace_path) {
        const size_t words = (size_t)(2 * (mpos + 1) + 8 + 1024 + 19 * ng) + 4 * ((size_t)D.total + 1);
        (void)hipMalloc(&dbg, words * sizeof(unsigned long long));
        (void)hipMemsetAsync(dbg, 0, words * sizeof(unsigned long long), s0);
    }
    D.dbg = dbg;
    D.dbg_main = dbg ? dbg + 2 * (mpos + 1) : nullptr;
    D.dbg_items = dbg ? dbg + 2 * (mpos + 1) + 8 + 1024 + 19 * (size_t)ng : nullptr;
    const unsigned grid = (unsigned)(2 * job.ws.update_cus);  // two 256-VGPR workgroups per CU: every register file full
    const bool tm = upd_ev && max_upd_ev >= 2;
    if (tm) (void)hipEventRecord(upd_ev[0], s0);
    if (P.g > 1)
        GDCA_LAUNCH_DIRECT((k_sweep<true>), dim3(grid), dim3(256), 0, s0, D);
    else
        GDCA_LAUNCH_DIRECT((k_sweep<false>), dim3(grid), dim3(256), 0, s0, D);
    if (tm) (void)hipEventRecord(upd_ev[1], s0);
    if (dbg) {
        (void)hipStreamSynchronize(s0);
        write_sweep_trace(trace_path, P, grid, dbg);
        (void)hipFree(dbg);
    }
    if (n_upd_launch) *n_upd_launch = 1;
    if (upd_flops) *upd_flops = 2.0 * T * T * KC * P.chunks;
}

// K <= SWEEP_MAX_MERGE inverses as one launch of k_sweep_merged; upd_flops[k] receives member k's issued MFMA flops
void gdca_launch_spd_inverse_merged(hipStream_t s0, const gdca_inverse_job *jobs, int K, hipEvent_t *upd_ev, int max_upd_ev,
                                    double *upd_flops)
{
    SweepBatch B{};
    B.K = K;
    SweepPrep prep{};
    prep.K = K;
    int cus = 0;
    bool multi = false;
    for (int k = 0; k < K; ++k) {
        SweepPlan P = plan_sweep(s0, jobs[k], true, K);
        B.fam[k] = P.D;
        prep.fam[k] = P.prep;
        multi = multi || P.g > 1;
        if (upd_flops) upd_flops[k] = 2.0 * T * T * KC * P.chunks;
        cus = std::max(cus, jobs[k].ws.update_cus);
    }
    GDCA_LAUNCH_DIRECT(k_sweep_prep, dim3(16, (unsigned)K), dim3(256), 0, s0, prep);
    const unsigned grid = (unsigned)(2 * cus);
    const bool tm = upd_ev && max_upd_ev >= 2;
    if (tm) (void)hipEventRecord(upd_ev[0], s0);
    if (multi)
        GDCA_LAUNCH_DIRECT(k_sweep_merged<true>, dim3(grid), dim3(256), 0, s0, B);
    else
        GDCA_LAUNCH_DIRECT(k_sweep_merged<false>, dim3(grid), dim3(256), 0, s0, B);
    if (tm) (void)hipEventRecord(upd_ev[1], s0);
}

static void write_sweep_trace(const char *trace_path, const SweepPlan &P, unsigned grid, unsigned long long *dbg)
{
    const int nblk = P.D.nblk, g = P.g, ng = P.D.ng;
    const long long mpos = P.mpos;
    const bool slab = P.slab;
    const int *mit = P.mit.data();
    {
        std::vector<unsigned long long> h((size_t)2 * mpos), hm(8 + 1024 + 19 * (size_t)ng);
        (void)hipMemcpy(h.data(), dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        (void)hipMemcpy(hm.data(), dbg + 2 * (mpos + 1), hm.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        if (FILE *fp = fopen(trace_path, "w")) {
            fprintf(fp, "# nblk %d g %d ng %d; per M-list item: group local_index start_us end_us (since the first stamp)\n", nblk, g, ng);
            unsigned long long tend = 0, tmin = ~0ull;
            for (unsigned w = 0; w < grid; ++w) {
                if (hm[8 + w] > tend) tend = hm[8 + w];
                if (hm[8 + w] && hm[8 + w] < tmin) tmin = hm[8 + w];
            }
            double idle_end = 0;
            for (unsigned w = 0; w < grid; ++w)
                if (hm[8 + w]) idle_end += (double)(tend - hm[8 + w]) / 100.0;
            fprintf(fp, "# main list: tile items %llu, %.1f us each incl. %.1f us waiting; panel items %llu, %.1f us each incl. %.1f us waiting; "
                        "wb items %llu, %.1f us each; workers idle at the end: %.1f us on average (first out %.1f us before the last)\n",
                    hm[7], hm[7] ? hm[6] / 100.0 / hm[7] : 0.0, hm[7] ? hm[0] / 100.0 / hm[7] : 0.0, hm[3],
                    hm[3] ? hm[2] / 100.0 / hm[3] : 0.0, hm[3] ? hm[1] / 100.0 / hm[3] : 0.0, hm[5], hm[5] ? hm[4] / 100.0 / hm[5] : 0.0,
                    idle_end / grid, (double)(tend - tmin) / 100.0);
            fprintf(fp, "# shader clock during tile items: %.3f GHz (s_memtime cycles / 100 MHz wall clock)\n",
                    hm[6] ? (double)hm[8 + 1023] / (double)hm[6] * 0.1 : 0.0);
            if (hm[8 + 603])
                fprintf(fp, "# pivot items: %llu; loads and first micro-pivot %.1f us, blocked sweep %.1f us, stores and publication %.1f us each\n", hm[8 + 603],
                        hm[8 + 600] / 100.0 / hm[8 + 603], hm[8 + 601] / 100.0 / hm[8 + 603], hm[8 + 602] / 100.0 / hm[8 + 603]);
            fprintf(fp, "# per group: tile items, us each, shader clock GHz:");
            for (int p = 0; p < ng; ++p) {
                const unsigned long long *gp = &hm[8 + 1024 + 3 * (size_t)p];
                if (gp[2]) fprintf(fp, " [%d] %llu %.1f %.3f", p, gp[2], gp[1] / 100.0 / gp[2], gp[1] ? (double)gp[0] / (double)gp[1] * 0.1 : 0.0);
            }
            fprintf(fp, "\n");
            unsigned long long t0 = ~0ull;
            for (size_t x = 0; x < h.size(); x += 2)
                if (h[x] && h[x] < t0) t0 = h[x];
            if (slab) {
                fprintf(fp, "# main-list items of row block b0 + 3, us since the first stamp: group: panel halves first taken .. last done | tile (b0+3, b0+1) taken .. done | "
                            "(b0+3, b0+2) | (b0+3, b0+3)\n");
                for (int p = 0; p + 3 < ng; ++p) {
                    const unsigned long long *xs = &hm[8 + 1024 + 3 * (size_t)ng + 16 * (size_t)p];
                    fprintf(fp, "# x %d:", p);
                    for (int k = 0; k < 8; k += 2)
                        fprintf(fp, " %8.1f (ready %8.1f) .. %8.1f |", xs[k == 0 ? 0 : k + 1] ? (double)(long long)(xs[k == 0 ? 0 : k + 1] - t0) / 100.0 : -1.0,
                                xs[9 + k / 2] ? (double)(long long)(xs[9 + k / 2] - t0) / 100.0 : -1.0,
                                xs[k == 0 ? 1 : k + 2] ? (double)(long long)(xs[k == 0 ? 1 : k + 2] - t0) / 100.0 : -1.0);
                    fprintf(fp, "\n");
                }
            }
            int q = 0;
            for (long long x = 0; x < mpos; ++x) {
                while (x >= mit[q + 1]) ++q;
                fprintf(fp, "%d %lld %.2f %.2f\n", q, x - mit[q], (double)(h[2 * x] - t0) / 100.0, (double)(h[2 * x + 1] - t0) / 100.0);
            }
            // every main-list item: kind p a b workgroup | taken, end of its wait (= taken where it did not have to poll), done
            std::vector<unsigned long long> hi((size_t)4 * P.D.total);
            (void)hipMemcpy(hi.data(), dbg + 2 * (mpos + 1) + 8 + 1024 + 19 * (size_t)ng, hi.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            for (int x = 0; x < P.D.total; ++x) {
                const unsigned long long w = hi[4 * (size_t)x], tk = hi[4 * (size_t)x + 1], rd = hi[4 * (size_t)x + 2], dn = hi[4 * (size_t)x + 3];
                if (!tk) continue;  // (an empty slot)
                fprintf(fp, "m %d %llu %llu %llu %llu %llu %.2f %.2f %.2f\n", x, w >> 56, (w >> 44) & 0xfff, (w >> 32) & 0xfff, (w >> 20) & 0xfff, w & 0xfffff,
                        (double)(long long)(tk - t0) / 100.0, (double)(long long)((rd ? rd : tk) - t0) / 100.0, (double)(long long)(dn - t0) / 100.0);
            }
            fclose(fp);
        }
    }
}

// =====================================================================================================================
// How good is the inverse?  A norm estimate and, for ill-conditioned matrices, one Newton-Schulz step
// =====================================================================================================================
// The block sweep is Gauss-Jordan elimination: its forward error grows like cond(C)^2 u where LAPACK's potrf + potri (the
// reference's inv(cholesky(C)), src/GaussDCA.jl:34) stays near cond(C) u (tests/test_gpu_conditioning.py, against columns refined
// in extended precision: 60 x LAPACK's error at cond 2e5, 5e4 x at cond 1e8).  At the pseudocounts gDCA is used with (0.2 .. 0.8:
// cond 1e2 .. 1e4) both are at rounding level; a tiny pseudocount is legal input, though (:50).  So every inverse gets a cheap
// measure of its conditioning -- ||X||_1 from the lower triangle it has just produced; with ||C||_1 that is kappa_1 -- and
// beyond a threshold ONE step of Newton-Schulz iteration
//       X1 = X0 + X0 (I - C X0)
// which squares the residual I - C X0 (it converges while that residual is below 1: cond up to ~1e7 for this sweep) and leaves
// the error at the cond u level of the two f64 products themselves.  Two plain tiled products on the sweep's 128 x 128 x 128 MFMA
// tile product (3 n^3 flops at ~30 TFLOP/s: several times the inverse itself -- a path for rare inputs, not a fast one).

// colsum[c] += sum_r |A(r, c)| over the FULL symmetric matrix whose lower block triangle (diagonal tiles in full) is A: tile (I, J),
// I >= J, feeds the columns of block J and, mirrored, those of block I
__global__ __launch_bounds__(256) void k_sym_colabs(const double *__restrict__ A, size_t ld, int nblk, double *__restrict__ colsum)
{
    int I = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
    while ((long long)I * (I + 1) / 2 > (long long)blockIdx.x) --I;
    while ((long long)(I + 1) * (I + 2) / 2 <= (long long)blockIdx.x) ++I;
    const int J = (int)(blockIdx.x - (long long)I * (I + 1) / 2);
    const double *At = A + (size_t)I * T + (size_t)J * T * ld;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // a wave walks 32 columns; a column is two coalesced loads (rows lane, lane + 64), its sum a butterfly; the lanes keep the
    // row sums of their two rows over the wave's columns (= column sums of the mirror tile)
    double r0 = 0.0, r1 = 0.0;
    for (int c = 32 * wv; c < 32 * wv + 32; ++c) {
        const double a0 = fabs(At[(size_t)lane + (size_t)c * ld]), a1 = fabs(At[(size_t)lane + 64 + (size_t)c * ld]);
        r0 += a0;
        r1 += a1;
        double cs = a0 + a1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cs += __shfl_xor(cs, o, 64);
        if (lane == 0) atomicAdd(&colsum[(size_t)J * T + c], cs);
    }
    if (I != J) {
        atomicAdd(&colsum[(size_t)I * T + lane], r0);
        atomicAdd(&colsum[(size_t)I * T + lane + 64], r1);
    }
}

// column sums of |.| of a plain n x n matrix (ld): the caller's C of the operator-level entry
__global__ __launch_bounds__(256) void k_colabs_full(const double *__restrict__ C, size_t ld, int n, double *__restrict__ colsum)
{
    __shared__ double red[256];
    const int c = blockIdx.x;
    double a = 0.0;
    for (int r = threadIdx.x; r < n; r += 256) a += fabs(C[(size_t)r + (size_t)c * ld]);
    red[threadIdx.x] = a;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) colsum[c] = red[0];
}

// *out = max over the n entries of colsum
__global__ __launch_bounds__(256) void k_vec_max(const double *__restrict__ v, int n, double *__restrict__ out)
{
    __shared__ double red[256];
    double a = 0.0;
    for (int k = threadIdx.x; k < n; k += 256) a = fmax(a, v[k]);
    red[threadIdx.x] = a;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + w]);
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}

void gdca_launch_inverse_norm1(hipStream_t s, const double *A, int n_pad, int n, double *colsum_ws, double *out)
{
    (void)hipMemsetAsync(colsum_ws, 0, (size_t)n_pad * sizeof(double), s);
    const int nblk = n_pad / T;
    GDCA_LAUNCH_DIRECT(k_sym_colabs, dim3((unsigned)(nblk * (nblk + 1) / 2)), dim3(256), 0, s, A, (size_t)n_pad, nblk, colsum_ws);
    GDCA_LAUNCH_DIRECT(k_vec_max, dim3(1), dim3(256), 0, s, colsum_ws, n, out);   // (the padding's unit columns stay out of it)
}

void gdca_launch_matrix_norm1(hipStream_t s, const double *C, size_t ld, int n, double *colsum_ws, double *out)
{
    GDCA_LAUNCH_DIRECT(k_colabs_full, dim3((unsigned)n), dim3(256), 0, s, C, ld, n, colsum_ws);
    GDCA_LAUNCH_DIRECT(k_vec_max, dim3(1), dim3(256), 0, s, colsum_ws, n, out);
}

// One 128 x 128 tile of  G H^T  over the whole inner dimension (nblk blocks of 128), G(r, k) = G[r + k ld], H(c, k) = H[c + k ld]
// (both operands are symmetric matrices or stored transposed by the caller), then
//   MODE 0:  Out = I - G H^T                       (all tiles: Rt = I - X0 C = (I - C X0)^T)
//   MODE 1:  Out = -(X0 + G H^T)                   (tiles I >= J: the new -X1 = -(X0 + X0 (I - C X0)) in the sweep's storage)
//   resid (MODE 0): max |I - G H^T| over the matrix, as the bit pattern of a non-negative double (atomicMax): the step converges only
//   while the residual is below one -- the caller reports a refinement that could not have worked instead of returning it silently
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_ns_gemm(const double *__restrict__ G, const double *__restrict__ H, size_t ld, int nblk,
                                                    double *__restrict__ Out, const double *__restrict__ X0, double *__restrict__ resid)
{
    __shared__ __attribute__((aligned(16))) double GHs[2][KC][LDS_LD];
    int I, J;
    if (MODE == 0) {
        I = (int)(blockIdx.x / (unsigned)nblk);
        J = (int)(blockIdx.x % (unsigned)nblk);
    } else {
        I = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
        while ((long long)I * (I + 1) / 2 > (long long)blockIdx.x) --I;
        while ((long long)(I + 1) * (I + 2) / 2 <= (long long)blockIdx.x) ++I;
        J = (int)(blockIdx.x - (long long)I * (I + 1) / 2);
    }
    const int tid = opaque_tid(), lane = tid & 63, wv = tid >> 6;
    const int wr = wv & 1, wc = wv >> 1, l15 = lane & 15, lq = lane >> 4;
    double4_t acc[4][4];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = (double4_t){0.0, 0.0, 0.0, 0.0};
    const double *g = G + (size_t)I * T, *h = H + (size_t)J * T;
#pragma unroll 1
    for (int K = 0; K < nblk; ++K)
        tile_product<false, 4>(acc, g + (size_t)K * T * ld, ld, h + (size_t)K * T * ld, ld, GHs[0], GHs[1], nullptr, 0);
    double *Ot = Out + (size_t)I * T + (size_t)J * T * ld;
    const double *Xt = MODE == 1 ? X0 + (size_t)I * T + (size_t)J * T * ld : nullptr;
    double worst = 0.0;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = wr * 64 + tn * 16 + l15;
                const int c = wc * 64 + tm * 16 + lq + 4 * reg;
                const size_t e = (size_t)r + (size_t)c * ld;
                if (MODE == 0) {
                    const double v = ((I == J && r == c) ? 1.0 : 0.0) - acc[tm][tn][reg];
                    Ot[e] = v;
                    worst = fmax(worst, v == v ? fabs(v) : __builtin_huge_val());   // (a NaN counts as "diverged")
                } else {
                    Ot[e] = -(Xt[e] + acc[tm][tn][reg]);
                }
            }
    if (MODE == 0 && resid) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) worst = fmax(worst, __shfl_xor(worst, o, 64));
        if (lane == 0) atomicMax(reinterpret_cast<unsigned long long *>(resid), (unsigned long long)__double_as_longlong(worst));
    }
}

// A: the sweep's result (-X0 in the lower block triangle, ld = n_pad) on entry, -X1 there on exit.  C2: the matrix that was
// inverted, full symmetric, padded with the identity.  B0, Rt: n_pad x n_pad workspaces.  *resid (optional) receives max |I - X0 C|.
void gdca_launch_newton_schulz(hipStream_t s, double *A, const double *C2, double *B0, double *Rt, int n_pad, double *resid)
{
    const int nblk = n_pad / T;
    if (resid) (void)hipMemsetAsync(resid, 0, sizeof(double), s);
    gdca_launch_copy_out_neg_sym(s, A, n_pad, B0, n_pad);                                      // B0 = X0, full symmetric
    GDCA_LAUNCH_DIRECT(k_ns_gemm<0>, dim3((unsigned)(nblk * nblk)), dim3(256), 0, s, B0, C2, (size_t)n_pad, nblk, Rt, nullptr, resid);
    GDCA_LAUNCH_DIRECT(k_ns_gemm<1>, dim3((unsigned)(nblk * (nblk + 1) / 2)), dim3(256), 0, s, B0, Rt, (size_t)n_pad, nblk, A, B0, nullptr);
}

// =====================================================================================================================
// The stable way round for the inputs the sweep cannot handle: blocked Cholesky + triangular inverse + U U^T
// =====================================================================================================================
// Beyond cond ~1e10 the sweep's Schur complements go non-positive through rounding (it reports "not positive definite" where
// LAPACK's dpotrf still factors the matrix) or its Newton-Schulz step cannot converge.  For exactly those runs -- never on the
// ordinary path -- the inverse is computed again the way the reference does it (src/GaussDCA.jl:34: dpotrf + dpotri), as plain
// multi-launch blocked algorithms on the 128 x 128 x 128 MFMA tile product:
//     1. right-looking Cholesky, block column k:  diagonal tile (one workgroup, in LDS; also W_kk = L_kk^-1),
//        panel  L_Ik = A_Ik W_kk^T,  trailing update  A_IJ -= L_Ik L_Jk^T                                    [n^3 / 3]
//     2. U = L^-T by block forward substitution, block row i:  T_ij = sum_{k=j}^{i-1} L_ik U_jk^T,  U_ji = -T_ij^T W_ii^T  [n^3 / 3]
//     3. X = U U^T on the lower block triangle, written as -X into the sweep's storage                         [n^3 / 3]
// ~5 nblk launches, a few TFLOP/s: tens of milliseconds where the sweep needs ten -- a fallback, not a fast path.  Its `info` is
// dpotrf's by construction (the first non-positive pivot of the Cholesky factorisation).

// acc(r, c) += sum over `count` blocks of 128:  G(r, k) H(c, k),  G / H advancing by gstep / hstep doubles per block
template <bool GT>
__device__ __forceinline__ void chol_tile_acc(double4_t (&acc)[4][4], const double *g, size_t gld, size_t gstep, const double *h, size_t hld,
                                              size_t hstep, int count, double (*Gs)[LDS_LD], double (*Hs)[LDS_LD])
{
#pragma unroll 1
    for (int k = 0; k < count; ++k) tile_product<GT, 4>(acc, g + (size_t)k * gstep, gld, h + (size_t)k * hstep, hld, Gs, Hs, nullptr, 0);
}

#define CHOL_ACC_ZERO(acc)                                   \
    _Pragma("unroll") for (int tm_ = 0; tm_ < 4; ++tm_)      \
        _Pragma("unroll") for (int tn_ = 0; tn_ < 4; ++tn_) acc[tm_][tn_] = (double4_t){0.0, 0.0, 0.0, 0.0};

// out(r, c) = f(old, acc) for the workgroup's tile: MODE 0: acc;  1: old - acc;  2: -acc
template <int MODE>
__device__ __forceinline__ void chol_tile_store(const double4_t (&acc)[4][4], double *Ot, size_t ld)
{
    const int tid = opaque_tid(), lane = tid & 63, wv = tid >> 6;
    const int wr = wv & 1, wc = wv >> 1, l15 = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const size_t e = (size_t)(wr * 64 + tn * 16 + l15) + (size_t)(wc * 64 + tm * 16 + lq + 4 * reg) * ld;
                Ot[e] = MODE == 0 ? acc[tm][tn][reg] : (MODE == 1 ? Ot[e] - acc[tm][tn][reg] : -acc[tm][tn][reg]);
            }
}

// diagonal tile k: Cholesky in LDS (lower L written back, upper part zeroed), W = L^-1 to Wd (128 x 128, ld 128), U_kk = W^T (full tile)
__global__ __launch_bounds__(256) void k_chol_diag(double *__restrict__ Lm, size_t ld, int k, double *__restrict__ Wd, double *__restrict__ U,
                                                   int n_real, gdca_dev_scalars *sc)
{
    extern __shared__ __attribute__((aligned(16))) double a[];  // [c][r], 128 x 128
    __shared__ int bad;
    const int tid = threadIdx.x;
    double *At = Lm + (size_t)k * T + (size_t)k * T * ld;
    for (int e = tid; e < T * T; e += 256) a[e] = At[(size_t)(e & 127) + (size_t)(e >> 7) * ld];
    if (tid == 0) bad = 0;
    for (int j = 0; j < T; ++j) {
        __syncthreads();
        double d = a[j + j * T];
        if (!(d > 0.0)) {
            if (tid == 0 && bad == 0) bad = j + 1;
            d = 1.0;  // keep going with finite numbers: the result is discarded
        }
        const double sq = sqrt(d);
        __syncthreads();
        if (tid >= j && tid < T) a[tid + j * T] = (tid == j) ? sq : a[tid + j * T] / sq;
        __syncthreads();
        for (int e = tid; e < T * T; e += 256) {
            const int r = e & 127, c = e >> 7;
            if (c > j && r >= c) a[e] -= a[r + j * T] * a[c + j * T];
        }
    }
    __syncthreads();
    if (tid == 0 && bad != 0) {
        const int idx = k * T + bad;
        if (idx <= n_real && sc->info == 0) sc->info = idx;   // (tiles are factored one after the other: the first report is the smallest index)
    }
    for (int e = tid; e < T * T; e += 256) {
        const int r = e & 127, c = e >> 7;
        At[(size_t)r + (size_t)c * ld] = r >= c ? a[e] : 0.0;
    }
    // W = L^-1, column c by thread c (forward substitution; the thread reads back its own earlier entries)
    double *W = Wd;
    if (tid < T) {
        const int c = tid;
        for (int r = 0; r < c; ++r) W[r + c * T] = 0.0;
        W[c + c * T] = 1.0 / a[c + c * T];
        for (int r = c + 1; r < T; ++r) {
            double sacc = 0.0;
            for (int m = c; m < r; ++m) sacc += a[r + m * T] * W[m + c * T];
            W[r + c * T] = -sacc / a[r + r * T];
        }
    }
    __threadfence_block();
    __syncthreads();
    double *Ut = U + (size_t)k * T + (size_t)k * T * ld;
    for (int e = tid; e < T * T; e += 256) {
        const int r = e & 127, c = e >> 7;        // U_kk(r, c) = W(c, r) for c >= r, else 0
        Ut[(size_t)r + (size_t)c * ld] = c >= r ? W[c + r * T] : 0.0;
    }
}

// panel of block column k: tile (I, k), I = k + 1 + blockIdx.x  <-  tile W_kk^T   (in place: a workgroup reads only the tile it writes)
__global__ __launch_bounds__(256, 2) void k_chol_panel(double *__restrict__ Lm, size_t ld, int k, const double *__restrict__ Wd)
{
    __shared__ __attribute__((aligned(16))) double GHs[2][KC][LDS_LD];
    const int I = k + 1 + (int)blockIdx.x;
    double4_t acc[4][4];
    CHOL_ACC_ZERO(acc);
    double *At = Lm + (size_t)I * T + (size_t)k * T * ld;
    chol_tile_acc<false>(acc, At, ld, 0, Wd, (size_t)T, 0, 1, GHs[0], GHs[1]);
    __syncthreads();
    chol_tile_store<0>(acc, At, ld);
}

// trailing update after block column k: tile (I, J), I >= J > k  -=  L_Ik L_Jk^T
__global__ __launch_bounds__(256, 2) void k_chol_trail(double *__restrict__ Lm, size_t ld, int k)
{
    __shared__ __attribute__((aligned(16))) double GHs[2][KC][LDS_LD];
    int a_ = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
    while ((long long)a_ * (a_ + 1) / 2 > (long long)blockIdx.x) --a_;
    while ((long long)(a_ + 1) * (a_ + 2) / 2 <= (long long)blockIdx.x) ++a_;
    const int b_ = (int)(blockIdx.x - (long long)a_ * (a_ + 1) / 2);
    const int I = k + 1 + a_, J = k + 1 + b_;
    double4_t acc[4][4];
    CHOL_ACC_ZERO(acc);
    chol_tile_acc<false>(acc, Lm + (size_t)I * T + (size_t)k * T * ld, ld, 0, Lm + (size_t)J * T + (size_t)k * T * ld, ld, 0, 1, GHs[0], GHs[1]);
    chol_tile_store<1>(acc, Lm + (size_t)I * T + (size_t)J * T * ld, ld);
}

// forward substitution, block row i, first half: T_ij = sum_{k=j}^{i-1} L_ik U_jk^T, j = blockIdx.x < i   (Tm: tile (i, j))
__global__ __launch_bounds__(256, 2) void k_chol_fwd_t(const double *__restrict__ Lm, const double *__restrict__ U, double *__restrict__ Tm, size_t ld, int i)
{
    __shared__ __attribute__((aligned(16))) double GHs[2][KC][LDS_LD];
    const int j = (int)blockIdx.x;
    double4_t acc[4][4];
    CHOL_ACC_ZERO(acc);
    chol_tile_acc<false>(acc, Lm + (size_t)i * T + (size_t)j * T * ld, ld, (size_t)T * ld, U + (size_t)j * T + (size_t)j * T * ld, ld, (size_t)T * ld, i - j,
                         GHs[0], GHs[1]);
    chol_tile_store<0>(acc, Tm + (size_t)i * T + (size_t)j * T * ld, ld);
}

// ... second half: U_ji = -T_ij^T W_ii^T
__global__ __launch_bounds__(256, 2) void k_chol_fwd_u(const double *__restrict__ Tm, const double *__restrict__ Wd, double *__restrict__ U, size_t ld, int i)
{
    __shared__ __attribute__((aligned(16))) double GHs[2][KC][LDS_LD];
    const int j = (int)blockIdx.x;
    double4_t acc[4][4];
    CHOL_ACC_ZERO(acc);
    // out(c, r) = sum_m T_ij(m, c) W_ii(r, m):  G(c, m) = T_ij[m + c ld] (transposed access), H(r, m) = W_ii[r + m 128]
    chol_tile_acc<true>(acc, Tm + (size_t)i * T + (size_t)j * T * ld, ld, 0, Wd, (size_t)T, 0, 1, GHs[0], GHs[1]);
    chol_tile_store<2>(acc, U + (size_t)j * T + (size_t)i * T * ld, ld);
}

// X = U U^T on the lower block triangle: tile (I, J), I >= J:  sum_{k >= I} U_Ik U_Jk^T, written as -X (the sweep's storage)
__global__ __launch_bounds__(256, 2) void k_chol_uut(const double *__restrict__ U, double *__restrict__ Aout, size_t ld, int nblk)
{
    __shared__ __attribute__((aligned(16))) double GHs[2][KC][LDS_LD];
    int I = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
    while ((long long)I * (I + 1) / 2 > (long long)blockIdx.x) --I;
    while ((long long)(I + 1) * (I + 2) / 2 <= (long long)blockIdx.x) ++I;
    const int J = (int)(blockIdx.x - (long long)I * (I + 1) / 2);
    double4_t acc[4][4];
    CHOL_ACC_ZERO(acc);
    chol_tile_acc<false>(acc, U + (size_t)I * T + (size_t)I * T * ld, ld, (size_t)T * ld, U + (size_t)J * T + (size_t)I * T * ld, ld, (size_t)T * ld, nblk - I,
                         GHs[0], GHs[1]);
    chol_tile_store<2>(acc, Aout + (size_t)I * T + (size_t)J * T * ld, ld);
}

// C2: the matrix (full symmetric or lower block triangle, n_pad x n_pad, identity padding) -- overwritten by its Cholesky factor;
// U, Tm: n_pad x n_pad workspaces; Wd: 128 x 128; Aout receives -inverse in its lower block triangle.  sc->info: dpotrf's index.
void gdca_launch_cholesky_inverse(hipStream_t s, double *C2, double *U, double *Tm, double *Wd, double *Aout, int n_pad, int n_real,
                                  gdca_dev_scalars *sc)
{
    const int nblk = n_pad / T;
    const size_t ld = (size_t)n_pad;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_chol_diag), hipFuncAttributeMaxDynamicSharedMemorySize, T * T * (int)sizeof(double));
    for (int k = 0; k < nblk; ++k) {
        GDCA_LAUNCH_DIRECT(k_chol_diag, dim3(1), dim3(256), T * T * sizeof(double), s, C2, ld, k, Wd + (size_t)k * T * T, U, n_real, sc);
        const int below = nblk - k - 1;
        if (below > 0) {
            GDCA_LAUNCH_DIRECT(k_chol_panel, dim3((unsigned)below), dim3(256), 0, s, C2, ld, k, Wd + (size_t)k * T * T);
            GDCA_LAUNCH_DIRECT(k_chol_trail, dim3((unsigned)(below * (below + 1) / 2)), dim3(256), 0, s, C2, ld, k);
        }
    }
    for (int i = 1; i < nblk; ++i) {
        GDCA_LAUNCH_DIRECT(k_chol_fwd_t, dim3((unsigned)i), dim3(256), 0, s, C2, U, Tm, ld, i);
        GDCA_LAUNCH_DIRECT(k_chol_fwd_u, dim3((unsigned)i), dim3(256), 0, s, Tm, Wd + (size_t)i * T * T, U, ld, i);
    }
    GDCA_LAUNCH_DIRECT(k_chol_uut, dim3((unsigned)(nblk * (nblk + 1) / 2)), dim3(256), 0, s, U, Aout, ld, nblk);
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
    GDCA_LAUNCH_DIRECT(k_probe_mfma_f64, dim3(blocks), dim3(256), 0, s, out, iters);
}
