// Weighted one-/two-site frequency accumulation and the fused covariance build
// (DCAUtils compute_weighted_frequencies' accumulation, add_pseudocount, and the in-tree
// compute_C; reference call sites src/GaussDCA.jl:28, :30, :32/:76).
//
//   Pij[(i,a),(j,b)] = (1/Meff) sum_k W_k [Z[i,k]==a][Z[j,k]==b]          a,b in 1..s, s = q-1
//
// MI355X design.  This is a sparse tally (N(N+1)/2 * M weighted increments), not a GEMM: the
// dense one-hot product would be 400x more work.  A workgroup owns column i against a block
// of TJ = 32 columns j and keeps the 32 s x s histograms in LDS (100 KB at s = 20), laid out
// [a][b][j] so that a wave-instruction's 64 lanes (32 columns x 2 sequences) always touch 32
// different bank pairs whatever the symbols are.  Z is read in its natural layout: for one
// sequence the 32 bytes of the column block are contiguous.
//
// Accumulation is in 64-bit FIXED POINT (ds_add_u64): W_k is scaled by 2^shift with
// M * 2^shift <= 2^63, so the integer sums cannot overflow, are independent of the order in
// which waves reach the LDS atomics (bit-reproducible run to run), and resolve 2^-shift
// (7e-15 at M = 50k) -- finer than an f64 running sum of magnitude ~Meff/20.
//
// The epilogue applies 1/Meff, the pseudocount rule (diagonal blocks get pc/q on their
// diagonal only, off-diagonal blocks pc/q^2 everywhere) and subtracts Pi'Pi'^T, so the n x n
// covariance is written to HBM exactly once (8 n^2 bytes) and Pij never exists in memory.
#include "gdca_internal.h"
#include "gdca_launch.h"
#include <cstdlib>

typedef unsigned long long u64;

// ---- single-site sums -----------------------------------------------------------------------------
struct k_pi_tally_args {
    const int8_t *Z;
    const u64 *Wfix;
    u64 *Pifix;
    int N;
    int M;
    int seq_per_block;
    int q;
    gdca_dev_scalars *sc;
};
static inline k_pi_tally_args k_pi_tally_mk(const int8_t *Z, const u64 *Wfix, u64 *Pifix, int N, int M, int seq_per_block, int q, gdca_dev_scalars *sc)
{
    return k_pi_tally_args{Z, Wfix, Pifix, N, M, seq_per_block, q, sc};
}
template <int CAP>
__global__ __launch_bounds__(128) void k_pi_tally(const BatchArgs<k_pi_tally_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const int8_t *__restrict__ Z = a_.Z;
    const u64 *__restrict__ Wfix = a_.Wfix;
    u64 *__restrict__ Pifix = a_.Pifix;
    int N = a_.N;
    int M = a_.M;
    int seq_per_block = a_.seq_per_block;
    int q = a_.q;
    gdca_dev_scalars *sc = a_.sc;
    __shared__ u64 h[32][128];
    unsigned bad = 0;  // any byte outside 1..q
    const int t = threadIdx.x;
    const int i = blockIdx.x * 128 + t;
#pragma unroll
    for (int z = 0; z < 32; ++z) h[z][t] = 0;
    const int kbeg = blockIdx.y * seq_per_block;
    const int kend = min(M, kbeg + seq_per_block);
    if (i < N) {
        const int8_t *p = Z + (size_t)kbeg * N + i;
        int k = kbeg;
        for (; k + 16 <= kend; k += 16) {  // 16 strided byte loads in flight before the dependent LDS adds
            int z[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const unsigned raw = (uint8_t)p[(size_t)u * N];
                bad |= (raw - 1u) >= (unsigned)q;
                z[u] = raw & 31;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) h[z[u]][t] += Wfix[k + u];
            p += (size_t)16 * N;
        }
        for (; k < kend; ++k) {
            const unsigned raw = (uint8_t)p[0];
            bad |= (raw - 1u) >= (unsigned)q;
            h[raw & 31][t] += Wfix[k];
            p += N;
        }
        if (bad) atomicOr(&sc->bad_symbol, 1);
#pragma unroll
        for (int z = 0; z < 32; ++z) {
            const u64 v = h[z][t];
            if (v) atomicAdd(&Pifix[(size_t)i * 32 + z], v);
        }
    }
}

void gdca_launch_pi_tally(hipStream_t s, const int8_t *Z, const u64 *Wfix, u64 *Pifix, int N, int M, int q,
                          gdca_dev_scalars *sc)
{
    const int cb = (N + 127) / 128;
    int chunks = (512 + cb - 1) / cb;  // every chunk ends in one global atomic per counter
    int spb = (M + chunks - 1) / chunks;
    if (spb < 64) spb = 64;
    chunks = (M + spb - 1) / spb;
    (gdca_launch<k_pi_tally_args, k_pi_tally<1>, k_pi_tally<GDCA_MAXB>>(dim3(cb, chunks), dim3(128), 0, s, k_pi_tally_mk(Z, Wfix, Pifix, N, M, spb, q, sc)));
}

struct k_pi_finalize_args {
    const u64 *Pifix;
    int N;
    int q;
    int fix_shift;
    const double *Meff_dev;
    double pc;
    double *Pi_true;
    double *Pi_pc;
    double *pi_max;
};
static inline k_pi_finalize_args k_pi_finalize_mk(const u64 *Pifix, int N, int q, int fix_shift, const double *Meff_dev, double pc, double *Pi_true, double *Pi_pc, double *pi_max)
{
    return k_pi_finalize_args{Pifix, N, q, fix_shift, Meff_dev, pc, Pi_true, Pi_pc, pi_max};
}
template <int CAP>
__global__ __launch_bounds__(256) void k_pi_finalize(const BatchArgs<k_pi_finalize_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const u64 *__restrict__ Pifix = a_.Pifix;
    int N = a_.N;
    int q = a_.q;
    int fix_shift = a_.fix_shift;
    const double *__restrict__ Meff_dev = a_.Meff_dev;
    double pc = a_.pc;
    double *__restrict__ Pi_true = a_.Pi_true;
    double *__restrict__ Pi_pc = a_.Pi_pc;
    double *__restrict__ pi_max = a_.pi_max;
    const int s = q - 1;
    const int e = blockIdx.x * 256 + threadIdx.x;
    double mine = 0.0;
    if (e < N * s) {
        const int i = e / s, a = e % s;  // state a+1
        const double Meff = *Meff_dev;
        const double pt = ldexp((double)Pifix[(size_t)i * 32 + a + 1], -fix_shift) / Meff;
        if (Pi_true) Pi_true[e] = pt;
        mine = (1.0 - pc) * pt + pc / (double)q;
        if (Pi_pc) Pi_pc[e] = mine;
    }
    if (pi_max) {  // (uniform) the largest frequency with pseudocount: ||C||_1 <= 2 N pi_max, what k_cov_norm1 decides on
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mine = fmax(mine, __shfl_xor(mine, o, 64));
        // non-negative doubles order like their bit patterns (*pi_max was zeroed with the run's scalars)
        if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<u64 *>(pi_max), (u64)__double_as_longlong(mine));
    }
}

// ||C||_1 of the covariance the pair tally has just written (full symmetric, ld) -- but only where it is NEEDED: the refinement
// screen of the fused path (gdca_api.hip, cond_bound) is  cond_2(C) <= ||C||_1 q^2 / pc,  and  ||C||_1 <= 2 N pi_max  is at hand for
// nothing (a column of C sums, in absolute value, to at most  sum_j sum_b [Pij(jb, ia) + Pi(jb) Pi(ia)] <= 2 N Pi(ia)).  At the
// pseudocounts gDCA is used with that cheap bound already settles the question and every workgroup leaves at once; where it does
// not (pc = 0.2 with a conserved column; small pc), the columns are summed -- one pass over C before the sweep overwrites it.
struct k_cov_norm1_args {
    const double *C;
    size_t ld;
    int n;
    int N;
    int q;
    double pc;
    double cond_limit;
    gdca_dev_scalars *sc;
    int always;
};
static inline k_cov_norm1_args k_cov_norm1_mk(const double *C, size_t ld, int n, int N, int q, double pc, double cond_limit, gdca_dev_scalars *sc, int always)
{
    return k_cov_norm1_args{C, ld, n, N, q, pc, cond_limit, sc, always};
}
template <int CAP>
__global__ __launch_bounds__(256) void k_cov_norm1(const BatchArgs<k_cov_norm1_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const double *__restrict__ C = a_.C;
    size_t ld = a_.ld;
    int n = a_.n;
    int N = a_.N;
    int q = a_.q;
    double pc = a_.pc;
    double cond_limit = a_.cond_limit;
    gdca_dev_scalars *__restrict__ sc = a_.sc;
    int always = a_.always;
    const double cheap = pc > 0.0 ? 2.0 * (double)N * sc->pi_max * (double)q * (double)q / pc : HUGE_VAL;
    if (!always && cheap <= cond_limit) return;
    __shared__ double red[256];
    double best = 0.0;
    for (int c = blockIdx.x; c < n; c += gridDim.x) {
        double a = 0.0;
        for (int r = threadIdx.x; r < n; r += 256) a += fabs(C[(size_t)r + (size_t)c * ld]);
        red[threadIdx.x] = a;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
            __syncthreads();
        }
        best = fmax(best, red[0]);
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<u64 *>(&sc->mat_norm1), (u64)__double_as_longlong(best));
}

void gdca_launch_cov_norm1(hipStream_t s, const double *C, size_t ld, int N, int q, double pc, double cond_limit, gdca_dev_scalars *sc, int always)
{
    (gdca_launch<k_cov_norm1_args, k_cov_norm1<1>, k_cov_norm1<GDCA_MAXB>>(dim3(512), dim3(256), 0, s, k_cov_norm1_mk(C, ld, N * (q - 1), N, q, pc, cond_limit, sc, always)));
}

void gdca_launch_pi_finalize(hipStream_t s, const u64 *Pifix, int N, int q, int fix_shift, const double *Meff_dev,
                             double pc, double *Pi_true, double *Pi_pc, double *pi_max)
{
    const int n = N * (q - 1);
    (gdca_launch<k_pi_finalize_args, k_pi_finalize<1>, k_pi_finalize<GDCA_MAXB>>(dim3((n + 255) / 256), dim3(256), 0, s, k_pi_finalize_mk(Pifix, N, q, fix_shift, Meff_dev, pc, Pi_true, Pi_pc, pi_max)));
}

// ---- pair tallies -----------------------------------------------------------------------------------
#define TALLY_THREADS 1024
#define TALLY_CHUNK 1024  // sequences staged per pass (one per thread)

// Z [M][N] -> Zc [ceil(N/TJ)][M][TJ] (zero padded): for one column block the TJ bytes of
// consecutive sequences are consecutive in memory, so a workgroup's staging loads are fully
// coalesced 16-byte accesses with no over-fetch.  Tile transpose through LDS.
struct k_colblock_args {
    const int8_t *Z;
    int8_t *Zc;
    int N;
    int M;
    int TJ;
};
static inline k_colblock_args k_colblock_mk(const int8_t *Z, int8_t *Zc, int N, int M, int TJ)
{
    return k_colblock_args{Z, Zc, N, M, TJ};
}
template <int CAP>
__global__ __launch_bounds__(256) void k_colblock(const BatchArgs<k_colblock_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const int8_t *__restrict__ Z = a_.Z;
    int8_t *__restrict__ Zc = a_.Zc;
    int N = a_.N;
    int M = a_.M;
    int TJ = a_.TJ;
    __shared__ int8_t tile[64][64 + 4];
    const int k0 = blockIdx.y * 64, c0 = blockIdx.x * 64;  // 64 sequences x 64 columns
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = k0 + r * 4 + ty, c = c0 + tx;
        tile[r * 4 + ty][tx] = (k < M && c < N) ? Z[(size_t)k * N + c] : (int8_t)0;
    }
    __syncthreads();
    // write: for each column block inside this 64-column strip, rows of TJ bytes per sequence
    const int nb = 64 / TJ;  // column blocks in the strip (2 or 4)
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int blk = e / (64 * TJ), rem = e % (64 * TJ);
        const int kl = rem / TJ, cl = rem % TJ;
        const int k = k0 + kl;
        const int cb = c0 / TJ + blk;
        if (blk < nb && k < M && cb * TJ < ((N + TJ - 1) / TJ) * TJ)
            Zc[((size_t)cb * M + k) * TJ + cl] = tile[kl][blk * TJ + cl];
    }
}

void gdca_launch_colblock(hipStream_t s, const int8_t *Z, int8_t *Zc, int N, int M, int TJ)
{
    dim3 grid((N + 63) / 64, (M + 63) / 64);
    (gdca_launch<k_colblock_args, k_colblock<1>, k_colblock<GDCA_MAXB>>(grid, dim3(256), 0, s, k_colblock_mk(Z, Zc, N, M, TJ)));
}

// Workgroup = (column i) x (block of TJ columns j >= i's block), 1024 threads = 16 waves (the
// histograms take most of the LDS, so one workgroup per CU: the waves have to come from here).
// Per pass of 1024 sequences the block's TJ bytes of every sequence and a packed
// {row(Z[i,k]), Wfix[k]} word are staged in LDS (the next pass's global loads are in flight while
// the current one is tallied), then each wave walks 64 sequences, 64/TJ at a time: lane =
// (sequence, column j): one ds_read_u8 for Z[j,k], one broadcast ds_read_b64 for the packed word,
// one ds_add_u64 into hist[a][b][j].
//
// The histogram has s+2 columns per row (b = 0 and b = q are junk columns for padding / gaps) so
// the inner loop needs no validity test at all: an invalid Z[i,k] is staged as weight 0, lanes left
// of the diagonal tally into columns the epilogue never reads.
struct k_pair_tally_args {
    const int8_t *Zc;
    const int8_t *Zt;
    const u64 *Wfix;
    int N;
    int M;
    int q;
    int fix_shift;
    const double *Meff_dev;
    double pc;
    const double *Pi_pc;
    int mode;
    double *out;
    size_t ld;
};
static inline k_pair_tally_args k_pair_tally_mk(const int8_t *Zc, const int8_t *Zt, const u64 *Wfix, int N, int M, int q, int fix_shift, const double *Meff_dev, double pc, const double *Pi_pc, int mode, double *out, size_t ld)
{
    return k_pair_tally_args{Zc, Zt, Wfix, N, M, q, fix_shift, Meff_dev, pc, Pi_pc, mode, out, ld};
}
template <int CAP, int TJ>
__global__ __launch_bounds__(TALLY_THREADS) void k_pair_tally(const BatchArgs<k_pair_tally_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const int8_t *__restrict__ Zc = a_.Zc;
    const int8_t *__restrict__ Zt = a_.Zt;
    const u64 *__restrict__ Wfix = a_.Wfix;
    int N = a_.N;
    int M = a_.M;
    int q = a_.q;
    int fix_shift = a_.fix_shift;
    const double *__restrict__ Meff_dev = a_.Meff_dev;
    double pc = a_.pc;
    const double *__restrict__ Pi_pc = a_.Pi_pc;
    int mode = a_.mode;
    double *__restrict__ out = a_.out;
    size_t ld = a_.ld;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int s = q - 1;
    const int i = blockIdx.y;
    // Skewed column-block index: workgroups go to the 8 XCDs round-robin in linear block order, and with the
    // plain mapping XCD x would get the column blocks == x (mod 8), whose share of non-empty (j >= i) blocks
    // differs by 2x between the first and the last.  The skew spreads the early-exit blocks evenly
    // (measured at N=500, M=50k: 5.9 -> 3.7 ms).
    const int jblk = (blockIdx.x + blockIdx.y) % gridDim.x;
    const int j0 = jblk * TJ;
    if (j0 + TJ - 1 < i) return;  // block entirely left of the diagonal: its mirror does the work

    const int RS = (s + 2) * TJ;                                                  // u64 per histogram row a
    u64 *hist = reinterpret_cast<u64 *>(smem);                                    // [s][s+2][TJ]
    u64 *meta_s = hist + (size_t)s * RS;                                          // [TALLY_CHUNK]
    uint8_t *zs = reinterpret_cast<uint8_t *>(meta_s + TALLY_CHUNK);              // [TALLY_CHUNK][TJ]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int SPI = 64 / TJ;  // sequences per wave-instruction
    constexpr int NWAVE = TALLY_THREADS / 64;
    constexpr int SEQ_PER_WAVE = TALLY_CHUNK / NWAVE;  // 64
    constexpr int NV = TJ / 16;                        // dwordx4 loads per sequence
    static_assert(TALLY_CHUNK == TALLY_THREADS, "one staged sequence per thread");
    const int jl = lane % TJ, sub = lane / TJ;
    const u64 wmask = (1ull << 59) - 1;
    const unsigned qclamp = (unsigned)(s + 1);

    for (int e = tid; e < s * RS; e += TALLY_THREADS) hist[e] = 0;

    // prefetch registers for one pass: this thread's sequence kc + tid.  The raw loads stay
    // untouched in registers until the LDS write of the NEXT pass, so a whole pass of tallies
    // covers their latency.  Plain scalars: an indexed uint4 array here goes to scratch.
    const int8_t *Zblk = Zc + (size_t)jblk * M * TJ;
    uint4 z0 = make_uint4(0, 0, 0, 0), z1 = z0;
    u64 wf = 0;
    int8_t av = 0;
#define TALLY_FETCH(KC)                                                                       \
    do {                                                                                      \
        int k_ = (KC) + tid;                                                                  \
        k_ = k_ < M ? k_ : M - 1; /* clamp: the tail is staged with weight 0 */               \
        const uint4 *src_ = reinterpret_cast<const uint4 *>(Zblk + (size_t)k_ * TJ);          \
        z0 = src_[0];                                                                         \
        if (NV > 1) z1 = src_[1];                                                             \
        av = Zt[(size_t)i * M + k_];                                                          \
        wf = Wfix[k_];                                                                        \
    } while (0)
    TALLY_FETCH(0);
    for (int kc = 0; kc < M; kc += TALLY_CHUNK) {
        __syncthreads();  // previous pass fully consumed (and hist zeroed, first time)
        {
            reinterpret_cast<uint4 *>(zs + (size_t)tid * TJ)[0] = z0;
            if (NV > 1) reinterpret_cast<uint4 *>(zs + (size_t)tid * TJ)[1] = z1;
            const unsigned a = (unsigned)(uint8_t)av - 1u;  // row index 0..s-1 when valid
            const bool valid = (a < (unsigned)s) && (kc + tid < M);
            meta_s[tid] = valid ? (((u64)a << 59) | (wf & wmask)) : 0ull;
        }
        __syncthreads();
        if (kc + TALLY_CHUNK < M) TALLY_FETCH(kc + TALLY_CHUNK);
        const int kw = wv * SEQ_PER_WAVE;
#pragma unroll 2
        for (int it = 0; it < SEQ_PER_WAVE / SPI; it += 4) {
            u64 m4[4];
            unsigned b4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = kw + (it + u) * SPI + sub;
                m4[u] = meta_s[kk];
                b4[u] = zs[kk * TJ + jl];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned a = (unsigned)(m4[u] >> 59);
                const unsigned b = min(b4[u], qclamp);
                const unsigned idx = __umul24(a, (unsigned)RS) + __umul24(b, (unsigned)TJ) + (unsigned)jl;
                atomicAdd(&hist[idx], m4[u] & wmask);
            }
        }
    }
    __syncthreads();

    // ---- epilogue: histograms -> Pij_true (mode 0) or covariance C (mode 1) ----
    // Pi' of column i and of the TJ columns of the block are staged in LDS first (the staging
    // area is free now): the per-element math then touches LDS and registers only.
    double *pi_i = reinterpret_cast<double *>(meta_s);  // [s]
    double *pi_j = pi_i + 32;                            // [TJ * s]
    if (mode == 1) {
        for (int e = tid; e < s; e += TALLY_THREADS) pi_i[e] = Pi_pc[i * s + e];
        for (int e = tid; e < TJ * s; e += TALLY_THREADS) {
            const int jj = j0 + e / s;
            pi_j[e] = (jj < N) ? Pi_pc[jj * s + (e % s)] : 0.0;
        }
    }
    __syncthreads();
    const double Meff = *Meff_dev;
    const double pcq = pc / (double)q;
    const double off_add = pcq / (double)q;
    const int total = s * s * TJ;
    // pass 1: element (row j*s+b, col i*s+a): contiguous over (j, b) for fixed a
    for (int e = tid; e < total; e += TALLY_THREADS) {
        const int a = e / (TJ * s), rem = e - a * (TJ * s);
        const int l = rem / s, b = rem - l * s;
        const int jj = j0 + l;
        if (jj >= N || jj < i) continue;
        const double pt = ldexp((double)hist[(size_t)a * RS + (b + 1) * TJ + l], -fix_shift) / Meff;
        double v = pt;
        if (mode == 1) {
            const double pij = (jj != i) ? ((1.0 - pc) * pt + off_add) : ((1.0 - pc) * pt + ((a == b) ? pcq : 0.0));
            v = pij - pi_i[a] * pi_j[rem];
        }
        out[(size_t)(jj * s + b) + (size_t)(i * s + a) * ld] = v;
    }
    // pass 2: the mirror element (row i*s+a, col j*s+b): contiguous over a for fixed (j, b)
    for (int e = tid; e < total; e += TALLY_THREADS) {
        const int l = e / (s * s), rem = e - l * (s * s);
        const int b = rem / s, a = rem - b * s;
        const int jj = j0 + l;
        if (jj >= N || jj <= i) continue;  // the diagonal block was written in full by pass 1
        const double pt = ldexp((double)hist[(size_t)a * RS + (b + 1) * TJ + l], -fix_shift) / Meff;
        double v = pt;
        if (mode == 1) {
            const double pij = (1.0 - pc) * pt + off_add;
            v = pij - pi_i[a] * pi_j[l * s + b];
        }
        out[(size_t)(i * s + a) + (size_t)(jj * s + b) * ld] = v;
    }
}

static size_t tally_lds_bytes(int s, int TJ)
{
    return (size_t)s * (s + 2) * TJ * 8 + (size_t)TALLY_CHUNK * 8 + (size_t)TALLY_CHUNK * TJ;
}

int gdca_tally_tj(int q, int tj_wanted)
{
    // 16 columns per workgroup: 80 KB of histograms at s = 20, so TWO 1024-thread workgroups share a CU and one's staging
    // barriers and epilogue hide behind the other's atomics (measured at config C: 3.26 ms against 4.03 ms with 32 columns
    // and one workgroup per CU; the context option GDCA_TALLY_TJ=32 brings the wide form back for comparison)
    if (tj_wanted == 32 && tally_lds_bytes(q - 1, 32) <= 160 * 1024) return 32;
    return 16;
}

void gdca_launch_pair_tally(hipStream_t st, const int8_t *Zc, const int8_t *Zt, const u64 *Wfix, int N, int M, int q,
                            int fix_shift, const double *Meff_dev, double pc, const double *Pi_pc, int mode,
                            double *out, size_t ld, int TJ)
{
    const int s = q - 1;
    if (TJ == 32) {
        (gdca_launch<k_pair_tally_args, k_pair_tally<1, 32>, k_pair_tally<GDCA_MAXB, 32>>(dim3((N + 31) / 32, N), dim3(TALLY_THREADS), tally_lds_bytes(s, 32), st, k_pair_tally_mk(Zc, Zt, Wfix, N, M, q, fix_shift, Meff_dev, pc, Pi_pc, mode, out, ld)));
    } else {
        (gdca_launch<k_pair_tally_args, k_pair_tally<1, 16>, k_pair_tally<GDCA_MAXB, 16>>(dim3((N + 15) / 16, N), dim3(TALLY_THREADS), tally_lds_bytes(s, 16), st, k_pair_tally_mk(Zc, Zt, Wfix, N, M, q, fix_shift, Meff_dev, pc, Pi_pc, mode, out, ld)));
    }
}
