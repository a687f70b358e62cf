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

typedef unsigned long long u64;

// ---- single-site sums -----------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_pi_tally(const int8_t *__restrict__ Z, const u64 *__restrict__ Wfix,
                                                   u64 *__restrict__ Pifix, int N, int M, int seq_per_block)
{
    __shared__ u64 h[32][128];
    const int t = threadIdx.x;
    const int i = blockIdx.x * 128 + t;
#pragma unroll
    for (int z = 0; z < 32; ++z) h[z][t] = 0;
    const int kbeg = blockIdx.y * seq_per_block;
    const int kend = min(M, kbeg + seq_per_block);
    if (i < N) {
        const int8_t *p = Z + (size_t)kbeg * N + i;
        for (int k = kbeg; k < kend; ++k) {
            h[p[0] & 31][t] += Wfix[k];
            p += N;
        }
#pragma unroll
        for (int z = 0; z < 32; ++z) {
            const u64 v = h[z][t];
            if (v) atomicAdd(&Pifix[(size_t)i * 32 + z], v);
        }
    }
}

void gdca_launch_pi_tally(hipStream_t s, const int8_t *Z, const u64 *Wfix, u64 *Pifix, int N, int M)
{
    const int cb = (N + 127) / 128;
    int chunks = (1024 + cb - 1) / cb;
    int spb = (M + chunks - 1) / chunks;
    if (spb < 64) spb = 64;
    chunks = (M + spb - 1) / spb;
    hipLaunchKernelGGL(k_pi_tally, dim3(cb, chunks), dim3(128), 0, s, Z, Wfix, Pifix, N, M, spb);
}

__global__ __launch_bounds__(256) void k_pi_finalize(const u64 *__restrict__ Pifix, int N, int q, int fix_shift,
                                                      const double *__restrict__ Meff_dev, double pc,
                                                      double *__restrict__ Pi_true, double *__restrict__ Pi_pc)
{
    const int s = q - 1;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= N * s) return;
    const int i = e / s, a = e % s;  // state a+1
    const double Meff = *Meff_dev;
    const double pt = ldexp((double)Pifix[(size_t)i * 32 + a + 1], -fix_shift) / Meff;
    if (Pi_true) Pi_true[e] = pt;
    if (Pi_pc) {
        const double pcq = pc / (double)q;
        Pi_pc[e] = (1.0 - pc) * pt + pcq;
    }
}

void gdca_launch_pi_finalize(hipStream_t s, const u64 *Pifix, int N, int q, int fix_shift, const double *Meff_dev,
                             double pc, double *Pi_true, double *Pi_pc)
{
    const int n = N * (q - 1);
    hipLaunchKernelGGL(k_pi_finalize, dim3((n + 255) / 256), dim3(256), 0, s, Pifix, N, q, fix_shift, Meff_dev, pc,
                       Pi_true, Pi_pc);
}

// ---- pair tallies -----------------------------------------------------------------------------------
#define TALLY_THREADS 512
#define TALLY_CHUNK 512  // sequences staged per pass (one per thread)

template <int TJ>
__global__ __launch_bounds__(TALLY_THREADS) void k_pair_tally(
    const int8_t *__restrict__ Z, const int8_t *__restrict__ Zt, const u64 *__restrict__ Wfix, int N, int M, int q,
    int fix_shift, const double *__restrict__ Meff_dev, double pc, const double *__restrict__ Pi_pc, int mode,
    double *__restrict__ out, size_t ld)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int s = q - 1;
    const int i = blockIdx.y;
    const int j0 = blockIdx.x * TJ;
    if (j0 + TJ - 1 < i) return;  // block entirely left of the diagonal: its mirror does the work

    u64 *hist = reinterpret_cast<u64 *>(smem);                       // [s*s][TJ]
    u64 *w_s = hist + (size_t)s * s * TJ;                            // [TALLY_CHUNK]
    int8_t *a_s = reinterpret_cast<int8_t *>(w_s + TALLY_CHUNK);     // [TALLY_CHUNK]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int SPI = 64 / TJ;                 // sequences per wave-instruction
    constexpr int NWAVE = TALLY_THREADS / 64;
    constexpr int SEQ_PER_WAVE = TALLY_CHUNK / NWAVE;  // 64
    const int jl = lane % TJ, sub = lane / TJ;
    const int j = j0 + jl;
    const bool jok = (j < N) && (j >= i);

    for (int e = tid; e < s * s * TJ; e += TALLY_THREADS) hist[e] = 0;

    for (int kc = 0; kc < M; kc += TALLY_CHUNK) {
        __syncthreads();
        {
            const int k = kc + tid;
            a_s[tid] = (k < M) ? Zt[(size_t)i * M + k] : (int8_t)0;
            w_s[tid] = (k < M) ? Wfix[k] : 0ull;
        }
        __syncthreads();
        const int kw = wv * SEQ_PER_WAVE;
#pragma unroll 2
        for (int it = 0; it < SEQ_PER_WAVE / SPI; it += 4) {
            int bsym[4], asym[4];
            u64 wv4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = kw + (it + u) * SPI + sub;
                const int k = kc + kk;
                asym[u] = a_s[kk];
                wv4[u] = w_s[kk];
                bsym[u] = (jok && k < M) ? (int)Z[(size_t)k * N + j] : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int a = asym[u], b = bsym[u];
                if (a >= 1 && a <= s && b >= 1 && b <= s)
                    atomicAdd(&hist[(size_t)((a - 1) * s + (b - 1)) * TJ + jl], wv4[u]);
            }
        }
    }
    __syncthreads();

    // ---- epilogue: histograms -> Pij_true (mode 0) or covariance C (mode 1) ----
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
        const double pt = ldexp((double)hist[(size_t)(a * s + b) * TJ + l], -fix_shift) / Meff;
        double v = pt;
        if (mode == 1) {
            const double pij = (jj != i) ? ((1.0 - pc) * pt + off_add) : ((1.0 - pc) * pt + ((a == b) ? pcq : 0.0));
            v = pij - Pi_pc[i * s + a] * Pi_pc[jj * s + b];
        }
        out[(size_t)(jj * s + b) + (size_t)(i * s + a) * ld] = v;
    }
    // pass 2: the mirror element (row i*s+a, col j*s+b): contiguous over a for fixed (j, b)
    for (int e = tid; e < total; e += TALLY_THREADS) {
        const int l = e / (s * s), rem = e - l * (s * s);
        const int b = rem / s, a = rem - b * s;
        const int jj = j0 + l;
        if (jj >= N || jj <= i) continue;  // the diagonal block was written in full by pass 1
        const double pt = ldexp((double)hist[(size_t)(a * s + b) * TJ + l], -fix_shift) / Meff;
        double v = pt;
        if (mode == 1) {
            const double pij = (1.0 - pc) * pt + off_add;
            v = pij - Pi_pc[i * s + a] * Pi_pc[jj * s + b];
        }
        out[(size_t)(i * s + a) + (size_t)(jj * s + b) * ld] = v;
    }
}

void gdca_launch_pair_tally(hipStream_t st, const int8_t *Z, const int8_t *Zt, const u64 *Wfix, int N, int M, int q,
                            int fix_shift, const double *Meff_dev, double pc, const double *Pi_pc, int mode,
                            double *out, size_t ld)
{
    const int s = q - 1;
    const size_t aux = (size_t)TALLY_CHUNK * 8 + TALLY_CHUNK;
    if ((size_t)s * s * 32 * 8 + aux <= 160 * 1024) {
        const size_t lds = (size_t)s * s * 32 * 8 + aux;
        static bool attr32 = false;
        if (!attr32) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_pair_tally<32>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr32 = true;
        }
        hipLaunchKernelGGL(k_pair_tally<32>, dim3((N + 31) / 32, N), dim3(TALLY_THREADS), lds, st, Z, Zt, Wfix, N, M,
                           q, fix_shift, Meff_dev, pc, Pi_pc, mode, out, ld);
    } else {
        const size_t lds = (size_t)s * s * 16 * 8 + aux;
        static bool attr16 = false;
        if (!attr16) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_pair_tally<16>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr16 = true;
        }
        hipLaunchKernelGGL(k_pair_tally<16>, dim3((N + 15) / 16, N), dim3(TALLY_THREADS), lds, st, Z, Zt, Wfix, N, M,
                           q, fix_shift, Meff_dev, pc, Pi_pc, mode, out, ld);
    }
}
