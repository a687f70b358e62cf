// Column histograms and theta = :auto  (compute_theta inside DCAUtils'
// compute_weighted_frequencies; reference call site src/GaussDCA.jl:28, doc README.md:78-80).
//
// The reference makes an all-pairs pass to get the mean pair identity.  The same integer
//   sum_{k<l} #{i : Z[i,k] == Z[i,l]}  =  sum_i sum_a c_ia (c_ia - 1) / 2
// follows from the per-column symbol counts c_ia, so this stage is one O(N*M) streaming
// pass (HBM-bound, N*M bytes) instead of M^2 N / 2 compares.  All integer, exact.
#include "gdca_internal.h"
#include "gdca_launch.h"

// ---- Z [M][N] -> Zt [N][M] (byte transpose through LDS, 64 x 64 tiles) ---------------------
struct k_transpose_i8_args {
    const int8_t *Z;
    int8_t *Zt;
    int N;
    int M;
};
static inline k_transpose_i8_args k_transpose_i8_mk(const int8_t *Z, int8_t *Zt, int N, int M)
{
    return k_transpose_i8_args{Z, Zt, N, M};
}
template <int CAP>
__global__ __launch_bounds__(256) void k_transpose_i8(const BatchArgs<k_transpose_i8_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const int8_t *__restrict__ Z = a_.Z;
    int8_t *__restrict__ Zt = a_.Zt;
    int N = a_.N;
    int M = a_.M;
    __shared__ int8_t tile[64][65];
    const int k0 = blockIdx.y * 64, i0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = k0 + r * 4 + ty, i = i0 + tx;
        tile[r * 4 + ty][tx] = (k < M && i < N) ? Z[(size_t)k * N + i] : (int8_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = i0 + r * 4 + ty, k = k0 + tx;
        if (i < N && k < M) Zt[(size_t)i * M + k] = tile[tx][r * 4 + ty];
    }
}

void gdca_launch_transpose_i8(hipStream_t s, const int8_t *Z, int8_t *Zt, int N, int M)
{
    dim3 grid((N + 63) / 64, (M + 63) / 64);
    (gdca_launch<k_transpose_i8_args, k_transpose_i8<1>, k_transpose_i8<GDCA_MAXB>>(grid, dim3(256), 0, s, k_transpose_i8_mk(Z, Zt, N, M)));
}

// ---- per-column symbol counts ---------------------------------------------------------------
// One thread owns one alignment column; its 32 counters sit in LDS as h[z][thread] so that a
// wave's 64 lanes always hit 64 different banks whatever the symbols are.  No atomics inside
// the workgroup (each counter has one owner); one global integer atomic per non-zero counter.
struct k_column_hist_args {
    const int8_t *Z;
    uint32_t *cnt;
    int N;
    int M;
    int seq_per_block;
};
static inline k_column_hist_args k_column_hist_mk(const int8_t *Z, uint32_t *cnt, int N, int M, int seq_per_block)
{
    return k_column_hist_args{Z, cnt, N, M, seq_per_block};
}
template <int CAP>
__global__ __launch_bounds__(256) void k_column_hist(const BatchArgs<k_column_hist_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const int8_t *__restrict__ Z = a_.Z;
    uint32_t *__restrict__ cnt = a_.cnt;
    int N = a_.N;
    int M = a_.M;
    int seq_per_block = a_.seq_per_block;
    __shared__ uint32_t h[32][256];
    const int t = threadIdx.x;
    const int i = blockIdx.x * 256 + t;
#pragma unroll
    for (int z = 0; z < 32; ++z) h[z][t] = 0;
    const int kbeg = blockIdx.y * seq_per_block;
    const int kend = min(M, kbeg + seq_per_block);
    if (i < N) {
        const int8_t *p = Z + (size_t)kbeg * N + i;
        int k = kbeg;
        // 16 strided byte loads in flight per thread before the dependent LDS increments: the loop is bound by
        // memory latency, not by bandwidth (25 MB in total)
        for (; k + 16 <= kend; k += 16) {
            int z[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) z[u] = p[(size_t)u * N] & 31;
#pragma unroll
            for (int u = 0; u < 16; ++u) h[z[u]][t] += 1;
            p += (size_t)16 * N;
        }
        for (; k < kend; ++k) {
            h[p[0] & 31][t] += 1;
            p += N;
        }
#pragma unroll
        for (int z = 0; z < 32; ++z) {
            const uint32_t v = h[z][t];
            if (v) atomicAdd(&cnt[(size_t)i * 32 + z], v);
        }
    }
}

void gdca_launch_column_hist(hipStream_t s, const int8_t *Z, uint32_t *cnt, int N, int M)
{
    const int cb = (N + 255) / 256;
    int chunks = (256 + cb - 1) / cb;  // ~256 workgroups: every chunk ends in one global atomic per counter
    int spb = (M + chunks - 1) / chunks;
    if (spb < 64) spb = 64;
    chunks = (M + spb - 1) / spb;
    (gdca_launch<k_column_hist_args, k_column_hist<1>, k_column_hist<GDCA_MAXB>>(dim3(cb, chunks), dim3(256), 0, s, k_column_hist_mk(Z, cnt, N, M, spb)));
}

// ---- theta, threshold -------------------------------------------------------------------------
struct k_theta_finalize_args {
    const uint32_t *cnt;
    int N;
    int M;
    double theta_in;
    gdca_dev_scalars *sc;
};
static inline k_theta_finalize_args k_theta_finalize_mk(const uint32_t *cnt, int N, int M, double theta_in, gdca_dev_scalars *sc)
{
    return k_theta_finalize_args{cnt, N, M, theta_in, sc};
}
template <int CAP>
__global__ __launch_bounds__(256) void k_theta_finalize(const BatchArgs<k_theta_finalize_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const uint32_t *__restrict__ cnt = a_.cnt;
    int N = a_.N;
    int M = a_.M;
    double theta_in = a_.theta_in;
    gdca_dev_scalars *sc = a_.sc;
    __shared__ unsigned long long red[256];
    double theta = theta_in;
    if (theta_in < 0.0) {
        unsigned long long acc = 0;
        for (int e = threadIdx.x; e < N * 32; e += 256) {
            const unsigned long long c = cnt[e];
            acc += c * (c - 1) / 2;  // c == 0 -> 0 * (2^64-1) / 2 == 0
        }
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const unsigned long long tot = red[0];
            sc->pair_sum = tot;
            if (M < 2) {
                theta = 0.0;
            } else {
                // same operation order as the oracle: tot / (N * (0.5 * M * (M - 1)))
                const double phi = (double)tot / ((double)N * (0.5 * (double)M * (double)(M - 1)));
                const double c = 0.38 * 0.32;
                const double t = c / phi;
                theta = t < 0.5 ? t : 0.5;
            }
        }
    } else if (threadIdx.x == 0) {
        sc->pair_sum = 0;
    }
    if (threadIdx.x == 0) {
        sc->theta = theta;
        sc->thresh = (int)floor(theta * (double)N);
    }
}

void gdca_launch_theta_finalize(hipStream_t s, const uint32_t *cnt, int N, int M, double theta_in,
                                gdca_dev_scalars *sc)
{
    (gdca_launch<k_theta_finalize_args, k_theta_finalize<1>, k_theta_finalize<GDCA_MAXB>>(dim3(1), dim3(256), 0, s, k_theta_finalize_mk(cnt, N, M, theta_in, sc)));
}

struct k_set_thresh_args {
    gdca_dev_scalars *sc;
    int thresh;
};
static inline k_set_thresh_args k_set_thresh_mk(gdca_dev_scalars *sc, int thresh)
{
    return k_set_thresh_args{sc, thresh};
}
template <int CAP>
__global__ void k_set_thresh(const BatchArgs<k_set_thresh_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    gdca_dev_scalars *sc = a_.sc;
    int thresh = a_.thresh;
    sc->thresh = thresh;
}

void gdca_launch_set_thresh(hipStream_t s, gdca_dev_scalars *sc, int thresh)
{
    (gdca_launch<k_set_thresh_args, k_set_thresh<1>, k_set_thresh<GDCA_MAXB>>(dim3(1), dim3(1), 0, s, k_set_thresh_mk(sc, thresh)));
}
