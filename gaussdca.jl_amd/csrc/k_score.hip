// Per-(i,j) scoring on the s x s blocks of the inverse covariance, and APC
// (DCAUtils compute_FN / compute_DI_gauss, reference call sites src/GaussDCA.jl:39 and :37;
// correct_APC, src/GaussDCA.jl:78-86).
//
// Input is the lower triangle of A = -mJ as the sweep leaves it (ld = n_pad).  For sites i < j
// the kernels read the block at rows j*s.., columns i*s.. = -(mJ block (i,j))^T.  Both scores
// are invariant under that transpose and sign: FN is a Frobenius norm; DI depends on the block
// only through the singular values of L_j^T X L_i.
//
// FN: one wave per site pair, one pass over the block (HBM-bound: 8 s^2 bytes per pair,
// 8 n (n - s) / 2 bytes in all).  DI: one wave per pair, everything in LDS: two triangular
// products, V = MM MM^T, then a cyclic Jacobi eigenvalue iteration in round-robin ordering
// (s/2 independent rotations per round) -- latency/VALU-bound, no HBM traffic to speak of.
#include "gdca_internal.h"

__device__ __forceinline__ void pair_decode(long long p, int &i, int &j)
{
    // p = j (j - 1) / 2 + i,  0 <= i < j
    int jj = (int)((1.0 + sqrt(1.0 + 8.0 * (double)p)) * 0.5);
    while ((long long)jj * (jj - 1) / 2 > p) --jj;
    while ((long long)(jj + 1) * jj / 2 <= p) ++jj;
    j = jj;
    i = (int)(p - (long long)jj * (jj - 1) / 2);
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- FN ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fn(const double *__restrict__ A, size_t ld, int N, int sdim, long long npairs,
                                             double *__restrict__ S)
{
    extern __shared__ __attribute__((aligned(16))) double fsm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ss = sdim * sdim;
    double *blk = fsm + (size_t)wv * (ss + 64);
    double *rm = blk + ss, *cm = rm + 32;
    long long p = (long long)blockIdx.x * 4 + wv;
    const bool live = p < npairs;
    if (!live) p = npairs - 1;
    int i, j;
    pair_decode(p, i, j);
    const double *src = A + (size_t)j * sdim + (size_t)i * sdim * ld;
    for (int e = lane; e < ss; e += 64) {
        const int r = e % sdim, c = e / sdim;
        blk[e] = src[(size_t)r + (size_t)c * ld];
    }
    __syncthreads();
    if (lane < sdim) {
        double a = 0.0;
        for (int c = 0; c < sdim; ++c) a += blk[lane + c * sdim];
        rm[lane] = a / (double)sdim;
    } else if (lane >= 32 && lane - 32 < sdim) {
        const int c = lane - 32;
        double a = 0.0;
        for (int r = 0; r < sdim; ++r) a += blk[r + c * sdim];
        cm[c] = a / (double)sdim;
    }
    __syncthreads();
    double tot = 0.0;
    for (int r = 0; r < sdim; ++r) tot += rm[r];
    tot /= (double)sdim;  // = sum(block) / s^2
    double f = 0.0;
    for (int e = lane; e < ss; e += 64) {
        const int r = e % sdim, c = e / sdim;
        const double kx = blk[e] - rm[r] - cm[c] + tot;
        f += kx * kx;
    }
    f = wave_sum(f);
    if (live && lane == 0) {
        const double v = sqrt(f);
        S[(size_t)i + (size_t)j * N] = v;
        S[(size_t)j + (size_t)i * N] = v;
    }
}

void gdca_launch_fn(hipStream_t s, const double *A, size_t ld, int N, int sdim, double *S)
{
    (void)hipMemsetAsync(S, 0, (size_t)N * N * sizeof(double), s);
    const long long npairs = (long long)N * (N - 1) / 2;
    if (npairs <= 0) return;
    const size_t lds = (size_t)4 * (sdim * sdim + 64) * sizeof(double);
    hipLaunchKernelGGL(k_fn, dim3((unsigned)((npairs + 3) / 4)), dim3(256), lds, s, A, ld, N, sdim, npairs, S);
}

// ---- Cholesky factors of the diagonal blocks of C ------------------------------------------------------
__global__ __launch_bounds__(64) void k_diag_chol(const double *__restrict__ D, int sdim, double *__restrict__ Ld)
{
    __shared__ double m[32 * 32];
    const int i = blockIdx.x, t = threadIdx.x;
    const int ss = sdim * sdim;
    for (int e = t; e < ss; e += 64) m[e] = D[(size_t)i * ss + e];  // column-major r + c*s
    __syncthreads();
    for (int j = 0; j < sdim; ++j) {
        const double d = sqrt(m[j + j * sdim]);
        __syncthreads();
        if (t >= j && t < sdim) m[t + j * sdim] = (t == j) ? d : m[t + j * sdim] / d;
        __syncthreads();
        // trailing update: column c > j handled by thread c
        if (t > j && t < sdim) {
            const double lcj = m[t + j * sdim];
            for (int r = t; r < sdim; ++r) m[r + t * sdim] -= m[r + j * sdim] * lcj;
        }
        __syncthreads();
    }
    for (int e = t; e < ss; e += 64) {
        const int r = e % sdim, c = e / sdim;
        Ld[(size_t)i * ss + e] = (r >= c) ? m[e] : 0.0;
    }
}

void gdca_launch_diag_chol(hipStream_t s, const double *D, int N, int sdim, double *Ld)
{
    hipLaunchKernelGGL(k_diag_chol, dim3(N), dim3(64), 0, s, D, sdim, Ld);
}

// ---- DI ---------------------------------------------------------------------------------------------
#define DI_MAX_SWEEPS 14

__global__ __launch_bounds__(64) void k_di(const double *__restrict__ A, size_t ld, const double *__restrict__ Ld,
                                            int N, int sdim, double *__restrict__ S)
{
    extern __shared__ __attribute__((aligned(16))) double dsm[];
    const int lane = threadIdx.x;
    const int ss = sdim * sdim;
    double *B0 = dsm, *B1 = dsm + ss, *B2 = dsm + 2 * ss, *B3 = dsm + 3 * ss;
    double *alpha = dsm + 4 * ss;      // [32] rotation cos per index
    double *beta = alpha + 32;         // [32] signed sin per index
    int *partner = reinterpret_cast<int *>(beta + 32);  // [32]

    int i, j;
    pair_decode((long long)blockIdx.x, i, j);
    const double *src = A + (size_t)j * sdim + (size_t)i * sdim * ld;
    for (int e = lane; e < ss; e += 64) {
        const int r = e % sdim, c = e / sdim;
        B0[e] = src[(size_t)r + (size_t)c * ld];        // X (r, c), column-major
        B1[e] = Ld[(size_t)i * ss + e];                 // L_i
        B2[e] = Ld[(size_t)j * ss + e];                 // L_j
    }
    __syncthreads();
    // T1 = X L_i :  T1(r,c) = sum_{m >= c} X(r,m) L_i(m,c)
    for (int e = lane; e < ss; e += 64) {
        const int r = e % sdim, c = e / sdim;
        double a = 0.0;
        for (int m = c; m < sdim; ++m) a += B0[r + m * sdim] * B1[m + c * sdim];
        B3[e] = a;
    }
    __syncthreads();
    // MM = L_j^T T1 :  MM(r,c) = sum_{m >= r} L_j(m,r) T1(m,c)
    for (int e = lane; e < ss; e += 64) {
        const int r = e % sdim, c = e / sdim;
        double a = 0.0;
        for (int m = r; m < sdim; ++m) a += B2[m + r * sdim] * B3[m + c * sdim];
        B0[e] = a;
    }
    __syncthreads();
    // V = MM MM^T (symmetric): V(r,c) = sum_m MM(r,m) MM(c,m)
    for (int e = lane; e < ss; e += 64) {
        const int r = e % sdim, c = e / sdim;
        const int rr = r >= c ? r : c, cc = r >= c ? c : r;  // compute from the ordered pair: bitwise symmetric
        double a = 0.0;
        for (int m = 0; m < sdim; ++m) a += B0[rr + m * sdim] * B0[cc + m * sdim];
        B1[e] = a;
    }
    __syncthreads();

    // cyclic Jacobi, round-robin ordering over ne = even(s) players (a dummy player if s is odd)
    double *V = B1, *Vn = B2;
    const int ne = (sdim + 1) & ~1, np = ne / 2;
    double frob2 = 0.0;
    for (int e = lane; e < ss; e += 64) frob2 += V[e] * V[e];
    frob2 = wave_sum(frob2);
    for (int sweep = 0; sweep < DI_MAX_SWEEPS; ++sweep) {
        double off2 = 0.0;
        for (int e = lane; e < ss; e += 64) {
            const int r = e % sdim, c = e / sdim;
            if (r != c) off2 += V[e] * V[e];
        }
        off2 = wave_sum(off2);
        if (off2 <= 1e-30 * frob2) break;
        for (int round = 0; round < ne - 1; ++round) {
            if (lane < np) {
                int pp, qq;
                if (lane == 0) {
                    pp = ne - 1;
                    qq = round;
                } else {
                    pp = (round + lane) % (ne - 1);
                    qq = (round - lane + (ne - 1)) % (ne - 1);
                }
                if (pp > qq) {
                    const int t = pp;
                    pp = qq;
                    qq = t;
                }
                double c = 1.0, sn = 0.0;
                if (qq < sdim) {
                    const double app = V[pp + pp * sdim], aqq = V[qq + qq * sdim], apq = V[pp + qq * sdim];
                    if (apq != 0.0) {
                        const double tau = (aqq - app) / (2.0 * apq);
                        const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        c = 1.0 / sqrt(1.0 + t * t);
                        sn = t * c;
                    }
                    alpha[pp] = c;
                    beta[pp] = -sn;
                    partner[pp] = qq;
                    alpha[qq] = c;
                    beta[qq] = sn;
                    partner[qq] = pp;
                } else if (pp < sdim) {  // paired with the dummy player: identity
                    alpha[pp] = 1.0;
                    beta[pp] = 0.0;
                    partner[pp] = pp;
                }
            }
            __syncthreads();
            for (int e = lane; e < ss; e += 64) {
                const int r = e % sdim, c = e / sdim;
                const int rp = partner[r], cp = partner[c];
                const double ar = alpha[r], br = beta[r], ac = alpha[c], bc = beta[c];
                const double v = ar * (ac * V[r + c * sdim] + bc * V[r + cp * sdim]) +
                                 br * (ac * V[rp + c * sdim] + bc * V[rp + cp * sdim]);
                Vn[e] = v;
            }
            __syncthreads();
            double *tmp = V;
            V = Vn;
            Vn = tmp;
        }
    }
    // DI = z + 0.5 sum_k log(1 + sqrt(1 + 4 gamma_k)),  z = 0.5 s log 0.5
    double acc = 0.0;
    if (lane < sdim) {
        double g = V[lane + lane * sdim];
        g = g > 0.0 ? g : 0.0;
        acc = log(1.0 + sqrt(1.0 + 4.0 * g));
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        const double z = 0.5 * (double)sdim * log(0.5);
        const double v = z + 0.5 * acc;
        S[(size_t)i + (size_t)j * N] = v;
        S[(size_t)j + (size_t)i * N] = v;
    }
}

void gdca_launch_di(hipStream_t s, const double *A, size_t ld, const double *Ld, int N, int sdim, double *S)
{
    (void)hipMemsetAsync(S, 0, (size_t)N * N * sizeof(double), s);
    const long long npairs = (long long)N * (N - 1) / 2;
    if (npairs <= 0) return;
    const size_t lds = (size_t)(4 * sdim * sdim + 64) * sizeof(double) + 32 * sizeof(int);
    hipLaunchKernelGGL(k_di, dim3((unsigned)npairs), dim3(64), lds, s, A, ld, Ld, N, sdim, S);
}

// ---- APC --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_colsum(const double *__restrict__ S, int N, double *__restrict__ cs)
{
    __shared__ double red[256];
    const int c = blockIdx.x;
    double a = 0.0;
    for (int r = threadIdx.x; r < N; r += 256) a += S[(size_t)r + (size_t)c * N];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) cs[c] = red[0];
}

// S <- S - (Sj * Si) / Sa,  Sa = sum(S) (1 - 1/N)   (src/GaussDCA.jl:78-86; S symmetric: Si = Sj^T)
__global__ __launch_bounds__(256) void k_apc_apply(double *__restrict__ S, int N, const double *__restrict__ cs)
{
    __shared__ double red[256];
    double a = 0.0;
    for (int r = threadIdx.x; r < N; r += 256) a += cs[r];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    const double Sa = red[0] * (1.0 - 1.0 / (double)N);
    const int c = blockIdx.x;
    const double sc = cs[c];
    for (int r = threadIdx.x; r < N; r += 256) {
        const size_t e = (size_t)r + (size_t)c * N;
        S[e] = S[e] - (cs[r] * sc) / Sa;
    }
}

void gdca_launch_apc(hipStream_t s, double *S, int N, double *colsum_ws)
{
    hipLaunchKernelGGL(k_colsum, dim3(N), dim3(256), 0, s, S, N, colsum_ws);
    hipLaunchKernelGGL(k_apc_apply, dim3(N), dim3(256), 0, s, S, N, colsum_ws);
}
