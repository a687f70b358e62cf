/* CPU oracle accelerators -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C (OpenMP) forms of the loops in oracle/gdca_oracle.py, used so the oracle
 * finishes in seconds at parity-test sizes and as the multi-threaded "port" CPU baseline
 * in bench.py.  Same arithmetic, same accumulation order per output cell as the numpy
 * forms (tests/test_oracle_golden.py checks one against the other).
 *
 * Each function stands behind a DCAUtils.jl call made by the reference (the package is
 * not vendored in /root/reference; see the header of gdca_oracle.py):
 *   orc_pair_identity_sum  -> compute_theta's all-pairs pass   (src/GaussDCA.jl:28)
 *   orc_neighbour_counts   -> compute_weights                  (src/GaussDCA.jl:28)
 *   orc_frequencies        -> weighted Pi / Pij accumulation   (src/GaussDCA.jl:28)
 *   orc_fn                 -> compute_FN                       (src/GaussDCA.jl:39)
 *
 * Z is int8, M sequences of N contiguous bytes (Julia's N x M column-major matrix).
 * Built by __graft_entry__.build():  gcc -O3 -fopenmp -shared -fPIC
 */
#include <math.h>
#include <stdint.h>
#ifdef __AVX2__
#include <immintrin.h>
#endif
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

static inline int hamming_bytes(const int8_t *a, const int8_t *b, int N)
{
    int d = 0, i = 0;
#ifdef __AVX2__
    /* 32 symbol compares per instruction: the byte-wise counterpart of the 5-bit packed XOR/popcount
     * DCAUtils uses on UInt64 words */
    while (i + 32 <= N) {
        __m256i acc = _mm256_setzero_si256();
        int blocks = (N - i) / 32;
        if (blocks > 255) blocks = 255;
        for (int t = 0; t < blocks; ++t, i += 32) {
            const __m256i va = _mm256_loadu_si256((const __m256i *)(a + i));
            const __m256i vb = _mm256_loadu_si256((const __m256i *)(b + i));
            acc = _mm256_sub_epi8(acc, _mm256_cmpeq_epi8(va, vb)); /* +1 per equal byte */
        }
        const __m256i sad = _mm256_sad_epu8(acc, _mm256_setzero_si256());
        const int eq = (int)(_mm256_extract_epi64(sad, 0) + _mm256_extract_epi64(sad, 1) + _mm256_extract_epi64(sad, 2) +
                             _mm256_extract_epi64(sad, 3));
        d += blocks * 32 - eq;
    }
#endif
    for (; i < N; ++i) d += (a[i] != b[i]);
    return d;
}

/* sum_{k<l} #{i: Z[k][i] == Z[l][i]} -- the literal all-pairs pass */
uint64_t orc_pair_identity_sum(const int8_t *Z, int N, int M)
{
    uint64_t tot = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : tot)
    for (int k = 0; k < M - 1; ++k) {
        const int8_t *a = Z + (size_t)k * N;
        uint64_t t = 0;
        for (int l = k + 1; l < M; ++l) t += (uint64_t)(N - hamming_bytes(a, Z + (size_t)l * N, N));
        tot += t;
    }
    return tot;
}

/* n[k] (pre-set to 1 by the caller) += #{l != k : Hamming(k,l) < thresh}
 * Cache-blocked: a block of KB sequences k meets blocks of LB sequences l (both resident in L1/L2 while they are
 * compared), instead of streaming the whole alignment past every k (which is DRAM-bound on a many-core host).  Integer
 * counts: the result does not depend on the blocking or on the thread count. */
void orc_neighbour_counts(const int8_t *Z, int N, int M, int thresh, int32_t *n)
{
    const int KB = 32, LB = 256;
    const int nkb = (M + KB - 1) / KB;
#pragma omp parallel
    {
        int32_t *loc = (int32_t *)calloc((size_t)M, sizeof(int32_t));
#pragma omp for schedule(dynamic, 1)
        for (int kb = 0; kb < nkb; ++kb) {
            const int k0 = kb * KB, k1 = k0 + KB < M ? k0 + KB : M;
            for (int l0 = k0; l0 < M; l0 += LB) {
                const int l1 = l0 + LB < M ? l0 + LB : M;
                for (int k = k0; k < k1; ++k) {
                    const int8_t *a = Z + (size_t)k * N;
                    int32_t mine = 0;
                    for (int l = (l0 > k + 1 ? l0 : k + 1); l < l1; ++l) {
                        if (hamming_bytes(a, Z + (size_t)l * N, N) < thresh) {
                            ++mine;
                            ++loc[l];
                        }
                    }
                    loc[k] += mine;
                }
            }
        }
#pragma omp critical
        for (int k = 0; k < M; ++k) n[k] += loc[k];
        free(loc);
    }
}

/* Pi[n], Pij[n][n] (row-major == column-major: symmetric), n = N*(q-1).
 * Every cell is accumulated over k = 0..M-1 in order, then divided by Meff once. */
void orc_frequencies(const int8_t *Z, int N, int M, int q, const double *W, double Meff,
                     double *Pi, double *Pij)
{
    const int s = q - 1;
    const size_t n = (size_t)N * s;
    memset(Pi, 0, n * sizeof(double));
    for (int k = 0; k < M; ++k) {
        const int8_t *z = Z + (size_t)k * N;
        for (int i = 0; i < N; ++i)
            if (z[i] < q) Pi[(size_t)i * s + (z[i] - 1)] += W[k];
    }
    for (size_t x = 0; x < n; ++x) Pi[x] /= Meff;

#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < N; ++i) {
        /* rows i*s .. i*s+s-1, columns >= i*s */
        for (int a = 0; a < s; ++a) memset(Pij + ((size_t)i * s + a) * n, 0, n * sizeof(double));
        for (int k = 0; k < M; ++k) {
            const int8_t *z = Z + (size_t)k * N;
            const int a = z[i];
            if (a >= q) continue;
            double *row = Pij + ((size_t)i * s + (a - 1)) * n;
            const double w = W[k];
            for (int j = i; j < N; ++j) {
                const int b = z[j];
                if (b < q) row[(size_t)j * s + (b - 1)] += w;
            }
        }
        for (int a = 0; a < s; ++a) {
            double *row = Pij + ((size_t)i * s + a) * n;
            for (size_t y = (size_t)i * s; y < n; ++y) row[y] /= Meff;
        }
    }
    /* mirror: the diagonal blocks were filled in full (j == i covers both orders) */
#pragma omp parallel for schedule(static)
    for (size_t x = 0; x < n; ++x) {
        const size_t i = x / s;
        for (size_t y = (i + 1) * s; y < n; ++y) Pij[y * n + x] = Pij[x * n + y];
    }
}

/* FN[i][j] = || B - rowmean - colmean + mean ||_F over the s x s block (i,j) of mJ */
void orc_fn(const double *mJ, int N, int s, double *FN)
{
    const size_t n = (size_t)N * s;
    memset(FN, 0, (size_t)N * N * sizeof(double));
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < N - 1; ++i) {
        double rm[32], cm[32];
        for (int j = i + 1; j < N; ++j) {
            const double *B = mJ + (size_t)i * s * n + (size_t)j * s;
            double tot = 0.0;
            for (int a = 0; a < s; ++a) rm[a] = 0.0;
            for (int b = 0; b < s; ++b) cm[b] = 0.0;
            for (int a = 0; a < s; ++a)
                for (int b = 0; b < s; ++b) {
                    const double x = B[(size_t)a * n + b];
                    rm[a] += x;
                    cm[b] += x;
                    tot += x;
                }
            for (int a = 0; a < s; ++a) rm[a] /= s;
            for (int b = 0; b < s; ++b) cm[b] /= s;
            tot /= (double)s * s;
            double f = 0.0;
            for (int a = 0; a < s; ++a)
                for (int b = 0; b < s; ++b) {
                    const double kx = B[(size_t)a * n + b] - rm[a] - cm[b] + tot;
                    f += kx * kx;
                }
            f = sqrt(f);
            FN[(size_t)i * N + j] = f;
            FN[(size_t)j * N + i] = f;
        }
    }
}
