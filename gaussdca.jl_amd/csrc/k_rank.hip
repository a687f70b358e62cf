// compute_ranking(S, min_separation) on the device (reference: src/GaussDCA.jl:88-99, call :44):
//     R = [(i, j, S[j, i]) for i = 1:N-sep for j = i+sep:N];  sort!(R, by = x -> x[3], rev = true)
// a stable sort by `isless` on the score, reversed: NaN first, 0.0 before -0.0, equal scores in generation order.  Same 64-bit key
// as the host form (gdca_host.cpp, gdca_ranking), so the two agree entry for entry; the point of doing it here is that the run
// then hands back the ranking itself -- the host sort was 1.3 ms of a 31 ms gDCA() at N = 500 and 6.2 of 190 at N = 1000.
//
// Stable LSD radix sort, 8 bits a pass, eight passes, three small launches a pass:
//   k_rank_hist     per chunk of 4096 entries: counts of the pass's digit                            -> hist[digit][chunk]
//   k_rank_scan     one workgroup: hist -> global start of (digit, chunk), digit-major
//   k_rank_scatter  per chunk, thread t owns 16 CONSECUTIVE entries: its rank among equal digits = entries of earlier threads
//                   (a [thread][digit] table of u16 in LDS, prefix over the threads by one thread per digit) + its own earlier ones
// HBM-bound in principle (n x 12 B read and written per pass), launch-bound in practice at these sizes (n = 1.2e5 .. 5e5).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "gdca_internal.h"
#include "gdca_launch.h"

namespace {

constexpr int RK_THREADS = 256, RK_E = 16, RK_CHUNK = RK_THREADS * RK_E, RK_ROW = 258;  // (row of 258 u16 = 129 words: equal digits of different threads fall into different banks)

__device__ __forceinline__ uint64_t rank_key(double x)
{
    const uint64_t u = (uint64_t)__double_as_longlong(x);
    uint64_t asc = (u >> 63) ? ~u : (u | 0x8000000000000000ull);  // ascending in isless order: -0.0 < 0.0
    if (x != x) asc = ~0ull;                                        // every NaN greatest, all equal
    return ~asc;                                                    // descending
}

// row i = blockIdx.x + 1 of the enumeration: entries (i, j), j = i + sep .. N, at offset (i-1)(N-sep) - (i-1)(i-2)/2
struct k_rank_keys_args {
    const double *S;
    int N;
    int sep;
    uint64_t *key;
    uint32_t *val;
};
static inline k_rank_keys_args k_rank_keys_mk(const double *S, int N, int sep, uint64_t *key, uint32_t *val)
{
    return k_rank_keys_args{S, N, sep, key, val};
}
template <int CAP>
__global__ __launch_bounds__(256) void k_rank_keys(const BatchArgs<k_rank_keys_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const double *__restrict__ S = a_.S;
    int N = a_.N;
    int sep = a_.sep;
    uint64_t *__restrict__ key = a_.key;
    uint32_t *__restrict__ val = a_.val;
    const int i = (int)blockIdx.x + 1;
    const long long d = N - sep;
    const long long off = (long long)(i - 1) * d - (long long)(i - 1) * (i - 2) / 2;
    for (int j = i + sep + (int)threadIdx.x; j <= N; j += 256) {
        const long long t = off + (j - i - sep);
        key[t] = rank_key(S[(size_t)(j - 1) + (size_t)(i - 1) * N]);
        val[t] = ((uint32_t)i << 16) | (uint32_t)j;
    }
}

struct k_rank_hist_args {
    const uint64_t *key;
    size_t n;
    int shift;
    uint32_t *hist;
    int nchunk;
};
static inline k_rank_hist_args k_rank_hist_mk(const uint64_t *key, size_t n, int shift, uint32_t *hist, int nchunk)
{
    return k_rank_hist_args{key, n, shift, hist, nchunk};
}
template <int CAP>
__global__ __launch_bounds__(256) void k_rank_hist(const BatchArgs<k_rank_hist_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const uint64_t *__restrict__ key = a_.key;
    size_t n = a_.n;
    int shift = a_.shift;
    uint32_t *__restrict__ hist = a_.hist;
    int nchunk = a_.nchunk;
    __shared__ uint32_t h[256];
    const int tid = (int)threadIdx.x;
    h[tid] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * RK_CHUNK;
#pragma unroll 4
    for (int e = 0; e < RK_E; ++e) {
        const size_t t = base + (size_t)e * 256 + tid;
        if (t < n) atomicAdd(&h[(key[t] >> shift) & 255], 1u);
    }
    __syncthreads();
    hist[(size_t)tid * nchunk + blockIdx.x] = h[tid];
}

// one workgroup; thread d owns digit d's row
struct k_rank_scan_args {
    uint32_t *hist;
    int nchunk;
};
static inline k_rank_scan_args k_rank_scan_mk(uint32_t *hist, int nchunk)
{
    return k_rank_scan_args{hist, nchunk};
}
template <int CAP>
__global__ __launch_bounds__(256) void k_rank_scan(const BatchArgs<k_rank_scan_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    uint32_t *__restrict__ hist = a_.hist;
    int nchunk = a_.nchunk;
    __shared__ uint32_t tot[256];
    const int d = (int)threadIdx.x;
    uint32_t *row = hist + (size_t)d * nchunk;
    uint32_t run = 0;
    for (int b = 0; b < nchunk; ++b) {
        const uint32_t c = row[b];
        row[b] = run;
        run += c;
    }
    tot[d] = run;
    __syncthreads();
    if (d == 0) {
        uint32_t acc = 0;
        for (int k = 0; k < 256; ++k) {
            const uint32_t c = tot[k];
            tot[k] = acc;
            acc += c;
        }
    }
    __syncthreads();
    const uint32_t g = tot[d];
    if (g != 0)
        for (int b = 0; b < nchunk; ++b) row[b] += g;
}

struct k_rank_scatter_args {
    const uint64_t *kin;
    const uint32_t *vin;
    uint64_t *kout;
    uint32_t *vout;
    size_t n;
    int shift;
    const uint32_t *hist;
    int nchunk;
};
static inline k_rank_scatter_args k_rank_scatter_mk(const uint64_t *kin, const uint32_t *vin, uint64_t *kout, uint32_t *vout, size_t n, int shift, const uint32_t *hist, int nchunk)
{
    return k_rank_scatter_args{kin, vin, kout, vout, n, shift, hist, nchunk};
}
template <int CAP>
__global__ __launch_bounds__(256) void k_rank_scatter(const BatchArgs<k_rank_scatter_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const uint64_t *__restrict__ kin = a_.kin;
    const uint32_t *__restrict__ vin = a_.vin;
    uint64_t *__restrict__ kout = a_.kout;
    uint32_t *__restrict__ vout = a_.vout;
    size_t n = a_.n;
    int shift = a_.shift;
    const uint32_t *__restrict__ hist = a_.hist;
    int nchunk = a_.nchunk;
    extern __shared__ __attribute__((aligned(16))) uint16_t cnt[];  // [thread][RK_ROW]
    __shared__ uint32_t start[256];
    const int tid = (int)threadIdx.x;
    for (int e = tid; e < RK_THREADS * RK_ROW / 2; e += 256) reinterpret_cast<uint32_t *>(cnt)[e] = 0;
    start[tid] = hist[(size_t)tid * nchunk + blockIdx.x];
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * RK_CHUNK + (size_t)tid * RK_E;
    uint64_t k[RK_E];
    uint32_t v[RK_E];
    uint16_t own[RK_E];
    uint16_t *mine = cnt + tid * RK_ROW;
#pragma unroll
    for (int e = 0; e < RK_E; ++e) {
        const size_t t = base + e;
        k[e] = t < n ? kin[t] : 0;
        v[e] = t < n ? vin[t] : 0;
    }
#pragma unroll
    for (int e = 0; e < RK_E; ++e) {
        if (base + e < n) {
            const int d = (int)((k[e] >> shift) & 255);
            own[e] = mine[d];
            mine[d] = (uint16_t)(own[e] + 1);
        }
    }
    __syncthreads();
    {  // exclusive prefix over the threads, digit `tid`
        uint32_t run = 0;
        for (int t = 0; t < RK_THREADS; ++t) {
            const uint16_t c = cnt[t * RK_ROW + tid];
            cnt[t * RK_ROW + tid] = (uint16_t)run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < RK_E; ++e) {
        if (base + e < n) {
            const int d = (int)((k[e] >> shift) & 255);
            const size_t pos = (size_t)start[d] + mine[d] + own[e];
            kout[pos] = k[e];
            vout[pos] = v[e];
        }
    }
}

struct k_rank_emit_args {
    const uint32_t *val;
    size_t n;
    const double *S;
    int N;
    int32_t *ii;
    int32_t *jj;
    double *sc;
};
static inline k_rank_emit_args k_rank_emit_mk(const uint32_t *val, size_t n, const double *S, int N, int32_t *ii, int32_t *jj, double *sc)
{
    return k_rank_emit_args{val, n, S, N, ii, jj, sc};
}
template <int CAP>
__global__ __launch_bounds__(256) void k_rank_emit(const BatchArgs<k_rank_emit_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const uint32_t *__restrict__ val = a_.val;
    size_t n = a_.n;
    const double *__restrict__ S = a_.S;
    int N = a_.N;
    int32_t *__restrict__ ii = a_.ii;
    int32_t *__restrict__ jj = a_.jj;
    double *__restrict__ sc = a_.sc;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const uint32_t v = val[e];
    const int i = (int)(v >> 16), j = (int)(v & 0xffffu);
    ii[e] = i;
    jj[e] = j;
    sc[e] = S[(size_t)(j - 1) + (size_t)(i - 1) * N];
}

inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

size_t gdca_ranking_ws_bytes(long long len)
{
    const size_t n = (size_t)len, nchunk = (n + RK_CHUNK - 1) / RK_CHUNK;
    return 2 * up256(n * 8) + 2 * up256(n * 4) + up256(256 * nchunk * 4) + 2 * up256(n * 4) + up256(n * 8);
}

void gdca_launch_ranking(hipStream_t s, const double *S_dev, int N, int sep, long long len, void *ws, int32_t **ii, int32_t **jj, double **sc)
{
    const size_t n = (size_t)len;
    const int nchunk = (int)((n + RK_CHUNK - 1) / RK_CHUNK);
    char *p = (char *)ws;
    uint64_t *key[2];
    uint32_t *val[2];
    key[0] = (uint64_t *)p, p += up256(n * 8);
    key[1] = (uint64_t *)p, p += up256(n * 8);
    val[0] = (uint32_t *)p, p += up256(n * 4);
    val[1] = (uint32_t *)p, p += up256(n * 4);
    uint32_t *hist = (uint32_t *)p;
    p += up256((size_t)256 * nchunk * 4);
    *ii = (int32_t *)p, p += up256(n * 4);
    *jj = (int32_t *)p, p += up256(n * 4);
    *sc = (double *)p;
    const size_t lds = (size_t)RK_THREADS * RK_ROW * sizeof(uint16_t);
    // (dynamic LDS beyond 48 KB: the launcher raises the kernel's limit, gdca_launch.h)
    (gdca_launch<k_rank_keys_args, k_rank_keys<1>, k_rank_keys<GDCA_MAXB>>(dim3((unsigned)(N - sep)), dim3(256), 0, s, k_rank_keys_mk(S_dev, N, sep, key[0], val[0])));
    for (int pass = 0; pass < 8; ++pass) {
        const int a = pass & 1, b = a ^ 1, shift = 8 * pass;
        (gdca_launch<k_rank_hist_args, k_rank_hist<1>, k_rank_hist<GDCA_MAXB>>(dim3((unsigned)nchunk), dim3(256), 0, s, k_rank_hist_mk(key[a], n, shift, hist, nchunk)));
        (gdca_launch<k_rank_scan_args, k_rank_scan<1>, k_rank_scan<GDCA_MAXB>>(dim3(1), dim3(256), 0, s, k_rank_scan_mk(hist, nchunk)));
        (gdca_launch<k_rank_scatter_args, k_rank_scatter<1>, k_rank_scatter<GDCA_MAXB>>(dim3((unsigned)nchunk), dim3(256), lds, s, k_rank_scatter_mk(key[a], val[a], key[b], val[b], n, shift, hist, nchunk)));
    }
    (gdca_launch<k_rank_emit_args, k_rank_emit<1>, k_rank_emit<GDCA_MAXB>>(dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, k_rank_emit_mk(val[0], n, S_dev, N, *ii, *jj, *sc)));
}
