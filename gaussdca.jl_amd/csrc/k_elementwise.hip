// HBM-bound elementwise kernels: the operator-level add_pseudocount (DCAUtils; reference call
// site src/GaussDCA.jl:30) and compute_C (src/GaussDCA.jl:76), plus layout helpers around the
// SPD inverse.  One pass each, 8-byte accesses contiguous along columns (column-major).
// In the fused pipeline add_pseudocount and compute_C never run as kernels of their own: the
// pair-tally epilogue (k_tally.hip) applies both while writing C.
#include "gdca_internal.h"
#include "gdca_launch.h"

// Pij' = (1-pc) Pij + pc/q^2 off the diagonal blocks; diagonal blocks (1-pc) Pij + (pc/q) I
struct k_add_pseudocount_args {
    const double *Pi_true;
    const double *Pij_true;
    int n;
    int sdim;
    int q;
    double pc;
    double *Pi;
    double *Pij;
};
static inline k_add_pseudocount_args k_add_pseudocount_mk(const double *Pi_true, const double *Pij_true, int n, int sdim, int q, double pc, double *Pi, double *Pij)
{
    return k_add_pseudocount_args{Pi_true, Pij_true, n, sdim, q, pc, Pi, Pij};
}
template <int CAP>
__global__ __launch_bounds__(256) void k_add_pseudocount(const BatchArgs<k_add_pseudocount_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const double *__restrict__ Pi_true = a_.Pi_true;
    const double *__restrict__ Pij_true = a_.Pij_true;
    int n = a_.n;
    int sdim = a_.sdim;
    int q = a_.q;
    double pc = a_.pc;
    double *__restrict__ Pi = a_.Pi;
    double *__restrict__ Pij = a_.Pij;
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (r >= n) return;
    const double pcq = pc / (double)q;
    const size_t e = (size_t)r + (size_t)c * n;
    const double x = (1.0 - pc) * Pij_true[e];
    double v;
    if (r / sdim == c / sdim)
        v = (r == c) ? (x + pcq) : x;
    else
        v = x + pcq / (double)q;
    Pij[e] = v;
    if (c == 0) Pi[r] = (1.0 - pc) * Pi_true[r] + pcq;
}

void gdca_launch_add_pseudocount(hipStream_t s, const double *Pi_true, const double *Pij_true, int N, int q, double pc,
                                 double *Pi, double *Pij)
{
    const int sdim = q - 1, n = N * sdim;
    (gdca_launch<k_add_pseudocount_args, k_add_pseudocount<1>, k_add_pseudocount<GDCA_MAXB>>(dim3((n + 255) / 256, n), dim3(256), 0, s, k_add_pseudocount_mk(Pi_true, Pij_true, n, sdim, q, pc, Pi, Pij)));
}

struct k_covariance_args {
    const double *Pi;
    const double *Pij;
    int n;
    double *C;
};
static inline k_covariance_args k_covariance_mk(const double *Pi, const double *Pij, int n, double *C)
{
    return k_covariance_args{Pi, Pij, n, C};
}
template <int CAP>
__global__ __launch_bounds__(256) void k_covariance(const BatchArgs<k_covariance_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const double *__restrict__ Pi = a_.Pi;
    const double *__restrict__ Pij = a_.Pij;
    int n = a_.n;
    double *__restrict__ C = a_.C;
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (r >= n) return;
    const size_t e = (size_t)r + (size_t)c * n;
    C[e] = Pij[e] - Pi[r] * Pi[c];
}

void gdca_launch_covariance(hipStream_t s, const double *Pi, const double *Pij, int n, double *C)
{
    (gdca_launch<k_covariance_args, k_covariance<1>, k_covariance<GDCA_MAXB>>(dim3((n + 255) / 256, n), dim3(256), 0, s, k_covariance_mk(Pi, Pij, n, C)));
}

// rows / columns >= n of the padded matrix become identity (SPD, decoupled from the real block): only the two strips are launched
// (blockIdx.y = 0: columns n .. n_pad-1, all rows; 1: rows n .. n_pad-1 of the columns left of them) -- as a grid over the whole
// padded matrix with an early exit it was 400 000 empty workgroups, 87 us, at n = 10 000
struct k_pad_identity_args {
    double *A;
    int n;
    int n_pad;
};
static inline k_pad_identity_args k_pad_identity_mk(double *A, int n, int n_pad)
{
    return k_pad_identity_args{A, n, n_pad};
}
template <int CAP>
__global__ __launch_bounds__(256) void k_pad_identity(const BatchArgs<k_pad_identity_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    double *__restrict__ A = a_.A;
    int n = a_.n;
    int n_pad = a_.n_pad;
    const int pad = n_pad - n;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    int r, c;
    if (blockIdx.y == 0) {
        if (idx >= (long long)pad * n_pad) return;
        c = n + (int)(idx / n_pad);
        r = (int)(idx % n_pad);
    } else {
        if (idx >= (long long)pad * n) return;
        c = (int)(idx / pad);
        r = n + (int)(idx % pad);
    }
    A[(size_t)r + (size_t)c * n_pad] = (r == c) ? 1.0 : 0.0;
}

void gdca_launch_pad_identity(hipStream_t s, double *A, int n, int n_pad)
{
    if (n_pad == n) return;
    const long long strip = (long long)(n_pad - n) * n_pad;
    (gdca_launch<k_pad_identity_args, k_pad_identity<1>, k_pad_identity<GDCA_MAXB>>(dim3((unsigned)((strip + 255) / 256), 2), dim3(256), 0, s, k_pad_identity_mk(A, n, n_pad)));
}

struct k_copy_in_args {
    const double *src;
    int n;
    double *dst;
    int n_pad;
};
static inline k_copy_in_args k_copy_in_mk(const double *src, int n, double *dst, int n_pad)
{
    return k_copy_in_args{src, n, dst, n_pad};
}
template <int CAP, bool NEG>
__global__ __launch_bounds__(256) void k_copy_in(const BatchArgs<k_copy_in_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const double *__restrict__ src = a_.src;
    int n = a_.n;
    double *__restrict__ dst = a_.dst;
    int n_pad = a_.n_pad;
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (r >= n_pad) return;
    double v;
    if (r < n && c < n)
        v = NEG ? -src[(size_t)r + (size_t)c * n] : src[(size_t)r + (size_t)c * n];
    else
        v = (r == c) ? 1.0 : 0.0;
    dst[(size_t)r + (size_t)c * n_pad] = v;
}

void gdca_launch_copy_in(hipStream_t s, const double *src, int n, double *dst, int n_pad)
{
    (gdca_launch<k_copy_in_args, k_copy_in<1, false>, k_copy_in<GDCA_MAXB, false>>(dim3((n_pad + 255) / 256, n_pad), dim3(256), 0, s, k_copy_in_mk(src, n, dst, n_pad)));
}

void gdca_launch_copy_in_neg(hipStream_t s, const double *src, int n, double *dst, int n_pad)
{
    (gdca_launch<k_copy_in_args, k_copy_in<1, true>, k_copy_in<GDCA_MAXB, true>>(dim3((n_pad + 255) / 256, n_pad), dim3(256), 0, s, k_copy_in_mk(src, n, dst, n_pad)));
}

// dst[r][c] = dst[c][r] = -A[max(r,c)][min(r,c)]  through a 32 x 32 LDS tile so that both the
// read of the lower triangle and the two writes stay contiguous along columns
struct k_copy_out_neg_sym_args {
    const double *A;
    int n_pad;
    double *dst;
    int n;
};
static inline k_copy_out_neg_sym_args k_copy_out_neg_sym_mk(const double *A, int n_pad, double *dst, int n)
{
    return k_copy_out_neg_sym_args{A, n_pad, dst, n};
}
template <int CAP>
__global__ __launch_bounds__(256) void k_copy_out_neg_sym(const BatchArgs<k_copy_out_neg_sym_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const double *__restrict__ A = a_.A;
    int n_pad = a_.n_pad;
    double *__restrict__ dst = a_.dst;
    int n = a_.n;
    __shared__ double tile[32][33];
    const int bi = blockIdx.x, bj = blockIdx.y;  // tile (bi, bj) of the lower triangle: bi >= bj
    if (bi < bj) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int cc = ty; cc < 32; cc += 8) {
        const int r = bi * 32 + tx, c = bj * 32 + cc;
        double v = 0.0;
        if (r < n && c < n) {
            v = -A[(size_t)r + (size_t)c * n_pad];
            if (r >= c) dst[(size_t)r + (size_t)c * n] = v;
        }
        tile[cc][tx] = v;
    }
    __syncthreads();
    // mirror: element (c, r) of dst for r > c, written contiguous along c
    for (int rr = ty; rr < 32; rr += 8) {
        const int c = bj * 32 + tx, r = bi * 32 + rr;
        if (r < n && c < n && r > c) dst[(size_t)c + (size_t)r * n] = tile[tx][rr];
    }
}

void gdca_launch_copy_out_neg_sym(hipStream_t s, const double *A, int n_pad, double *dst, int n)
{
    const int nt = (n + 31) / 32;
    (gdca_launch<k_copy_out_neg_sym_args, k_copy_out_neg_sym<1>, k_copy_out_neg_sym<GDCA_MAXB>>(dim3(nt, nt), dim3(256), 0, s, k_copy_out_neg_sym_mk(A, n_pad, dst, n)));
}

struct k_save_diag_blocks_args {
    const double *C;
    size_t ld;
    int N;
    int sdim;
    double *D;
};
static inline k_save_diag_blocks_args k_save_diag_blocks_mk(const double *C, size_t ld, int N, int sdim, double *D)
{
    return k_save_diag_blocks_args{C, ld, N, sdim, D};
}
template <int CAP>
__global__ __launch_bounds__(256) void k_save_diag_blocks(const BatchArgs<k_save_diag_blocks_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const double *__restrict__ C = a_.C;
    size_t ld = a_.ld;
    int N = a_.N;
    int sdim = a_.sdim;
    double *__restrict__ D = a_.D;
    (void)N;
    const int i = blockIdx.x;
    for (int e = threadIdx.x; e < sdim * sdim; e += 256) {
        const int r = e % sdim, c = e / sdim;
        D[(size_t)i * sdim * sdim + e] = C[(size_t)(i * sdim + r) + (size_t)(i * sdim + c) * ld];
    }
}

void gdca_launch_save_diag_blocks(hipStream_t s, const double *C, size_t ld, int N, int sdim, double *D)
{
    (gdca_launch<k_save_diag_blocks_args, k_save_diag_blocks<1>, k_save_diag_blocks<GDCA_MAXB>>(dim3(N), dim3(256), 0, s, k_save_diag_blocks_mk(C, ld, N, sdim, D)));
}

// The run's scalars into the context's pinned host copy, as the LAST kernel of an enqueued run: gdca_run_collect then needs a stream
// synchronisation and nothing else.  (A device-to-host hipMemcpyAsync of these 200 bytes is a blit KERNEL of the runtime's: issued
// at collect time it queued behind whatever persistent sweep another context of the pipeline had resident -- the collect of batch
// b-1 returned when the sweep of batch b ended, and only then did the host start to enqueue batch b+1:
// profiles/r05_B_merged8_timeline.log.)
struct k_publish_scalars_args {
    const unsigned *src;
    unsigned *dst_host;
    int words;
};
static inline k_publish_scalars_args k_publish_scalars_mk(const unsigned *src, unsigned *dst_host, int words)
{
    return k_publish_scalars_args{src, dst_host, words};
}
template <int CAP>
__global__ __launch_bounds__(64) void k_publish_scalars(const BatchArgs<k_publish_scalars_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const unsigned *__restrict__ src = a_.src;
    unsigned *__restrict__ dst_host = a_.dst_host;
    int words = a_.words;
    for (int i = threadIdx.x; i < words; i += 64) dst_host[i] = src[i];
    __threadfence_system();
}

void gdca_launch_publish_scalars(hipStream_t s, const gdca_dev_scalars *sc, gdca_dev_scalars *host_mapped)
{
    static_assert(sizeof(gdca_dev_scalars) % 4 == 0, "copied as 32-bit words");
    (gdca_launch<k_publish_scalars_args, k_publish_scalars<1>, k_publish_scalars<GDCA_MAXB>>(dim3(1), dim3(64), 0, s, k_publish_scalars_mk((const unsigned *)sc, (unsigned *)host_mapped, (int)(sizeof(gdca_dev_scalars) / 4))));
}
