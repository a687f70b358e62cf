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
// 8 n (n - s) / 2 bytes in all).  DI: the three small products and a Householder
// tridiagonalisation per pair (one wave each, LDS-resident), then implicit QL on the
// tridiagonals, one lane per pair -- latency/VALU-bound, no HBM traffic to speak of.
#include "gdca_internal.h"
#include "gdca_launch.h"

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
// HBM-bound: 8 s^2 bytes per site pair, one pass over the lower block triangle (399 MB at N = 500).  One wave per pair reading its
// own s column segments of 8 s bytes (160 B at a stride of ld) got 2.5 TB/s out of that.  Now a workgroup takes FN_PG
// consecutive row sites j of ONE column site i: in each of the s columns their blocks are one contiguous run of FN_PG 8 s
// bytes (1280 B), fetched with 16-byte loads, all s runs in flight at once, into an LDS image [column][run] (row stride
// padded so that a column walk hits different banks); then every wave scores its pairs from LDS.  The arithmetic of a pair --
// row means, column means, their order of summation, the per-lane partial sums of the norm and the butterfly that adds them -- is
// unchanged: scores are bit for bit those of the old kernel.
#define FN_PG 4

// FN_MAXU: 16-byte loads per thread that hold a workgroup's image, ceil(s (FN_PG s / 2) / 256).  SDIM: s as a compile-time
// constant (20 = the q = 21 alphabet: the loops over a block unroll, their LDS reads are issued back to back instead of one
// round trip per addition) or 0 = run-time s.
struct k_fn_args {
    const double *A;
    size_t ld;
    int N;
    int sdim_rt;
    double *S;
};
static inline k_fn_args k_fn_mk(const double *A, size_t ld, int N, int sdim_rt, double *S)
{
    return k_fn_args{A, ld, N, sdim_rt, S};
}
template <int CAP, int FN_MAXU, int SDIM>
__global__ __launch_bounds__(256) void k_fn(const BatchArgs<k_fn_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const double *__restrict__ A = a_.A;
    size_t ld = a_.ld;
    int N = a_.N;
    int sdim_rt = a_.sdim_rt;
    double *__restrict__ S = a_.S;
    const int sdim = SDIM ? SDIM : sdim_rt;
    extern __shared__ __attribute__((aligned(16))) double fsm[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ss = sdim * sdim;
    // grid: x = row chunk counted from the column site's first one, y = column site -- consecutive workgroups walk DOWN the s
    // columns of one site, so the reads in flight at any moment are s long sequential streams (with the column site running
    // fastest they were thousands of 1280-byte pieces 8 ld bytes apart)
    const int i = blockIdx.y, J = (i + 1) / FN_PG + (int)blockIdx.x;
    if (J * FN_PG >= N) return;
    const int j_lo = max(J * FN_PG, i + 1), j_hi = min(N, J * FN_PG + FN_PG);
    const int L = (j_hi - j_lo) * sdim;          // rows of the run
    const int Lp = FN_PG * sdim + 2;             // row stride of the LDS image (even: 16-byte stores stay aligned)
    double *img = fsm;                           // img[c * Lp + (row of the run)]
    double *rm = fsm + (size_t)sdim * Lp + (size_t)wv * 64, *cm = rm + 32;
    const double *src = A + (size_t)j_lo * sdim + (size_t)i * sdim * ld;
    if (((j_lo * sdim) & 1) == 0 && (L & 1) == 0 && sdim * (L >> 1) <= 256 * FN_MAXU) {
        // every load of the thread is issued before the first one is stored: the workgroup's whole image is in flight at once
        // (as a load / store loop each trip waited out its own load: 1-2 us apiece under load, seven trips at s = 20)
        const int L2 = L >> 1, total = sdim * L2;
        double2 v[FN_MAXU];
        int off[FN_MAXU];
#pragma unroll
        for (int u = 0; u < FN_MAXU; ++u) {
            const int e = min(tid + 256 * u, total - 1);  // (past the end: the last unit once more, loaded and not stored)
            const int c = e / L2, r2 = e - c * L2;
            off[u] = c * Lp + 2 * r2;
            v[u] = *reinterpret_cast<const double2 *>(src + (size_t)c * ld + 2 * r2);
        }
#pragma unroll
        for (int u = 0; u < FN_MAXU; ++u)
            if (tid + 256 * u < total) *reinterpret_cast<double2 *>(img + off[u]) = v[u];
    } else {
        for (int e = tid; e < sdim * L; e += 256) {
            const int c = e / L, r = e - c * L;
            img[c * Lp + r] = src[(size_t)c * ld + r];
        }
    }
    __syncthreads();
    // The scoring of a pair is bound by instruction issue, not by LDS or HBM (one wave per pair; ~2700 clocks a pair as first
    // written): no integer division in the loops (a lane's (r, c) advances by 64 elements at a time), row means and column means
    // in ONE loop (lanes 0.. take rows, lanes 32.. columns: the same sums in the same order).
    const int r_step = 64 % sdim, c_step = 64 / sdim;
    const int r0 = lane % sdim, c0 = lane / sdim;
    const bool is_row = lane < 32;
    const int mlane = is_row ? lane : lane - 32;
    for (int t = wv; t < j_hi - j_lo; t += 4) {   // (wave-uniform trip count: the barriers below are per wave, not per workgroup)
        const double *blk = img + t * sdim;       // element (r, c) of the pair's block: blk[r + c * Lp]
        if (mlane < sdim) {
            // rows: sum over c of blk[lane + c Lp];  columns: sum over r of blk[r + c Lp]
            const double *q = blk + (is_row ? mlane : mlane * Lp);
            const int step = is_row ? Lp : 1;
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < sdim; ++k) a += q[k * step];
            (is_row ? rm : cm)[mlane] = a / (double)sdim;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        double tot = 0.0;
#pragma unroll
        for (int r = 0; r < sdim; ++r) tot += rm[r];
        tot /= (double)sdim;  // = sum(block) / s^2
        double f = 0.0;
        int r = r0, c = c0;
#pragma unroll
        for (int e = lane; e < ss; e += 64) {
            const double kx = blk[r + c * Lp] - rm[r] - cm[c] + tot;
            f += kx * kx;
            r += r_step;
            c += c_step;
            if (r >= sdim) {
                r -= sdim;
                ++c;
            }
        }
        f = wave_sum(f);
        if (lane == 0) {
            const int j = j_lo + t;
            const double v = sqrt(f);
            S[(size_t)i + (size_t)j * N] = v;
            S[(size_t)j + (size_t)i * N] = v;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // rm / cm are rewritten by the wave's next pair
    }
}

// ---- FN, s = 20 (the q = 21 alphabet): a persistent form, three pairs per wave --------------------------------------------------
// `k_fn` above gets 0.44 of the HBM peak: half of its 62 000 workgroups leave at once (dispatch of the empty half of
// the triangle is a third of its time), a workgroup's loads and its scoring do not overlap, and one wave per pair spends ~220
// instructions on a 20 x 20 block with most lanes idle in the means.  Here FN20_WG workgroups per compute unit walk a LIST of items
// -- (column site i, FN20_PG consecutive row sites below it): one contiguous run of FN20_PG 160 B per column -- a stride apart; the
// NEXT item's twenty runs are requested (16-byte loads into registers) before the current one is scored from LDS, and a wave scores
// THREE pairs at a time: lanes 20 g .. 20 g + 19 own pair g, lane l of the group holds ROW l of the block in registers (one LDS read
// per element), sums it, sums column l from LDS, and the Frobenius norm of the centred block is row-wise partial sums added in row
// order.  ~70 instructions per pair.  Same formula as rule 9 of the survey (src/GaussDCA.jl:39), another order of summation than
// `k_fn`: scores agree to rounding (<= 1e-15 relative).
#ifndef FN20_PG
#define FN20_PG 6          // row sites per item (per column a run of FN20_PG 160 B)
#define FN20_THREADS 128   // threads per workgroup: 3 pairs per wave and round, FN20_PG = 3 FN20_THREADS / 64
#define FN20_WG 6          // workgroups per compute unit: what 142 registers and 22 KB of LDS let run at once
#endif
#define FN20_LP (FN20_PG * 20 + 2)
#define FN20_MAXU 10       // 16-byte loads per thread that hold an item: 20 columns x (10 FN20_PG) double2 / FN20_THREADS, rounded up
static_assert(20 * 10 * FN20_PG <= FN20_MAXU * FN20_THREADS, "ten units per thread hold an item");

// items of the column sites before site i: sum_{m = 1..i} [(N - 1) / PG - m / PG + 1], in closed form
__host__ __device__ __forceinline__ int fn20_first(int i, int N)
{
    const int C = (N - 1) / FN20_PG, q = i / FN20_PG, r = i - q * FN20_PG;
    return i * (C + 1) - (FN20_PG * (q * (q - 1) / 2) + q * (r + 1));
}

// item t of the list: column site i, row chunk J (rows J PG .. J PG + PG - 1, from i + 1 on); items of one column site are
// consecutive, so the streams in flight at any moment are few and long
__device__ __forceinline__ void fn20_item(int t, int N, int &i, int &J)
{
    int lo = 0, hi = N - 1;   // fn20_first(lo) <= t < fn20_first(hi)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (fn20_first(mid, N) <= t) lo = mid; else hi = mid;
    }
    i = lo;
    J = (i + 1) / FN20_PG + (t - fn20_first(i, N));
}

// One 16-byte unit of an item in flight: where it goes in the LDS image, and the data.  Ten NAMED units per thread (FN20_UNITS):
// as arrays `v[u]`, `off[u]` written at the bottom of the persistent loop and read at its top they stayed in scratch (176 B per lane).
struct Fn20Unit {
    double2 v;
    int off;
};
#define FN20_UNITS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9)

__device__ __forceinline__ void fn20_unit_load(Fn20Unit &q, int u, int tid, int n_units, int L2, const double *__restrict__ src, size_t ld)
{
    const int e = min(tid + FN20_THREADS * u, n_units - 1);  // (past the end: the last unit once more, loaded and not stored)
    const int c = e / L2, r2 = e - c * L2;
    q.off = c * FN20_LP + 2 * r2;
    q.v = *reinterpret_cast<const double2 *>(src + (size_t)c * ld + 2 * r2);
}

struct k_fn20_args {
    const double *A;
    size_t ld;
    int N;
    int total;
    double *S;
};
static inline k_fn20_args k_fn20_mk(const double *A, size_t ld, int N, int total, double *S)
{
    return k_fn20_args{A, ld, N, total, S};
}
template <int CAP>
__global__ __launch_bounds__(FN20_THREADS) void k_fn20(const BatchArgs<k_fn20_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const double *__restrict__ A = a_.A;
    size_t ld = a_.ld;
    int N = a_.N;
    int total = a_.total;
    double *__restrict__ S = a_.S;
    constexpr int s = 20, Lp = FN20_LP;
    static_assert(FN20_MAXU == 10, "FN20_UNITS lists ten units");
    __shared__ __attribute__((aligned(16))) double img[s * Lp];   // img[c * Lp + (row of the run)]
    __shared__ double red[FN20_THREADS / 64][3][3][s];             // per wave and pair: row means, column means, partial norms
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int g = lane / s, l = lane - g * s;                      // pair of the wave's three, row / column inside it (lanes 60..63: none)
#define FN20_DECL(u) Fn20Unit q##u;
    FN20_UNITS(FN20_DECL)
    int n_units = 0, i = 0, j_lo = 0, j_hi = 0;
    // the loads of item T (uniform): the ten units, n_units; and its (i, j_lo, j_hi)
#define FN20_LOAD(u) fn20_unit_load(q##u, u, tid, n_units, L2_, src_, ld);
#define FN20_REQUEST(T)                                                       \
    do {                                                                      \
        int J_;                                                               \
        fn20_item((T), N, i, J_);                                             \
        j_lo = max(J_ * FN20_PG, i + 1);                                      \
        j_hi = min(N, J_ * FN20_PG + FN20_PG);                                \
        const int L2_ = (j_hi - j_lo) * (s / 2);                              \
        n_units = s * L2_;                                                    \
        const double *src_ = A + (size_t)j_lo * s + (size_t)i * s * ld;       \
        FN20_UNITS(FN20_LOAD)                                                 \
    } while (0)
#define FN20_STORE(u) \
    if (tid + FN20_THREADS * u < n_units) *reinterpret_cast<double2 *>(img + q##u.off) = q##u.v;
    int t = blockIdx.x;
    if (t >= total) return;
    FN20_REQUEST(t);
    for (;;) {
        // the requested item goes to LDS (everybody is done with the image of the item before)
        FN20_UNITS(FN20_STORE)
        const int ci = i, cj_lo = j_lo, npair = j_hi - j_lo;
        __syncthreads();
        const int tn = t + (int)gridDim.x;
        if (tn < total) FN20_REQUEST(tn);   // in flight while this item is scored
        for (int p0 = 3 * wv; p0 < npair; p0 += 3 * (FN20_THREADS / 64)) {      // (wave-uniform trip count)
            const int tp = p0 + g;
            const bool live = g < 3 && tp < npair;
            const double *blk = img + (live ? tp : 0) * s;  // element (r, c) of the pair's block: blk[r + c * Lp]
            double x[s];
            double rs = 0.0, cs = 0.0;
#pragma unroll
            for (int c = 0; c < s; ++c) {
                x[c] = blk[l + c * Lp];
                rs += x[c];
            }
#pragma unroll
            for (int r = 0; r < s; ++r) cs += blk[r + l * Lp];
            const double rm = rs / (double)s, cm = cs / (double)s;
            double(*my)[s] = red[wv][g < 3 ? g : 0];
            if (live) {
                my[0][l] = rm;
                my[1][l] = cm;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            double tot = 0.0;
#pragma unroll
            for (int r = 0; r < s; ++r) tot += my[0][r];
            tot /= (double)s;  // = sum(block) / s^2
            double f = 0.0;
#pragma unroll
            for (int c = 0; c < s; ++c) {
                const double kx = x[c] - rm - my[1][c] + tot;
                f += kx * kx;
            }
            if (live) my[2][l] = f;
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (live && l == 0) {
                double ff = 0.0;
#pragma unroll
                for (int r = 0; r < s; ++r) ff += my[2][r];
                const int j = cj_lo + tp;
                const double sc = sqrt(ff);
                S[(size_t)ci + (size_t)j * N] = sc;
                S[(size_t)j + (size_t)ci * N] = sc;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // red is rewritten by the wave's next trip
        }
        if (tn >= total) break;
        t = tn;
        __syncthreads();   // the image is free
    }
#undef FN20_DECL
#undef FN20_LOAD
#undef FN20_REQUEST
#undef FN20_STORE
}

// ncu: compute units of the device (the s = 20 form is a persistent grid)
void gdca_launch_fn(hipStream_t s, const double *A, size_t ld, int N, int sdim, double *S, int ncu)
{
    gdca_fill_async(s, S, 0, (size_t)N * N * sizeof(double));
    if (N < 2) return;
    const int nJ = (N + FN_PG - 1) / FN_PG;   // row chunks (a column site's workgroups past its last chunk leave at once)
    const dim3 nwg((unsigned)nJ, (unsigned)(N - 1));
    const size_t lds = ((size_t)sdim * (FN_PG * sdim + 2) + 4 * 64) * sizeof(double);
    const int units = (sdim * (FN_PG * sdim / 2) + 255) / 256;   // s = 20: 7
    constexpr int units20 = (20 * (FN_PG * 20 / 2) + 255) / 256;
    if (sdim == 20 && (ld & 1) == 0) {
        const int total = fn20_first(N - 1, N);
        const int grid = std::min(total, FN20_WG * (ncu > 0 ? ncu : 256));
        (gdca_launch<k_fn20_args, k_fn20<1>, k_fn20<GDCA_MAXB>>(dim3((unsigned)grid), dim3(FN20_THREADS), 0, s, k_fn20_mk(A, ld, N, total, S)));
    } else if (sdim == 20)
        (gdca_launch<k_fn_args, k_fn<1, units20, 20>, k_fn<GDCA_MAXB, units20, 20>>(nwg, dim3(256), lds, s, k_fn_mk(A, ld, N, sdim, S)));
    else if (units <= 4)
        (gdca_launch<k_fn_args, k_fn<1, 4, 0>, k_fn<GDCA_MAXB, 4, 0>>(nwg, dim3(256), lds, s, k_fn_mk(A, ld, N, sdim, S)));
    else if (units <= 7)
        (gdca_launch<k_fn_args, k_fn<1, 7, 0>, k_fn<GDCA_MAXB, 7, 0>>(nwg, dim3(256), lds, s, k_fn_mk(A, ld, N, sdim, S)));
    else
        (gdca_launch<k_fn_args, k_fn<1, 16, 0>, k_fn<GDCA_MAXB, 16, 0>>(nwg, dim3(256), lds, s, k_fn_mk(A, ld, N, sdim, S)));
}

// ---- Cholesky factors of the diagonal blocks of C ------------------------------------------------------
struct k_diag_chol_args {
    const double *D;
    int sdim;
    double *Ld;
};
static inline k_diag_chol_args k_diag_chol_mk(const double *D, int sdim, double *Ld)
{
    return k_diag_chol_args{D, sdim, Ld};
}
template <int CAP>
__global__ __launch_bounds__(64) void k_diag_chol(const BatchArgs<k_diag_chol_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const double *__restrict__ D = a_.D;
    int sdim = a_.sdim;
    double *__restrict__ Ld = a_.Ld;
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
    (gdca_launch<k_diag_chol_args, k_diag_chol<1>, k_diag_chol<GDCA_MAXB>>(dim3(N), dim3(64), 0, s, k_diag_chol_mk(D, sdim, Ld)));
}

// ---- DI ---------------------------------------------------------------------------------------------
// gamma = eigenvalues of V = MM MM^T, MM = L_j^T X L_i (s x s).  Two kernels:
//   k_di_tridiag: one wave per TWO site pairs -- the three small products of each (MFMA, chained through registers), then a
//                 Householder reduction of both V to tridiagonal form side by side (4/3 s^3 flops; lane r of a half-wave owns
//                 row r, reflector vectors broadcast through LDS).  Writes the diagonal / sub-diagonal to HBM, index-major.
//   k_di_ql:      one LANE per site pair -- implicit QL (EISPACK tql1) on its tridiagonal, eigenvalues
//                 only, then DI = z + 1/2 sum log(1 + sqrt(1 + 4 gamma)).
// ~10x fewer flops than a Jacobi sweep on the full matrix, and the serial part (QL) runs 64 pairs wide.
// V = MM MM^T of one site pair into LDS (see k_di_tridiag): the three products on the matrix pipe, padded to 32 x 32 (2 x 2 tiles
// of 16 x 16, k in steps of 4), chained through registers: the MFMA operand layout (element (l15, 4 k4 + lq) in the register of
// step k4) is the accumulator layout (element (l15, lq + 4 reg) in register reg), so a product computed TRANSPOSED is the next
// product's operand as it stands.
//   T1^T(c, r) = sum_m L_i(m, c) X(r, m)              operands from memory (X in place, L_i from the factor array)
//   MM(r, c)   = sum_m L_j(m, r) T1(m, c)             a = T1^T from registers
//   V(r, r')   = sum_m MM(r, m) MM(r', m)             both operands = MM from registers
// (as 400 / 64 dot products per lane out of LDS the three of them were most of this kernel)
__device__ __forceinline__ void di_pair_products(const double *__restrict__ A, size_t ld, const double *__restrict__ Ld, int sdim,
                                                 long long pair, double *V, int lane)
{
    const int l15 = lane & 15, lq = lane >> 4;
    const int ss = sdim * sdim;
    int i, j;
    pair_decode(pair, i, j);
    const double *src = A + (size_t)j * sdim + (size_t)i * sdim * ld;   // X(r, c) = src[r + c ld]
    const double *Li = Ld + (size_t)i * ss, *Lj = Ld + (size_t)j * ss;  // lower triangular, zeros above the diagonal
    const int K4 = (sdim + 3) >> 2;
    const bool two = sdim > 16;
    typedef double double4_t __attribute__((ext_vector_type(4)));
    double4_t T1T[2][2], MMt[2][2], Vt[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) T1T[x][y] = MMt[x][y] = Vt[x][y] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) {
        if (k4 < K4) {
            const int m = 4 * k4 + lq;
            double aX[2], bL[2];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int rc = 16 * blk + l15;
                const bool ok = m < sdim && rc < sdim;
                aX[blk] = ok ? src[(size_t)rc + (size_t)m * ld] : 0.0;
                bL[blk] = ok ? Li[m + rc * sdim] : 0.0;
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
                    if (two || (cb == 0 && rb == 0)) T1T[cb][rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(aX[rb], bL[cb], T1T[cb][rb], 0, 0, 0);
        }
    }
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) {
        if (k4 < K4) {
            const int m = 4 * k4 + lq;
            double bJ[2];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int r = 16 * blk + l15;
                bJ[blk] = (m < sdim && r < sdim) ? Lj[m + r * sdim] : 0.0;
            }
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
                    if (two || (cb == 0 && rb == 0))
                        MMt[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(T1T[cb][k4 >> 2][k4 & 3], bJ[rb], MMt[rb][cb], 0, 0, 0);
        }
    }
    // V: the tiles on and below the diagonal; the others are their mirror images (V is bitwise symmetric by construction)
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) {
        if (k4 < K4) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cb = 0; cb <= rb; ++cb)
                    if (two || rb == 0)
                        Vt[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(MMt[cb][k4 >> 2][k4 & 3], MMt[rb][k4 >> 2][k4 & 3], Vt[rb][cb], 0, 0, 0);
        }
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb <= rb; ++cb)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = 16 * rb + l15, c = 16 * cb + lq + 4 * reg;
                if (r < sdim && c < sdim && r >= c) {
                    V[r + c * sdim] = Vt[rb][cb][reg];
                    V[c + r * sdim] = Vt[rb][cb][reg];
                }
            }
}

// sum over the 32 lanes of the caller's half of the wave
__device__ __forceinline__ double half_sum(double v)
{
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One wave per TWO site pairs: the products of each pair with the whole wave (the MFMA tiles are 64 lanes wide), then the
// Householder reduction of both V side by side, 32 lanes each (lane r of a half owns row r: the reduction keeps at most 31 lanes
// busy, so two pairs per wave halve its instruction count per pair).  Branch-free in the pair: a column that is already reduced
// gets the zero reflector.
struct k_di_tridiag_args {
    const double *A;
    size_t ld;
    const double *Ld;
    int sdim;
    long long npairs;
    long long tstride;
    double *Td;
    double *Te;
};
static inline k_di_tridiag_args k_di_tridiag_mk(const double *A, size_t ld, const double *Ld, int sdim, long long npairs, long long tstride, double *Td, double *Te)
{
    return k_di_tridiag_args{A, ld, Ld, sdim, npairs, tstride, Td, Te};
}
template <int CAP>
__global__ __launch_bounds__(64) void k_di_tridiag(const BatchArgs<k_di_tridiag_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const double *__restrict__ A = a_.A;
    size_t ld = a_.ld;
    const double *__restrict__ Ld = a_.Ld;
    int sdim = a_.sdim;
    long long npairs = a_.npairs;
    long long tstride = a_.tstride;
    double *__restrict__ Td = a_.Td;
    double *__restrict__ Te = a_.Te;
    extern __shared__ __attribute__((aligned(16))) double dsm[];
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int ss = sdim * sdim;
    const long long pair0 = 2 * (long long)blockIdx.x;
    di_pair_products(A, ld, Ld, sdim, pair0, dsm, lane);
    if (pair0 + 1 < npairs) di_pair_products(A, ld, Ld, sdim, pair0 + 1, dsm + ss + 64, lane);
    __syncthreads();
    const long long pair = pair0 + h;
    const bool live = pair < npairs;
    double *V = dsm + h * (ss + 64);
    double *vv = V + ss;   // [32] reflector
    double *ww = vv + 32;  // [32]
    for (int k = 0; k + 2 < sdim; ++k) {
        const int m = sdim - k - 1;                  // order of the trailing block V22 = V[k+1.., k+1..]
        const double *x = V + (k + 1) + k * sdim;    // x[t] = V(k+1+t, k)
        const double xt = (live && r < m) ? x[r] : 0.0;
        const double sigma = half_sum(xt * xt);
        const double x0 = live ? x[0] : 0.0;
        const double tail = sigma - x0 * x0;
        const bool act = tail > 0.0 && sigma > 0.0;
        const double alpha = act ? ((x0 >= 0.0) ? -sqrt(sigma) : sqrt(sigma)) : x0;  // x0: the column is already reduced
        const double v0 = x0 - alpha;
        const double beta = act ? 2.0 / (tail + v0 * v0) : 0.0;
        const double vr = (act && r < m) ? ((r == 0) ? v0 : xt) : 0.0;
        if (r < m) vv[r] = vr;
        __syncthreads();
        // p = beta V22 v: lane r sums row r
        double pr = 0.0;
        if (live && r < m) {
            const double *row = V + (k + 1 + r) + (k + 1) * sdim;
            for (int c = 0; c < m; ++c) pr += row[c * sdim] * vv[c];
        }
        pr *= beta;
        const double pv = half_sum(pr * vr);
        const double K = 0.5 * beta * pv;
        const double wr = pr - K * vr;
        if (r < m) ww[r] = wr;
        __syncthreads();
        // V22 -= v w^T + w v^T
        if (live && r < m) {
            double *row = V + (k + 1 + r) + (k + 1) * sdim;
            for (int c = 0; c < m; ++c) row[c * sdim] -= vr * ww[c] + wr * vv[c];
        }
        __syncthreads();
        if (live && r == 0) {
            Td[(size_t)k * tstride + pair] = V[k + k * sdim];
            Te[(size_t)k * tstride + pair] = alpha;
        }
    }
    if (live && r == 0) {
        if (sdim >= 2) {
            const int k = sdim - 2;
            Td[(size_t)k * tstride + pair] = V[k + k * sdim];
            Te[(size_t)k * tstride + pair] = V[(k + 1) + k * sdim];
        }
        Td[(size_t)(sdim - 1) * tstride + pair] = V[(sdim - 1) + (sdim - 1) * sdim];
        Te[(size_t)(sdim - 1) * tstride + pair] = 0.0;
    }
}

__device__ __forceinline__ double pythag(double a, double b)
{
    return sqrt(a * a + b * b);
}

struct k_di_ql_args {
    const double *Td;
    const double *Te;
    long long tstride;
    long long npairs;
    int N;
    int sdim;
    double *S;
    gdca_dev_scalars *sc;
};
static inline k_di_ql_args k_di_ql_mk(const double *Td, const double *Te, long long tstride, long long npairs, int N, int sdim, double *S, gdca_dev_scalars *sc)
{
    return k_di_ql_args{Td, Te, tstride, npairs, N, sdim, S, sc};
}
template <int CAP>
__global__ __launch_bounds__(64) void k_di_ql(const BatchArgs<k_di_ql_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const double *__restrict__ Td = a_.Td;
    const double *__restrict__ Te = a_.Te;
    long long tstride = a_.tstride;
    long long npairs = a_.npairs;
    int N = a_.N;
    int sdim = a_.sdim;
    double *__restrict__ S = a_.S;
    gdca_dev_scalars *sc = a_.sc;
    bool noconv = false;
    extern __shared__ __attribute__((aligned(16))) double qsm[];
    const int lane = threadIdx.x;
    double *d = qsm + lane;                 // d[t * 64]
    double *e = qsm + (size_t)sdim * 64 + lane;  // e[t * 64]
    const long long pair = (long long)blockIdx.x * 64 + lane;
    const bool live = pair < npairs;
    const long long pp = live ? pair : npairs - 1;
    const int n = sdim;
    // e[t] = sub-diagonal between t and t+1 (already "shifted down" in tql1's convention), e[n-1] = 0
    for (int t = 0; t < n; ++t) {
        d[t * 64] = Td[(size_t)t * tstride + pp];
        e[t * 64] = Te[(size_t)t * tstride + pp];
    }
    double f = 0.0, tst1 = 0.0;
    for (int l = 0; l < n; ++l) {
        int iter = 0;
        const double h0 = fabs(d[l * 64]) + fabs(e[l * 64]);
        if (tst1 < h0) tst1 = h0;
        int m = l;
        while (m < n - 1) {
            if (tst1 + fabs(e[m * 64]) == tst1) break;
            ++m;
        }
        if (m != l) {
            double tst2;
            do {
                if (++iter > 40) {  // tql1 gives up after 30; LAPACK's dsteqr would raise: reported, not swallowed
                    noconv = true;
                    break;
                }
                // form shift
                const int l1 = l + 1;
                double g = d[l * 64];
                double p = (d[l1 * 64] - g) / (2.0 * e[l * 64]);
                double r = pythag(p, 1.0);
                const double sr = (p >= 0.0) ? fabs(r) : -fabs(r);
                d[l * 64] = e[l * 64] / (p + sr);
                d[l1 * 64] = e[l * 64] * (p + sr);
                const double dl1 = d[l1 * 64];
                double h = g - d[l * 64];
                for (int t = l1 + 1; t < n; ++t) d[t * 64] -= h;
                f += h;
                // QL transformation
                p = d[m * 64];
                double c = 1.0, c2 = 1.0, c3 = 1.0;
                const double el1 = e[l1 * 64];
                double s = 0.0, s2 = 0.0;
                for (int t = m - 1; t >= l; --t) {
                    c3 = c2;
                    c2 = c;
                    s2 = s;
                    g = c * e[t * 64];
                    h = c * p;
                    r = pythag(p, e[t * 64]);
                    e[(t + 1) * 64] = s * r;
                    s = e[t * 64] / r;
                    c = p / r;
                    p = c * d[t * 64] - s * g;
                    d[(t + 1) * 64] = h + s * (c * g + s * d[t * 64]);
                }
                p = -s * s2 * c3 * el1 * e[l * 64] / dl1;
                e[l * 64] = s * p;
                d[l * 64] = c * p;
                tst2 = tst1 + fabs(e[l * 64]);
            } while (tst2 > tst1);
        }
        d[l * 64] = d[l * 64] + f;  // eigenvalue l (unordered)
    }
    if (live && noconv) atomicAdd(&sc->di_noconv, 1);
    if (live) {
        double acc = 0.0;
        for (int t = 0; t < n; ++t) {
            double gm = d[t * 64];
            gm = gm > 0.0 ? gm : 0.0;
            acc += log(1.0 + sqrt(1.0 + 4.0 * gm));
        }
        int i, j;
        pair_decode(pair, i, j);
        const double z = 0.5 * (double)sdim * log(0.5);
        const double v = z + 0.5 * acc;
        S[(size_t)i + (size_t)j * N] = v;
        S[(size_t)j + (size_t)i * N] = v;
    }
}

void gdca_launch_di(hipStream_t s, const double *A, size_t ld, const double *Ld, int N, int sdim, double *S,
                    double *Tws, gdca_dev_scalars *sc)
{
    gdca_fill_async(s, S, 0, (size_t)N * N * sizeof(double));
    const long long npairs = (long long)N * (N - 1) / 2;
    if (npairs <= 0) return;
    const long long tstride = (npairs + 63) / 64 * 64;
    double *Td = Tws, *Te = Tws + (size_t)sdim * tstride;
    const size_t lds1 = (size_t)2 * (sdim * sdim + 64) * sizeof(double);
    (gdca_launch<k_di_tridiag_args, k_di_tridiag<1>, k_di_tridiag<GDCA_MAXB>>(dim3((unsigned)((npairs + 1) / 2)), dim3(64), lds1, s, k_di_tridiag_mk(A, ld, Ld, sdim, npairs, tstride, Td, Te)));
    const size_t lds2 = (size_t)2 * sdim * 64 * sizeof(double);
    (gdca_launch<k_di_ql_args, k_di_ql<1>, k_di_ql<GDCA_MAXB>>(dim3((unsigned)(tstride / 64)), dim3(64), lds2, s, k_di_ql_mk(Td, Te, tstride, npairs, N, sdim, S, sc)));
}

size_t gdca_di_ws_bytes(int N, int sdim)
{
    const long long npairs = (long long)N * (N - 1) / 2;
    const long long tstride = (npairs + 63) / 64 * 64;
    return (size_t)2 * sdim * (tstride > 0 ? tstride : 64) * sizeof(double);
}

// ---- APC --------------------------------------------------------------------------------------------
struct k_colsum_args {
    const double *S;
    int N;
    double *cs;
};
static inline k_colsum_args k_colsum_mk(const double *S, int N, double *cs)
{
    return k_colsum_args{S, N, cs};
}
template <int CAP>
__global__ __launch_bounds__(256) void k_colsum(const BatchArgs<k_colsum_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const double *__restrict__ S = a_.S;
    int N = a_.N;
    double *__restrict__ cs = a_.cs;
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
struct k_apc_apply_args {
    double *S;
    int N;
    const double *cs;
};
static inline k_apc_apply_args k_apc_apply_mk(double *S, int N, const double *cs)
{
    return k_apc_apply_args{S, N, cs};
}
template <int CAP>
__global__ __launch_bounds__(256) void k_apc_apply(const BatchArgs<k_apc_apply_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    double *__restrict__ S = a_.S;
    int N = a_.N;
    const double *__restrict__ cs = a_.cs;
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
    (gdca_launch<k_colsum_args, k_colsum<1>, k_colsum<GDCA_MAXB>>(dim3(N), dim3(256), 0, s, k_colsum_mk(S, N, colsum_ws)));
    (gdca_launch<k_apc_apply_args, k_apc_apply<1>, k_apc_apply<GDCA_MAXB>>(dim3(N), dim3(256), 0, s, k_apc_apply_mk(S, N, colsum_ws)));
}
