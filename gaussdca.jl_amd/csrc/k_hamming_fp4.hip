// All-pairs Hamming reweighting (compute_weights inside DCAUtils' compute_weighted_frequencies; reference call site
// src/GaussDCA.jl:28): a LOWER BOUND of every pair's distance on the fp4 matrix pipe (round 6).
//
// Only pairs below the threshold count, so any cheap quantity that is never larger than the distance sorts the pairs: what passes
// goes to k_hamming_refine (k_hamming.hip), which counts exactly from the alignment's bytes -- the neighbour counts are the same
// integers whatever bound made the list.  Round 3's bound is the distance on the three low bit planes, d3 <= d, 4 VALU instructions
// per 32 symbol compares (k_hamming<3>: 2.39 ms at N = 500, M = 50 000, VALU issue bound).  This one is weaker and ten times cheaper:
//     D = number of BITS in which two sequences' three low planes differ;  a position that differs there differs in 1..3 bits, so
//     d >= d3 >= D / 3, and a pair can only be a neighbour if D < 3 thresh.
// D is a Gram matrix: with every bit b coded as the fp4 number (+1.0, -1.0)[b], S = sum over the 96 NW bit positions of a * b =
// (agreeing bits) - (differing bits) = K - 2 D.  v_mfma_scale_f32_32x32x64_f8f6f4 with both operands E2M1 multiplies 32 x 32 x 64 of
// them per instruction, exactly (|S| <= 3072 in f32 accumulators), at 4.3 PMAC/s out of LDS (tools/ubench_fp4.hip,
// profiles/r06_ubench_fp4.log): 1.9e12 MACs at config C instead of 6.25e11 x 4 / 32 VALU instructions.
//
// MEASURED, AND NOT CHOSEN BY THE AUTOMATIC RULE (profiles/r06_hamming_fp4.log).  The counts are exact in every case tried (the parity
// tests force this form: tests/test_gpu_parity.py), but the bound is too weak for the families gDCA sees.  theta = :auto puts the
// threshold at 0.1216 / (mean identity) -- 0.35 N for the benchmark family, whose TYPICAL pair sits at d = 0.65 N, not at N: the
// sequences of a protein family all descend from one root.  d3 loses 9 % of d, the bit count another factor 1.72 / 3: a typical pair has
// D / 3 = 0.52 d = 170 against thresh = 174, and 56 % of ALL pairs land in the list (2.5e-4 for the three-plane form, 1.7e-4 true
// neighbours); the list overflows and the exact form counts the family after all.  A bound that rejects typical pairs has to keep more
// than ~0.6 of d, i.e. needs the positions, not the bits: the class-one-hot form (8 nibbles per position, or 7 with a simplex code) is
// 5.0e12 MACs at config C = 1.2 ms at the rate the micro-benchmark sustains out of LDS, against 2.39 ms today -- and this kernel, as
// first written (one 256 x 256 tile per 1024-thread workgroup, one chunk of prefetch), reaches 1.07 PMAC/s of those 4.3 (1.77 ms for
// the 1.9e12 MACs of config C: a chunk's 0.86 us of MFMA time does not cover the ~2 us of its successor's loads, and 114 registers
// leave no room for a second chunk in flight).  The one-hot form would need this loop at three times its efficiency to win 0.8 ms of
// a 22.5-ms family: not built.  The form stays selectable (GDCA_HAMMING_MODE=mfma) for families of unrelated sequences and as the
// exactness-tested starting point of that kernel; the automatic rule chooses between the three-plane and the exact form as before.
//
// Layout.  k_fp4_image expands the three low bit planes of the bit-plane image (k_bitplane_pack) ONCE into a row-major image of
// nibbles: row = sequence, 16 bytes per (32-position word w, plane p) entry e = 3 w + p, entries padded to whole LDS chunks (eight) and
// rows to a multiple of 256 with 0.0 nibbles, which add nothing to S (a row of them: S = 0, never a candidate).  One k step of the MFMA is 64
// nibbles = two entries; a lane holds row (lane % 32) and entry (lane / 32) of the step for A and for B alike -- the sum over k does
// not care in which order the nibbles sit inside a lane as long as both operands use the same, and both come from the same image.
// k_hamming_fp4: one 1024-thread workgroup per 256 x 256 pair tile of the upper triangle (16 waves of 64 x 64 = 2 x 2 MFMA blocks),
// operands staged global -> registers -> LDS in chunks of four k steps, double-buffered, one barrier per chunk; the epilogue compares
// S with K - 6 thresh and lists the pairs (the three-plane form's list: one LDS counter, one device-wide atomic per tile).
#include <algorithm>

#include "gdca_internal.h"
#include "gdca_launch.h"

typedef int fp4x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

#define F4_TILE 256          // sequences per side of a pair tile
#define F4_KC 4              // k steps (of 64 nibbles = 32 bytes per row) per LDS chunk
#define F4_ROW (F4_KC * 32 + 16)  // bytes per row of a chunk in LDS: 128 + a 16-byte pad (16-byte reads of 32 rows: no bank is hit twice in a phase)
#define F4_OP (F4_TILE * F4_ROW)  // one operand's chunk
#define F4_LDS (4 * F4_OP)        // A and B, two buffers: 147 456 bytes

// entries per row of the image: the 3 NW real ones, padded with 0.0 nibbles (which add nothing to S) to whole LDS chunks of 2 F4_KC
static inline int f4_entries(int NW) { return (3 * NW + 2 * F4_KC - 1) / (2 * F4_KC) * (2 * F4_KC); }
size_t gdca_fp4_image_bytes(int N, int M)
{
    const size_t NW = (size_t)(N + 31) / 32, rows = ((size_t)M + F4_TILE - 1) / F4_TILE * F4_TILE;
    return rows * (size_t)f4_entries((int)NW) * 16;
}

// 8 bits -> 8 nibbles (0x2 | bit << 3: +1.0 / -1.0 in E2M1), bit i in nibble i
__device__ __forceinline__ uint32_t f4_spread8(uint32_t b)
{
    uint32_t x = b & 0xffu;
    x = (x | (x << 12)) & 0x000f000fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return (x << 3) | 0x22222222u;
}

// ---- the fp4 image of the three low bit planes -------------------------------------------------------------------------------------
struct k_fp4_image_args {
    const uint32_t *Zb;  // bit planes [Mt128][5][NW][128]
    uint4 *img;          // [rows][E] x 16 bytes
    int NW, M, E;
    const gdca_dev_scalars *sc;
};
template <int CAP>
__global__ __launch_bounds__(256) void k_fp4_image(const BatchArgs<k_fp4_image_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    if (a_.sc->ham_mode != 2) return;  // (another form was chosen for this family)
    const uint32_t *__restrict__ Zb = a_.Zb;
    uint4 *__restrict__ img = a_.img;
    const int NW = a_.NW, M = a_.M, E = a_.E;
    const int k = blockIdx.x * 256 + threadIdx.x, e = blockIdx.y;  // sequence (row of the image), entry
    const int w = e / 3, p = e - 3 * w;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);  // rows beyond the alignment and pad entries: 0.0
    if (k < M && e < 3 * NW) {
        const uint32_t word = Zb[(((size_t)(k >> 7) * 5 + p) * NW + w) * 128 + (k & 127)];
        v = make_uint4(f4_spread8(word), f4_spread8(word >> 8), f4_spread8(word >> 16), f4_spread8(word >> 24));
    }
    img[(size_t)k * E + e] = v;
}

// ---- S = X X^T by 256 x 256 tiles, thresholded ---------------------------------------------------------------------------------------
__device__ __forceinline__ void f4_tri_decode(int t, int Mt, int &I, int &J)
{
    // t in [0, Mt (Mt + 1) / 2) -> (I, J), I <= J, row-major over the upper triangle
    const double b = 2.0 * Mt + 1.0;
    int i = (int)((b - sqrt(b * b - 8.0 * (double)t)) * 0.5);
    i = max(0, min(i, Mt - 1));
    auto start = [Mt](int r) { return (long long)r * Mt - (long long)r * (r - 1) / 2; };
    while (i > 0 && start(i) > t) --i;
    while (i < Mt - 1 && start(i + 1) <= t) ++i;
    I = i;
    J = i + (int)(t - start(i));
}

struct k_hamming_fp4_args {
    const unsigned char *img;  // the image, E entries of 16 bytes per row
    int E, M, Mt, NW;          // Mt: 256-tiles per side
    gdca_dev_scalars *sc;
    int2 *cand_list;
    unsigned cand_cap;
};
__shared__ unsigned f4_tile_n, f4_tile_base;
template <int CAP>
__global__ __launch_bounds__(1024) void k_hamming_fp4(const BatchArgs<k_hamming_fp4_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    gdca_dev_scalars *__restrict__ sc = a_.sc;
    if (sc->ham_mode != 2) return;
    const int thresh = sc->thresh;
    if (thresh <= 0) return;  // theta == 0: every n_k = 1
    const unsigned char *__restrict__ img = a_.img;
    const int E = a_.E, M = a_.M, Mt = a_.Mt, NW = a_.NW;
    int2 *__restrict__ cand_list = a_.cand_list;
    const unsigned cand_cap = a_.cand_cap;
    extern __shared__ __attribute__((aligned(16))) unsigned char f4_lds[];

    int I, J;
    f4_tri_decode((int)blockIdx.x, Mt, I, J);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wr = wv & 3, wc = wv >> 2;
    const int l32 = lane & 31, lh = lane >> 5;
    if (tid == 0) f4_tile_n = 0u;
    const size_t rowbytes = (size_t)E * 16;
    const int nchunk = E / (2 * F4_KC);  // (rows hold whole chunks)

    // staging: a chunk is 2 x 256 rows x 8 pieces of 16 bytes; thread t takes pieces t, t + 1024 (operand A: tile row I) and t + 2048,
    // t + 3072 (operand B: tile row J): piece id -> row (id % 2048) / 8, piece id % 8 (eight consecutive threads: 128 contiguous bytes of
    // a row).  Named registers: as an array behind lambdas the four pieces lived in scratch.
    const int srow = tid >> 3, spc = tid & 7;  // (pieces t and t + 1024: rows srow and srow + 128)
    const unsigned char *ga = img + ((size_t)I * F4_TILE + srow) * rowbytes + spc * 16, *gb = img + ((size_t)J * F4_TILE + srow) * rowbytes + spc * 16;
    const size_t half = (size_t)128 * rowbytes;
    unsigned char *la = f4_lds + srow * F4_ROW + spc * 16;  // (+ 128 F4_ROW: the second piece; + F4_OP: operand B; + 2 F4_OP: the other buffer)
    uint4 s0, s1, s2, s3;
#define F4_LOAD(ch)                                                                       \
    do {                                                                                  \
        s0 = *reinterpret_cast<const uint4 *>(ga + (size_t)(ch) * (32 * F4_KC));          \
        s1 = *reinterpret_cast<const uint4 *>(ga + half + (size_t)(ch) * (32 * F4_KC));   \
        s2 = *reinterpret_cast<const uint4 *>(gb + (size_t)(ch) * (32 * F4_KC));          \
        s3 = *reinterpret_cast<const uint4 *>(gb + half + (size_t)(ch) * (32 * F4_KC));   \
    } while (0)
#define F4_STORE(buf)                                                                     \
    do {                                                                                  \
        unsigned char *d_ = la + (size_t)(2 * (buf)) * F4_OP;                             \
        *reinterpret_cast<uint4 *>(d_) = s0;                                              \
        *reinterpret_cast<uint4 *>(d_ + 128 * F4_ROW) = s1;                               \
        *reinterpret_cast<uint4 *>(d_ + F4_OP) = s2;                                      \
        *reinterpret_cast<uint4 *>(d_ + F4_OP + 128 * F4_ROW) = s3;                       \
    } while (0)

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int one = 0x7f7f7f7f;  // E8M0 scale 2^0 for both operands

    // this lane's rows of the two operands inside a chunk: row (lane % 32) of its wave's first block, entry (lane / 32) of a k step
    const unsigned char *abase = f4_lds + (wr * 64 + l32) * F4_ROW + lh * 16, *bbase = f4_lds + F4_OP + (wc * 64 + l32) * F4_ROW + lh * 16;
    F4_LOAD(0);
    F4_STORE(0);
    __syncthreads();
    for (int ch = 0; ch < nchunk; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunk) F4_LOAD(ch + 1);  // in flight under this chunk's MFMAs
        const unsigned char *As = abase + (size_t)(2 * buf) * F4_OP, *Bs = bbase + (size_t)(2 * buf) * F4_OP;
        // (one k step at a time: unrolled, the compiler hoists all sixteen 16-byte reads of the chunk in front of the first MFMA and the
        // kernel no longer fits the 128 registers a 1024-thread workgroup has per lane; four waves per SIMD cover a step's read latency)
#pragma unroll 1
        for (int ks = 0; ks < F4_KC; ++ks) {
            fp4x8_t a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint4 x = *reinterpret_cast<const uint4 *>(As + i * 32 * F4_ROW + ks * 32);
                a[i] = (fp4x8_t){(int)x.x, (int)x.y, (int)x.z, (int)x.w, 0, 0, 0, 0};
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint4 x = *reinterpret_cast<const uint4 *>(Bs + j * 32 * F4_ROW + ks * 32);
                b[j] = (fp4x8_t){(int)x.x, (int)x.y, (int)x.z, (int)x.w, 0, 0, 0, 0};
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[j], acc[i][j], 4, 4, 0, one, 0, one);
        }
        if (ch + 1 < nchunk) F4_STORE(buf ^ 1);  // (the other buffer: its readers passed the barrier of the chunk before)
        __syncthreads();
    }

    // ---- candidates: S > K - 6 thresh  <=>  D = (K - S) / 2 < 3 thresh ----
    // accumulator element e of block (i, j): row 8 (e / 4) + 4 (lane / 32) + e % 4, column lane % 32
    const float limit = (float)(96 * NW - 6 * thresh);  // K = the 96 NW real bit positions (pad bits of the last word agree: both 0)
    const bool diag = I == J;
    unsigned mine = 0;
    unsigned long long cand[2] = {0ull, 0ull};  // bit 16 j + e of cand[i]: element e of block (i, j)
    // (most lanes of most tiles hold nothing above the limit: one maximum over the 64 accumulators decides that)
    float mx = acc[0][0][0];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, acc[i][j][e]);
    if (mx > limit) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int gc = J * F4_TILE + wc * 64 + j * 32 + l32;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int gr = I * F4_TILE + wr * 64 + i * 32 + 8 * (e >> 2) + 4 * lh + (e & 3);
                    if (acc[i][j][e] > limit && gr < M && gc < M && (diag ? gr < gc : true)) {
                        cand[i] |= 1ull << (16 * j + e);
                        ++mine;
                    }
                }
            }
    }
    const unsigned off = mine ? atomicAdd(&f4_tile_n, mine) : 0u;
    __syncthreads();
    if (f4_tile_n == 0u) return;  // (uniform)
    if (tid == 0) {
        const unsigned long long base = atomicAdd(&sc->ham_ncand, (unsigned long long)f4_tile_n);
        f4_tile_base = base > (unsigned long long)cand_cap ? cand_cap : (unsigned)base;  // (beyond the capacity nothing is written)
    }
    __syncthreads();
    unsigned slot = f4_tile_base + off;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        unsigned long long c = cand[i];
        while (c) {
            const int bit = __builtin_ctzll(c), j = bit >> 4, e = bit & 15;
            const int gr = I * F4_TILE + wr * 64 + i * 32 + 8 * (e >> 2) + 4 * lh + (e & 3);
            const int gc = J * F4_TILE + wc * 64 + j * 32 + l32;
            if (slot < cand_cap) cand_list[slot] = make_int2(gr, gc);
            ++slot;
            c &= c - 1;
        }
    }
}

// ---- how many pairs would this form list?  (a sample of 128 x 128 tiles, as k_hamming<3, PROBE> takes for the three-plane form) ------
// D counted with plain popcounts from the bit planes: 192 tiles, microseconds.  sc->ham_cand2 += pairs of the sampled tiles with D < 3 thresh.
struct k_fp4_probe_args {
    const uint32_t *Zb;
    int NW, M, Mt;  // Mt: 128-tiles per side
    gdca_dev_scalars *sc;
};
template <int CAP>
__global__ __launch_bounds__(256) void k_fp4_probe(const BatchArgs<k_fp4_probe_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const uint32_t *__restrict__ Zb = a_.Zb;
    const int NW = a_.NW, M = a_.M, Mt = a_.Mt;
    gdca_dev_scalars *sc = a_.sc;
    const int thresh = sc->thresh;
    if (thresh <= 0) return;
    __shared__ int total;
    int I, J;
    // (gridDim.x tiles spread evenly over the upper triangle's Mt (Mt + 1) / 2, the very tiles the three-plane probe samples)
    f4_tri_decode((int)(((long long)blockIdx.x * ((long long)Mt * (Mt + 1) / 2)) / gridDim.x), Mt, I, J);
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    if (tid == 0) total = 0;
    __syncthreads();
    int D[8][8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 8; ++c) D[r][c] = 0;
    const uint32_t *Ag = Zb + (size_t)I * 5 * NW * 128, *Bg = Zb + (size_t)J * 5 * NW * 128;
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            uint32_t a[8], b[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) a[r] = Ag[((size_t)p * NW + w) * 128 + ty * 8 + r];
#pragma unroll
            for (int c = 0; c < 8; ++c) b[c] = Bg[((size_t)p * NW + w) * 128 + tx * 8 + c];
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c) D[r][c] += __builtin_popcount(a[r] ^ b[c]);
        }
    int cand = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int gr = I * 128 + ty * 8 + r, gc = J * 128 + tx * 8 + c;
            cand += (gr < M) && (gc < M) && (gr != gc) && (D[r][c] < 3 * thresh);
        }
    if (cand) atomicAdd(&total, cand);
    __syncthreads();
    if (tid == 0 && total) atomicAdd(&sc->ham_cand2, total);
}

void gdca_launch_hamming_fp4_probe(hipStream_t s, const uint32_t *Zb, int N, int M, int nprobe, gdca_dev_scalars *sc)
{
    const int NW = (N + 31) / 32, Mt = (M + 127) / 128;
    gdca_launch<k_fp4_probe_args, k_fp4_probe<1>, k_fp4_probe<GDCA_MAXB>>(dim3((unsigned)nprobe), dim3(256), 0, s, k_fp4_probe_args{Zb, NW, M, Mt, sc});
}

// Enqueues the image build and the tile products; both leave at once unless k_hamming_decide chose this form (sc->ham_mode == 2).
// img: gdca_fp4_image_bytes(N, M) of scratch; cand_list / cap: the list k_hamming_refine consumes.
void gdca_launch_hamming_fp4(hipStream_t s, const uint32_t *Zb, void *img, int N, int M, gdca_dev_scalars *sc, void *cand_list, unsigned cap)
{
    const int NW = (N + 31) / 32, E = f4_entries(NW), Mt = (M + F4_TILE - 1) / F4_TILE;
    const long long ntile = (long long)Mt * (Mt + 1) / 2;
    gdca_launch<k_fp4_image_args, k_fp4_image<1>, k_fp4_image<GDCA_MAXB>>(dim3((unsigned)Mt, (unsigned)E), dim3(256), 0, s,
                                                                           k_fp4_image_args{Zb, (uint4 *)img, NW, M, E, sc});
    gdca_launch<k_hamming_fp4_args, k_hamming_fp4<1>, k_hamming_fp4<GDCA_MAXB>>(dim3((unsigned)ntile), dim3(1024), F4_LDS, s,
                                                                                 k_hamming_fp4_args{(const unsigned char *)img, E, M, Mt, NW, sc, (int2 *)cand_list, cap});
}
