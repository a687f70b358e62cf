// All-pairs Hamming reweighting  (DCAUtils compute_weights; reference call site
// src/GaussDCA.jl:28):  n_k = 1 + #{l != k : Hamming(Z[:,k], Z[:,l]) < floor(theta N)},
// W_k = 1/n_k, Meff = sum_k W_k.
//
// MI355X design.  The reference packs 5-bit symbols into UInt64 words and XOR/popcounts them.
// Here the alignment is BIT-SLICED instead: for every sequence, plane p (p = 0..4) holds bit p
// of 32 consecutive positions in one dword.  Two sequences differ at a position iff any plane
// differs, so 32 symbol compares cost  5 x (xor|or as one v_bitop3) + 1 x v_bcnt(+acc)  = 6
// VALU ops -- 0.19 op per compare against >= 0.75 for byte-wise compares.  Everything is
// integer and exact; the result does not depend on any summation order.
//
// Tiling: a workgroup owns a 128 x 128 block of sequence pairs (upper-triangular tile
// schedule, symmetric pairs visited once), stages both sides' planes through LDS in chunks
// of 8 dwords (256 positions) with 16-byte coalesced loads, and every thread keeps an 8 x 8
// pair micro-tile of distances in registers.  Neighbour counts leave the workgroup as one
// integer atomic per sequence and tile (wave-contiguous addresses).
//
// Not HBM-bound: the bit-plane image (N*M*5/8 bytes, 16 MB at N=500, M=50k) stays in L2 /
// Infinity Cache and every tile is re-read M/128 times; the bound is VALU issue.
#include <algorithm>
#include <cstdlib>

#include "gdca_internal.h"
#include "gdca_launch.h"

#define NPLANES 5
#define WCHUNK 8

size_t gdca_bitplane_bytes(int N, int M)
{
    const size_t Mt = (M + GDCA_HTILE - 1) / GDCA_HTILE, NW = (N + 31) / 32;
    return Mt * NPLANES * NW * GDCA_HTILE * sizeof(uint32_t);
}

// ---- Z [M][N] bytes -> bit planes [Mt][5][NW][128] -------------------------------------------
// thread <-> sequence, blockIdx.y <-> dword of 32 positions: 128 contiguous dwords per store.
struct k_bitplane_pack_args {
    const int8_t *Z;
    uint32_t *Zb;
    int N;
    int M;
    int NW;
    int q;
    gdca_dev_scalars *sc;
};
static inline k_bitplane_pack_args k_bitplane_pack_mk(const int8_t *Z, uint32_t *Zb, int N, int M, int NW, int q, gdca_dev_scalars *sc)
{
    return k_bitplane_pack_args{Z, Zb, N, M, NW, q, sc};
}
template <int CAP>
__global__ __launch_bounds__(128) void k_bitplane_pack(const BatchArgs<k_bitplane_pack_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const int8_t *__restrict__ Z = a_.Z;
    uint32_t *__restrict__ Zb = a_.Zb;
    int N = a_.N;
    int M = a_.M;
    int NW = a_.NW;
    int q = a_.q;
    gdca_dev_scalars *sc = a_.sc;
    uint32_t bad = 0;  // any byte outside 1..q (checked here because every weights computation passes through)
    const uint32_t over = (0x7fu - (uint32_t)q) * 0x01010101u;
    const int tile = blockIdx.x, w = blockIdx.y, kl = threadIdx.x;
    const int k = tile * GDCA_HTILE + kl;
    uint32_t pl[NPLANES];
#pragma unroll
    for (int p = 0; p < NPLANES; ++p) pl[p] = 0;
    if (k < M) {
        const int8_t *src = Z + (size_t)k * N + (size_t)w * 32;
        const int nb = min(32, N - w * 32);
        if (nb == 32 && (N & 3) == 0) {
            // aligned fast path: 8 dword loads; bit p of 4 bytes -> 4 adjacent bits by one multiply
            const uint32_t *s4 = reinterpret_cast<const uint32_t *>(src);
            uint32_t d[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = s4[j];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bad |= (d[j] - 0x01010101u) & ~d[j] & 0x80808080u;                      // a zero byte
                bad |= (((d[j] & 0x7f7f7f7fu) + over) | d[j]) & 0x80808080u;           // a byte > q
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int p = 0; p < NPLANES; ++p) {
                    const uint32_t t = (d[j] >> p) & 0x01010101u;
                    pl[p] |= (((t * 0x01020408u) >> 24) & 0xFu) << (4 * j);
                }
        } else {
            for (int b = 0; b < nb; ++b) {
                const uint32_t raw = (uint32_t)(uint8_t)src[b];
                bad |= (raw - 1u) >= (uint32_t)q;
                const uint32_t z = raw & 31u;
#pragma unroll
                for (int p = 0; p < NPLANES; ++p) pl[p] |= ((z >> p) & 1u) << b;
            }
        }
    }
#pragma unroll
    for (int p = 0; p < NPLANES; ++p)
        Zb[(((size_t)tile * NPLANES + p) * NW + w) * GDCA_HTILE + kl] = pl[p];
    if (bad) atomicOr(&sc->bad_symbol, 1);
}

void gdca_launch_bitplane_pack(hipStream_t s, const int8_t *Z, uint32_t *Zb, int N, int M, int q, gdca_dev_scalars *sc)
{
    const int Mt = (M + GDCA_HTILE - 1) / GDCA_HTILE, NW = (N + 31) / 32;
    (gdca_launch<k_bitplane_pack_args, k_bitplane_pack<1>, k_bitplane_pack<GDCA_MAXB>>(dim3(Mt, NW), dim3(128), 0, s, k_bitplane_pack_mk(Z, Zb, N, M, NW, q, sc)));
}

// ---- all-pairs distances, thresholded neighbour counts ----------------------------------------
__device__ __forceinline__ void tri_decode(int t, int Mt, int &I, int &J)
{
    // t in [0, Mt(Mt+1)/2) -> (I, J), I <= J, row-major over the upper triangle
    const double b = 2.0 * Mt + 1.0;
    int i = (int)((b - sqrt(b * b - 8.0 * (double)t)) * 0.5);
    if (i < 0) i = 0;
    if (i > Mt - 1) i = Mt - 1;
    auto start = [Mt](int r) { return (long long)r * Mt - (long long)r * (r - 1) / 2; };
    while (i > 0 && start(i) > t) --i;
    while (i < Mt - 1 && start(i + 1) <= t) ++i;
    I = i;
    J = i + (int)(t - start(i));
}

// NP = planes compared in the main loop.  NP = 5: exact distances (6 VALU instructions per 32 symbol compares).
// NP = 3 (round 3): a LOWER BOUND first -- two symbols that differ in their three low bits differ, so the distance on planes
// 0..2 alone, d3 <= d, costs 4 instructions per 32 compares; only pairs with d3 < threshold can be neighbours.  Where most
// sequences are unrelated those are a few in ten thousand (d3 of two unrelated sequences is ~0.9 d, far beyond the threshold;
// two planes would save more instructions but leave 0.8 % of the benchmark family's pairs to refine: measured, 3.6x slower).
// The candidates of a tile go into a list in LDS and are then counted exactly from all five planes of the global image, 16
// lanes per pair (one per dword of 32 positions).  The counts are the same integers either way.  Which form pays depends on
// the family: k_hamming<3, PROBE> measures the candidate density on a sample of tiles and k_hamming_decide sets sc->ham_mode;
// both kernels are launched and the one not chosen returns at once.
#define HAM_BOUND_PLANES 3
#define HAM_CAND_CAP 1024  // candidate pairs of one tile per pass of the refinement

// "does any thread of the (one-dimensional, 256-thread) workgroup say yes?" with ONE barrier: call number k (the same in every thread, counted
// from 0) collects into slot k % 3 and clears slot (k + 1) % 3 for the next call before its barrier -- the readers of that slot (call
// k - 2) are all past the barrier of call k - 1.  hm_any[0 .. 1] are zeroed at the start of the kernel.  (__syncthreads_or builds a
// three-dimensional thread index for its own LDS slot: its y and z parts were hoisted out of the chunk loop and kept in scratch.)
__shared__ int hm_any[3];
__shared__ unsigned hm_tile_n, hm_tile_base;  // the bound form: candidates of the tile, and where its range of the list starts
__device__ __forceinline__ bool wg_any(bool pred, int &k)
{
    const int s = k % 3, n = (k + 1) % 3;
    ++k;
    if (threadIdx.x == 0) hm_any[n] = 0;
    if (__any(pred) && (threadIdx.x & 63) == 0) atomicOr(&hm_any[s], 1);
    __syncthreads();
    return hm_any[s] != 0;
}

// ---- the bound form's candidates ----------------------------------------------------------------------------------------------
// k_hamming<3, false> does not count anything: every in-range pair of a tile whose three-plane distance is below the threshold goes,
// as (k, l), k != l, into ONE list in HBM (a slot range per thread off a device-wide counter, sc->ham_ncand), and k_hamming_refine
// counts the listed pairs exactly from the alignment's bytes -- 16 lanes per pair, 64 contiguous bytes of either sequence per
// trip -- with one integer atomic per end of a neighbour pair.  (Round 3 refined inside the tile's workgroup, from the five bit
// planes: 16 scattered dwords per lane and plane, and the refinement's loops shared the main loop's registers -- 36 of them parked
// in scratch on EVERY tile, 2.2 GB written per launch at N = 500, M = 50 000 for 200 KB of results.)  The list holds HAM_CAND_PER_TILE
// pairs per tile on average -- sixteen times the density at which the bound form is chosen at all; a family that overflows it
// anyway (clustered where the sampled tiles were not) is counted by the exact form instead: nothing is ever dropped.
#define HAM_CAND_PER_TILE 16

size_t gdca_hamming_cand_cap(int M)
{
    const long long Mt = (M + GDCA_HTILE - 1) / GDCA_HTILE;
    // (the list's counter and slots are 32-bit: beyond 2^30 pairs -- M > ~1.4 million -- the list is simply no longer than that, and a
    // family that fills it is counted by the exact form like any other overflow)
    return (size_t)std::min<long long>(Mt * (Mt + 1) / 2 * HAM_CAND_PER_TILE + 4096, 1ll << 30);
}

struct k_hamming_refine_args {
    const int8_t *Z;
    const int2 *list;
    unsigned cap;
    int N;
    int32_t *cnt;
    const gdca_dev_scalars *sc;
};
static inline k_hamming_refine_args k_hamming_refine_mk(const int8_t *Z, const int2 *list, unsigned cap, int N, int32_t *cnt, const gdca_dev_scalars *sc)
{
    return k_hamming_refine_args{Z, list, cap, N, cnt, sc};
}
template <int CAP>
__global__ __launch_bounds__(256) void k_hamming_refine(const BatchArgs<k_hamming_refine_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const int8_t *__restrict__ Z = a_.Z;
    const int2 *__restrict__ list = a_.list;
    unsigned cap = a_.cap;
    int N = a_.N;
    int32_t *__restrict__ cnt = a_.cnt;
    const gdca_dev_scalars *__restrict__ sc = a_.sc;
    if (sc->ham_mode == 0) return;  // (the exact form counts this family; 1, 2: one of the bound forms made the list)
    if (sc->ham_ncand > (unsigned long long)cap) return;  // the list overflowed: the exact form counts this family
    const unsigned total = (unsigned)sc->ham_ncand;
    const int thresh = sc->thresh;
    const int tid = threadIdx.x, sub = tid & 15, grp = (tid & 63) >> 4;
    const unsigned wave = blockIdx.x * 4u + (unsigned)(tid >> 6), nwave = gridDim.x * 4u;
    for (unsigned eb = wave * 4u; eb < total; eb += nwave * 4u) {  // (a wave's trip count is uniform: the shuffles need all lanes)
        const unsigned e = eb + (unsigned)grp;
        uint32_t d = 0;
        int k = 0, l = 0;
        if (e < total) {
            const int2 pr = list[e];
            k = pr.x;
            l = pr.y;
            const int8_t *zk = Z + (size_t)k * N, *zl = Z + (size_t)l * N;
            if ((N & 3) == 0 && (reinterpret_cast<uintptr_t>(Z) & 3) == 0) {  // (Z is the caller's pointer: a view need not be dword-aligned)
                // a lane's dwords of both sequences, 1024 bytes of each per round, ALL requested before the first is looked at: the pairs
                // of a list are scattered over the alignment, and a round costs one trip to memory instead of sixteen dependent ones
                const uint32_t *a = reinterpret_cast<const uint32_t *>(zk), *b = reinterpret_cast<const uint32_t *>(zl);
                const int nw = N >> 2;
                for (int w0 = 0; w0 < nw; w0 += 256) {
                    uint32_t xa[16], xb[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int w = w0 + 16 * u + sub;
                        xa[u] = w < nw ? a[w] : 0u;
                        xb[u] = w < nw ? b[w] : 0u;
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        uint32_t x = xa[u] ^ xb[u];  // bytes that differ -> one bit each
                        x |= x >> 4;
                        x |= x >> 2;
                        x |= x >> 1;
                        d += __builtin_popcount(x & 0x01010101u);
                    }
                }
            } else {
                for (int i = sub; i < N; i += 16) d += zk[i] != zl[i];
            }
        }
        d += __shfl_xor(d, 1);
        d += __shfl_xor(d, 2);
        d += __shfl_xor(d, 4);
        d += __shfl_xor(d, 8);
        if (e < total && sub == 0 && (int)d < thresh) {
            atomicAdd(&cnt[k], 1);
            atomicAdd(&cnt[l], 1);
        }
    }
}

struct k_hamming_args {
    const uint32_t *Zb;
    int32_t *cnt;
    int NW;
    int M;
    int Mt;
    gdca_dev_scalars *sc;
    int2 *cand_list;
    unsigned cand_cap;
};
static inline k_hamming_args k_hamming_mk(const uint32_t *Zb, int32_t *cnt, int NW, int M, int Mt, gdca_dev_scalars *sc, int2 *cand_list, unsigned cand_cap)
{
    return k_hamming_args{Zb, cnt, NW, M, Mt, sc, cand_list, cand_cap};
}
template <int CAP, int NP, bool PROBE>
__global__ __launch_bounds__(256, 3) void k_hamming(const BatchArgs<k_hamming_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const uint32_t *__restrict__ Zb = a_.Zb;
    int32_t *__restrict__ cnt = a_.cnt;
    int NW = a_.NW;
    int M = a_.M;
    int Mt = a_.Mt;
    gdca_dev_scalars *__restrict__ sc = a_.sc;
    int2 *__restrict__ cand_list = a_.cand_list;
    unsigned cand_cap = a_.cand_cap;
    const int thresh = sc->thresh;
    if (thresh <= 0) return;  // theta == 0 (or floor(theta N) == 0): every n_k = 1
    if constexpr (!PROBE) {
        // the other form was chosen for this family -- unless the bound form's candidate list overflowed: then the exact form counts
        const bool mine = NP == NPLANES ? (sc->ham_mode == 0 || sc->ham_ncand > cand_cap) : sc->ham_mode == 1;
        if (!mine) return;
    }

    __shared__ __attribute__((aligned(16))) uint32_t As[NP][WCHUNK][GDCA_HTILE];
    __shared__ __attribute__((aligned(16))) uint32_t Bs[NP][WCHUNK][GDCA_HTILE];
    __shared__ int rc[GDCA_HTILE], cc[GDCA_HTILE];

    int I, J;
    // (PROBE: gridDim.x tiles spread evenly over the upper triangle's Mt (Mt + 1) / 2)
    tri_decode(PROBE ? (int)(((long long)blockIdx.x * ((long long)Mt * (Mt + 1) / 2)) / gridDim.x) : (int)blockIdx.x, Mt, I, J);
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    if (tid < GDCA_HTILE) {
        rc[tid] = 0;
        cc[tid] = 0;
    }
    if (tid == 0) {
        hm_any[0] = 0;
        hm_any[1] = 0;
        hm_tile_n = 0u;
    }
    int any_calls = 0;  // (wg_any: calls so far)

    uint32_t acc[8][8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[r][c] = 0;
    bool all_beyond = false;  // the main loop ended early: every pair of the tile is at or beyond the threshold

    const uint32_t *Ag = Zb + (size_t)I * NPLANES * NW * GDCA_HTILE;
    const uint32_t *Bg = Zb + (size_t)J * NPLANES * NW * GDCA_HTILE;

    for (int w0 = 0; w0 < NW; w0 += WCHUNK) {
        __syncthreads();
        // stage: per plane a contiguous run of WCHUNK*128 dwords = 4 KB = 256 threads x 16 B
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int wl = tid >> 5;  // dword row inside the chunk (32 threads x 16 B = 128 dwords)
            uint4 va = make_uint4(0, 0, 0, 0), vb = make_uint4(0, 0, 0, 0);
            if (w0 + wl < NW) {
                const size_t off = ((size_t)p * NW + w0 + wl) * GDCA_HTILE + (size_t)(tid & 31) * 4;
                va = *reinterpret_cast<const uint4 *>(Ag + off);
                vb = *reinterpret_cast<const uint4 *>(Bg + off);
            }
            *reinterpret_cast<uint4 *>(&As[p][wl][(tid & 31) * 4]) = va;
            *reinterpret_cast<uint4 *>(&Bs[p][wl][(tid & 31) * 4]) = vb;
        }
        __syncthreads();
        const int wn = min(WCHUNK, NW - w0);
        for (int w = 0; w < wn; ++w) {
            uint32_t a[NP][8], b[NP][8];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const uint4 a0 = *reinterpret_cast<const uint4 *>(&As[p][w][ty * 8]);
                const uint4 a1 = *reinterpret_cast<const uint4 *>(&As[p][w][ty * 8 + 4]);
                const uint4 b0 = *reinterpret_cast<const uint4 *>(&Bs[p][w][tx * 4]);
                const uint4 b1 = *reinterpret_cast<const uint4 *>(&Bs[p][w][64 + tx * 4]);
                a[p][0] = a0.x; a[p][1] = a0.y; a[p][2] = a0.z; a[p][3] = a0.w;
                a[p][4] = a1.x; a[p][5] = a1.y; a[p][6] = a1.z; a[p][7] = a1.w;
                b[p][0] = b0.x; b[p][1] = b0.y; b[p][2] = b0.z; b[p][3] = b0.w;
                b[p][4] = b1.x; b[p][5] = b1.y; b[p][6] = b1.z; b[p][7] = b1.w;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    uint32_t x = a[0][r] ^ b[0][c];
#pragma unroll
                    for (int p = 1; p < NP; ++p)  // x |= a ^ b as one v_bitop3_b32 (table 0xBE)
                        x = __builtin_amdgcn_bitop3_b32(a[p][r], b[p][c], x, 0xBE);
                    acc[r][c] += __builtin_popcount(x);
                }
        }
        // early exit: distances only grow, so once every pair of the tile is at or beyond the
        // threshold no later position can make it a neighbour
        if (w0 + WCHUNK < NW) {
            uint32_t mn = acc[0][0];
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c) mn = min(mn, acc[r][c]);
            if (!wg_any((int)mn < thresh, any_calls)) {
                all_beyond = true;  // (uniform over the workgroup)
                break;
            }
        }
    }

    const bool diag = (I == J);
    if constexpr (PROBE) {
        // how many pairs of this tile would have to be refined
        int cand = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int lc = (c < 4) ? (tx * 4 + c) : (64 + tx * 4 + (c - 4));
                const int gr = I * GDCA_HTILE + ty * 8 + r, gc = J * GDCA_HTILE + lc;
                cand += (gr < M) && (gc < M) && (gr != gc) && ((int)acc[r][c] < thresh);
            }
        if (cand) atomicAdd(&rc[0], cand);
        __syncthreads();
        if (tid == 0 && rc[0]) atomicAdd(&sc->ham_cand, rc[0]);
        return;
    }
    if (all_beyond) return;  // nothing of this tile counts (the usual end of a tile of unrelated sequences): no list, no atomics
    if constexpr (NP < NPLANES) {
        // candidates -> the list (the bound can only be too small, so nothing else can be a neighbour).  A diagonal tile holds every
        // pair twice: its upper half is listed.  (Most threads hold no candidate at all: 63 minima decide that.)
        uint32_t mn = acc[0][0];
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c) mn = min(mn, acc[r][c]);
        unsigned long long cand = 0ull;  // bit 8 r + c: pair (r, c) of this thread's micro-tile
        if ((int)mn < thresh) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int lc = (c < 4) ? (tx * 4 + c) : (64 + tx * 4 + (c - 4));
                    const int gr = I * GDCA_HTILE + ty * 8 + r, gc = J * GDCA_HTILE + lc;
                    if ((int)acc[r][c] < thresh && gr < M && gc < M && (diag ? gr < gc : true)) cand |= 1ull << (8 * r + c);
                }
        }
        if (!wg_any(cand != 0ull, any_calls)) return;  // (uniform)
        // slots: a thread's offset inside the tile's range from a counter in LDS, ONE device-wide atomic per tile (per candidate-holding
        // thread, a family of close relatives -- tens of thousands of neighbour pairs -- queued up on that one address)
        const unsigned mine = (unsigned)__builtin_popcountll(cand);
        const unsigned off = mine ? atomicAdd(&hm_tile_n, mine) : 0u;
        __syncthreads();
        // (a 64-bit counter: a dense family of M > 92 000 has more than 2^32 candidate pairs, and a 32-bit count of them could wrap back
        // below the capacity.  ONE returning atomic per tile -- a load of the counter in front of it, to stop adding once it is past the
        // capacity, cost config C 1.2 ms: loads of a line that is busy with atomics queue up behind them)
        if (tid == 0) {
            const unsigned long long base = atomicAdd(&sc->ham_ncand, (unsigned long long)hm_tile_n);
            hm_tile_base = base > (unsigned long long)cand_cap ? cand_cap : (unsigned)base;  // (beyond the capacity nothing is written)
        }
        __syncthreads();
        unsigned slot = hm_tile_base + off;
        while (cand) {
            const int e = __builtin_ctzll(cand), r = e >> 3, c = e & 7;
            const int lc = (c < 4) ? (tx * 4 + c) : (64 + tx * 4 + (c - 4));
            if (slot < cand_cap) cand_list[slot] = make_int2(I * GDCA_HTILE + ty * 8 + r, J * GDCA_HTILE + lc);
            ++slot;
            cand &= cand - 1;
        }
        return;
    }

    // The exact form: threshold, count (strict '<'), reduce over the workgroup.  (The bound form has nothing left to count: every
    // in-range pair whose bound is below the threshold went to the refinement above, which did the counting -- and as long as this
    // block was compiled for it too, its never-true `acc < thresh && !candidate` kept the 64 accumulators alive across the
    // refinement loop: 78 scratch stores and 74 loads per lane and tile, found with tools/kernel_resources.py.)
    if constexpr (NP == NPLANES) {
        int rowc[8], colc[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) rowc[r] = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) colc[c] = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int gr = I * GDCA_HTILE + ty * 8 + r;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int lc = (c < 4) ? (tx * 4 + c) : (64 + tx * 4 + (c - 4));
                const int gc = J * GDCA_HTILE + lc;
                const bool ok = (gr < M) && (gc < M) && (gr != gc) && ((int)acc[r][c] < thresh);
                rowc[r] += ok ? 1 : 0;
                colc[c] += ok ? 1 : 0;
            }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (rowc[r]) atomicAdd(&rc[ty * 8 + r], rowc[r]);
        if (!diag) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int lc = (c < 4) ? (tx * 4 + c) : (64 + tx * 4 + (c - 4));
                if (colc[c]) atomicAdd(&cc[lc], colc[c]);
            }
        }
    }
    __syncthreads();
    if (tid < GDCA_HTILE) {
        const int v = rc[tid];
        if (v) atomicAdd(&cnt[I * GDCA_HTILE + tid], v);
    } else if (!diag) {
        const int v = cc[tid - GDCA_HTILE];
        if (v) atomicAdd(&cnt[J * GDCA_HTILE + tid - GDCA_HTILE], v);
    }
}

// sc->ham_mode from the sample: 1 (lower bound first) if fewer than 1 pair in 1000 of the sampled tiles is a candidate -- beyond
// that the refinement costs more than the two instructions per word the bound saves
struct k_hamming_decide_args {
    gdca_dev_scalars *sc;
    long long sampled_pairs;
    int force;
    int fp4_ok;  // the fp4 form may be chosen (its image buffer exists)
};
static inline k_hamming_decide_args k_hamming_decide_mk(gdca_dev_scalars *sc, long long sampled_pairs, int force, int fp4_ok)
{
    return k_hamming_decide_args{sc, sampled_pairs, force, fp4_ok};
}
template <int CAP>
__global__ void k_hamming_decide(const BatchArgs<k_hamming_decide_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    gdca_dev_scalars *sc = a_.sc;
    long long sampled_pairs = a_.sampled_pairs;
    int force = a_.force;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        // the bit-count bound on the fp4 matrix pipe (k_hamming_fp4.hip) where it would list fewer than 2 pairs in 1000 (its product is
        // five times cheaper than the three-plane form's loop, its list about twice as long); else the three-plane form below 1 in 1000
        // (a pair refined costs ~50 pairs skipped); else exact distances
        const double pairs = (double)sampled_pairs;
        sc->ham_mode = force >= 0 ? force : ((a_.fp4_ok && (double)sc->ham_cand2 < 2e-3 * pairs) ? 2 : ((double)sc->ham_cand < 1e-3 * pairs ? 1 : 0));
        sc->ham_ncand = 0u;
    }
}

// Z: the alignment's bytes ([M][N], what the bit planes were packed from); cand_list: gdca_hamming_cand_cap(M) pairs of scratch;
// fp4_img: gdca_fp4_image_bytes(N, M) of scratch for the fp4 form, or nullptr (that form is then never chosen)
void gdca_launch_hamming(hipStream_t s, const uint32_t *Zb, const int8_t *Z, int32_t *cnt, int N, int M, gdca_dev_scalars *sc, int force,
                         void *cand_list, void *fp4_img)
{
    const int Mt = (M + GDCA_HTILE - 1) / GDCA_HTILE, NW = (N + 31) / 32;
    const long long ntile = (long long)Mt * (Mt + 1) / 2;
    int2 *list = (int2 *)cand_list;
    const unsigned cap = (unsigned)gdca_hamming_cand_cap(M);
    // the context option GDCA_HAMMING_MODE=full|bound|mfma forces a form (tests, measurements); default: decided per family from a
    // sample of tiles
    const int nprobe = (int)std::min<long long>(ntile, 192);
    if (force == 2 && !fp4_img) force = 1;
    // (alignments of at most 64 columns -- two words per plane -- go to the exact form unprobed: the bound saves two instructions per
    // word and costs a list and a second kernel; N = 64, M = 20 000: 0.136 ms against 0.200, profiles/r05_option_probes.log)
    if (force < 0 && NW <= 2) force = 0;
    if (force < 0 && ntile >= 64) {
        (gdca_launch<k_hamming_args, k_hamming<1, HAM_BOUND_PLANES, true>, k_hamming<GDCA_MAXB, HAM_BOUND_PLANES, true>>(dim3((unsigned)nprobe), dim3(256), 0, s, k_hamming_mk(Zb, cnt, NW, M, Mt, sc, list, cap)));
        if (fp4_img) gdca_launch_hamming_fp4_probe(s, Zb, N, M, nprobe, sc);
        (gdca_launch<k_hamming_decide_args, k_hamming_decide<1>, k_hamming_decide<GDCA_MAXB>>(dim3(1), dim3(1), 0, s, k_hamming_decide_mk(sc, (long long)nprobe * GDCA_HTILE * GDCA_HTILE, -1, fp4_img ? 1 : 0)));
    } else {
        (gdca_launch<k_hamming_decide_args, k_hamming_decide<1>, k_hamming_decide<GDCA_MAXB>>(dim3(1), dim3(1), 0, s, k_hamming_decide_mk(sc, 1ll, force < 0 ? 0 : force, 0)));  // tiny families: the exact form
    }
    // every form that may run is launched: where the device decides between them, and behind a forced bound form, whose list may
    // overflow (a form that has nothing to do exits on sc->ham_mode / sc->ham_ncand: empty workgroups)
    const bool decided = !(force < 0 && ntile >= 64);
    const int form = decided ? (force < 0 ? 0 : force) : -1;
    if (fp4_img && (form == 2 || form < 0)) gdca_launch_hamming_fp4(s, Zb, fp4_img, N, M, sc, list, cap);
    if (form == 1 || form < 0)
        (gdca_launch<k_hamming_args, k_hamming<1, HAM_BOUND_PLANES, false>, k_hamming<GDCA_MAXB, HAM_BOUND_PLANES, false>>(dim3((unsigned)ntile), dim3(256), 0, s, k_hamming_mk(Zb, cnt, NW, M, Mt, sc, list, cap)));
    if (form != 0)
        (gdca_launch<k_hamming_refine_args, k_hamming_refine<1>, k_hamming_refine<GDCA_MAXB>>(dim3(2048), dim3(256), 0, s, k_hamming_refine_mk(Z, (const int2 *)list, cap, N, cnt, (const gdca_dev_scalars *)sc)));
    (gdca_launch<k_hamming_args, k_hamming<1, NPLANES, false>, k_hamming<GDCA_MAXB, NPLANES, false>>(dim3((unsigned)ntile), dim3(256), 0, s, k_hamming_mk(Zb, cnt, NW, M, Mt, sc, list, cap)));
}

// ---- the second, independent implementation (GDCA_FORCE_FALLBACK) -------------------------------------------------------------
// The reference tests its Hamming reweighting twice: DCAUtils' packed XOR / popcount path and, with
// ENV["DCAUTILS_FORCE_FALLBACK"], its plain fallback, both against the same golden (test/runtests.jl:78-86).  The analogue
// here: a plain byte-compare kernel that shares nothing with the bit-sliced one -- no bit planes, no pair tiles, no symmetry, no
// atomics: workgroup <-> sequence k (its N bytes in LDS), thread <-> sequences l = tid, tid + 256, ..., every distance counted
// position by position, one tree reduction per k.  M^2 N byte compares instead of M^2 N / 2 x 0.19 instructions: for tests only.
struct k_hamming_bytes_args {
    const int8_t *Z;
    int32_t *cnt;
    int N;
    int M;
    const gdca_dev_scalars *sc;
};
static inline k_hamming_bytes_args k_hamming_bytes_mk(const int8_t *Z, int32_t *cnt, int N, int M, const gdca_dev_scalars *sc)
{
    return k_hamming_bytes_args{Z, cnt, N, M, sc};
}
template <int CAP>
__global__ __launch_bounds__(256) void k_hamming_bytes(const BatchArgs<k_hamming_bytes_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const int8_t *__restrict__ Z = a_.Z;
    int32_t *__restrict__ cnt = a_.cnt;
    int N = a_.N;
    int M = a_.M;
    const gdca_dev_scalars *sc = a_.sc;
    extern __shared__ int8_t zk[];
    __shared__ int red[256];
    const int k = blockIdx.x, tid = threadIdx.x, thresh = sc->thresh;
    for (int i = tid; i < N; i += 256) zk[i] = Z[(size_t)k * N + i];
    __syncthreads();
    int mine = 0;
    for (int l = tid; l < M; l += 256) {
        if (l == k) continue;
        const int8_t *zl = Z + (size_t)l * N;
        int d = 0;
        for (int i = 0; i < N; ++i) d += (zl[i] != zk[i]);
        mine += (d < thresh);
    }
    red[tid] = mine;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    if (tid == 0) cnt[k] = red[0];
}

void gdca_launch_hamming_fallback(hipStream_t s, const int8_t *Z, int32_t *cnt, int N, int M, const gdca_dev_scalars *sc)
{
    (gdca_launch<k_hamming_bytes_args, k_hamming_bytes<1>, k_hamming_bytes<GDCA_MAXB>>(dim3((unsigned)M), dim3(256), (size_t)N, s, k_hamming_bytes_mk(Z, cnt, N, M, sc)));
}

// ---- weights ------------------------------------------------------------------------------------
int gdca_fix_shift(int M)
{
    // fixed-point scale 2^shift for the weighted tallies: M * 2^shift <= 2^63, and every weight
    // (<= 2^shift) must fit the 59-bit field of the tally kernel's packed {symbol, weight} word
    int lg = 0;
    while ((1ll << lg) < (long long)M) ++lg;
    const int sh = 63 - lg;
    return sh < 58 ? sh : 58;
}

struct k_weights_args {
    const int32_t *cnt;
    int M;
    int fix_shift;
    int32_t *n_out;
    double *W;
    unsigned long long *Wfix;
};
static inline k_weights_args k_weights_mk(const int32_t *cnt, int M, int fix_shift, int32_t *n_out, double *W, unsigned long long *Wfix)
{
    return k_weights_args{cnt, M, fix_shift, n_out, W, Wfix};
}
template <int CAP>
__global__ __launch_bounds__(256) void k_weights(const BatchArgs<k_weights_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const int32_t *__restrict__ cnt = a_.cnt;
    int M = a_.M;
    int fix_shift = a_.fix_shift;
    int32_t *__restrict__ n_out = a_.n_out;
    double *__restrict__ W = a_.W;
    unsigned long long *__restrict__ Wfix = a_.Wfix;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= M) return;
    const int n = 1 + cnt[k];
    if (n_out) n_out[k] = n;
    const double w = 1.0 / (double)n;  // IEEE division, correctly rounded
    W[k] = w;
    if (Wfix) Wfix[k] = (unsigned long long)rint(ldexp(w, fix_shift));
}

void gdca_launch_weights(hipStream_t s, const int32_t *cnt, int M, int fix_shift, int32_t *n_out, double *W,
                         unsigned long long *Wfix)
{
    (gdca_launch<k_weights_args, k_weights<1>, k_weights<GDCA_MAXB>>(dim3((M + 255) / 256), dim3(256), 0, s, k_weights_mk(cnt, M, fix_shift, n_out, W, Wfix)));
}

struct k_fix_weights_args {
    const double *W;
    int M;
    int fix_shift;
    unsigned long long *Wfix;
    gdca_dev_scalars *sc;
};
static inline k_fix_weights_args k_fix_weights_mk(const double *W, int M, int fix_shift, unsigned long long *Wfix, gdca_dev_scalars *sc)
{
    return k_fix_weights_args{W, M, fix_shift, Wfix, sc};
}
template <int CAP>
__global__ __launch_bounds__(256) void k_fix_weights(const BatchArgs<k_fix_weights_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const double *__restrict__ W = a_.W;
    int M = a_.M;
    int fix_shift = a_.fix_shift;
    unsigned long long *__restrict__ Wfix = a_.Wfix;
    gdca_dev_scalars *sc = a_.sc;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= M) return;
    const double w = W[k];
    // a weight outside [0, 1] (or NaN) does not fit the 59-bit field of the tally's packed word: flagged (bit 1 of
    // bad_symbol), staged as 0
    const bool ok = w >= 0.0 && w <= 1.0;
    if (!ok) atomicOr(&sc->bad_symbol, 2);
    Wfix[k] = ok ? (unsigned long long)rint(ldexp(w, fix_shift)) : 0ull;
}

void gdca_launch_fix_weights(hipStream_t s, const double *W, int M, int fix_shift, unsigned long long *Wfix,
                             gdca_dev_scalars *sc)
{
    (gdca_launch<k_fix_weights_args, k_fix_weights<1>, k_fix_weights<GDCA_MAXB>>(dim3((M + 255) / 256), dim3(256), 0, s, k_fix_weights_mk(W, M, fix_shift, Wfix, sc)));
}

// Meff = the sum of the weights, EXACT and rounded once (round to nearest even): what Python's math.fsum returns, whatever the
// order of the terms.  (Rounds 1-4 walked the strictly sequential f64 sum ((W[0] + W[1]) + W[2]) + ... -- one thread, M dependent
// v_add_f64: 0.3 ms at M = 50 000 on a stream of its own.  The reference's own sum(W) is Julia's pairwise, SIMD-reassociated sum,
// so no f64 summation order pins its last bit; an exactly rounded sum is the one definition that needs no order.)
// Every weight 1 / n_k is m 2^(E - 1075) with a 53-bit m and E - 1023 in [-31, 0]: as an integer multiple of 2^-84 it has at most
// 85 bits.  A thread adds the three 32-bit limbs of its terms into 64-bit counters (no carries: < 2^24 terms per thread), the
// workgroup adds those into LDS, and thread 0 resolves the carries into one 128-bit integer and rounds it to 53 bits.
#define MEFF_FRAC 84
struct k_meff_args {
    const double *W;
    int M;
    gdca_dev_scalars *sc;
};
static inline k_meff_args k_meff_mk(const double *W, int M, gdca_dev_scalars *sc)
{
    return k_meff_args{W, M, sc};
}
template <int CAP>
__global__ __launch_bounds__(1024) void k_meff(const BatchArgs<k_meff_args, CAP> B_)
{
    GDCA_MEMBER(B_);
    const double *__restrict__ W = a_.W;
    int M = a_.M;
    gdca_dev_scalars *sc = a_.sc;
    __shared__ unsigned long long limb[3];
    const int tid = threadIdx.x;
    if (tid < 3) limb[tid] = 0ull;
    __syncthreads();
    unsigned long long l0 = 0ull, l1 = 0ull, l2 = 0ull;
    // (eight loads in flight per thread: the loop is the latency of its loads, 49 trips at M = 50 000)
    for (int k0 = tid; k0 < M; k0 += 8 * (int)blockDim.x) {
        double w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + u * (int)blockDim.x;
            w[u] = k < M ? W[k] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(w[u]);
        const int E = (int)((b >> 52) & 0x7ffull);
        unsigned long long m = b & 0xfffffffffffffull;
        int sh = E - 1075 + MEFF_FRAC;  // 1 / n_k, n_k < 2^31: 1 .. 32
        if (E != 0) m |= 1ull << 52; else sh += 1;  // (subnormal: no hidden bit -- never a weight)
        if (sh < 0) {  // smaller than anything compute_weights produces: what is below 2^-84 is dropped
            m = sh > -64 ? m >> (-sh) : 0ull;
            sh = 0;
        }
        if (sh > 43) sh = 43;  // (a weight above 2^11 does not exist either: the limbs below stay exact for everything <= 1)
        const unsigned long long lo = m << sh, hi = sh ? m >> (64 - sh) : 0ull;
        l0 += lo & 0xffffffffull;
        l1 += lo >> 32;
        l2 += hi;
        }
    }
    // wave sums first (64 lanes x 2^24 terms x 2^32 < 2^64), then one LDS atomic per wave and limb
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        l0 += __shfl_xor(l0, o);
        l1 += __shfl_xor(l1, o);
        l2 += __shfl_xor(l2, o);
    }
    if ((tid & 63) == 0) {
        atomicAdd(&limb[0], l0);
        atomicAdd(&limb[1], l1);
        atomicAdd(&limb[2], l2);
    }
    __syncthreads();
    if (tid == 0) {
        // S = limb0 + limb1 2^32 + limb2 2^64 as (hi : lo)
        unsigned long long lo = limb[0], hi = limb[2];
        const unsigned long long mid = limb[1];
        const unsigned long long add = mid << 32;
        lo += add;
        hi += (mid >> 32) + (lo < add ? 1ull : 0ull);
        double r;
        if (hi == 0ull && lo < (1ull << 53)) {
            r = ldexp((double)lo, -MEFF_FRAC);
        } else {
            const int p = hi ? 127 - __builtin_clzll(hi) : 63 - __builtin_clzll(lo);  // top bit of S
            const int shift = p - 52;                                                   // >= 1: bits to drop
            unsigned long long kept, rem_hi, rem_lo, half_hi, half_lo;
            if (shift >= 64) {
                kept = hi >> (shift - 64);
                rem_hi = shift > 64 ? hi & ((1ull << (shift - 64)) - 1ull) : 0ull;
                rem_lo = lo;
            } else {
                kept = (lo >> shift) | (hi << (64 - shift));  // (hi < 2^(p - 63): nothing of it is lost)
                rem_hi = 0ull;
                rem_lo = lo & ((1ull << shift) - 1ull);
            }
            if (shift - 1 >= 64) {
                half_hi = 1ull << (shift - 1 - 64);
                half_lo = 0ull;
            } else {
                half_hi = 0ull;
                half_lo = 1ull << (shift - 1);
            }
            const bool above = rem_hi > half_hi || (rem_hi == half_hi && rem_lo > half_lo);
            const bool tie = rem_hi == half_hi && rem_lo == half_lo;
            if (above || (tie && (kept & 1ull))) ++kept;  // (2^53 after the carry is still exact in f64)
            r = ldexp((double)kept, shift - MEFF_FRAC);
        }
        sc->Meff = r;
    }
}

void gdca_launch_meff(hipStream_t s, const double *W, int M, gdca_dev_scalars *sc)
{
    (gdca_launch<k_meff_args, k_meff<1>, k_meff<GDCA_MAXB>>(dim3(1), dim3(1024), 0, s, k_meff_mk(W, M, sc)));
}
