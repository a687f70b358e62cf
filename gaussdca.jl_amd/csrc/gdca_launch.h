// How the front-end and score kernels of libgdca.so are launched: alone (one grid per kernel and family, as ever) or -- for the
// K members of a phase batch (gdca_run_dev_phased) -- as ONE grid per kernel KIND that carries all members.
//
// Why (round 6, VERDICT r05 #1a): a small family's front end is ~25 kernels of a few dozen to a few hundred workgroups each.  Eight of
// them batched by phase were 481 host calls and 1.8 ms between two merged sweep launches of 3.6 ms; side by side on four streams
// their kernels overlapped 1.8-fold, not eightfold.  Every such kernel is therefore a template over CAP, the members its launch
// carries, and takes ONE argument, BatchArgs<its argument struct, CAP>: CAP = 1 is the launch of old (two-dimensional grid, the
// member's arguments, blockIdx / gridDim the built-ins' own values: the compiled code is what it was); CAP = GDCA_MAXB is a FLAT grid
// over the blocks of up to 16 members -- a block finds its member from a prefix table in the kernel-argument segment, then its own
// (x, y) inside that member's grid, and reads that member's arguments with scalar loads.  No empty blocks: members of different
// sizes share a launch at no cost.  The body is the same text for both (GDCA_MEMBER at its top shadows blockIdx / gridDim).
// (A first form -- the body as an always-inline __device__ function under two generic __global__ wrappers -- compiled the three-plane
// Hamming kernel to 168 VGPRs with 7 spills instead of 127 and none: the very same text inlined into a kernel is not compiled like
// the kernel, with or without __restrict__, early returns or LDS declared inside; tests/test_kernel_resources.py watches for it.)
//
// Host side: gdca_launch<Args, k<1>, k<GDCA_MAXB>>(grid, block, lds, stream, args) launches k<1> at once -- unless the calling thread
// is RECORDING (gdca_recorder): then the launch is appended to the current member's list, and gdca_recorder::flush() walks the
// members' lists in lockstep and issues one k<GDCA_MAXB> launch per run of equal kernel kinds.  Members' lists need not be equal (the
// Hamming form, the score kind, the alphabet may differ): a member whose next kernel is of another kind simply waits a round.  Any
// launch that is not recordable (GDCA_LAUNCH_DIRECT) flushes first, so the order of every member's own kernels is always kept.
// Results are bit for bit those of single launches: a kernel's body cannot tell which instantiation runs it.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <type_traits>
#include <vector>

#define GDCA_MAXB 16  // members one batched launch carries (a phase batch of more members is issued in slices)

// the kernel argument of every front-end / score kernel: the arguments of CAP members and where each member's blocks start in the flat
// grid.  CAP = 1 (a launch of its own): the member's arguments and nothing else; the grid is the member's own two-dimensional one.
template <typename Args, int CAP>
struct BatchArgs {
    Args m[CAP];
    unsigned first[CAP + 1];  // first[k] = blocks of the members before k; first[n .. CAP] = all
    unsigned gx[CAP], gy[CAP];
    int n;
};
template <typename Args>
struct BatchArgs<Args, 1> {
    Args m[1];
};

struct gdca_member {
    int k;       // which member this block works for
    uint3 b, g;  // the block's index and the grid's extent inside that member's own grid
};
template <typename Args, int CAP>
__device__ __forceinline__ gdca_member gdca_decode(const BatchArgs<Args, CAP> &B)
{
    if constexpr (CAP == 1) {
        return gdca_member{0, {blockIdx.x, blockIdx.y, blockIdx.z}, {gridDim.x, gridDim.y, gridDim.z}};
    } else {
        const unsigned bid = blockIdx.x;
        int k = 0;
        while (k + 1 < B.n && bid >= B.first[k + 1]) ++k;
        const unsigned local = bid - B.first[k], gx = B.gx[k];
        const unsigned by = local / gx;
        return gdca_member{k, {local - by * gx, by, 0u}, {gx, B.gy[k], 1u}};
    }
}
// first statement of every such kernel: a_ = this block's member's arguments; the built-ins' names blockIdx / gridDim, shadowed by the
// block's place in that member's grid (CAP = 1: the built-ins' own values -- the code of a single launch is what it always was)
#define GDCA_MEMBER(B)                                   \
    const gdca_member mem_ = gdca_decode(B);             \
    const auto &a_ = (B).m[mem_.k];                      \
    const uint3 blockIdx = mem_.b, gridDim = mem_.g;     \
    (void)blockIdx;                                      \
    (void)gridDim

// ---- the recorder ---------------------------------------------------------------------------------------------------------------
#define GDCA_OP_ARG_BYTES 168
struct gdca_op {
    // issues ops[0 .. n-1] (all of this kind, n <= GDCA_MAXB) as one launch on s
    void (*launch)(hipStream_t s, const gdca_op *const *ops, int n);
    dim3 grid, block;
    unsigned lds;
    alignas(8) unsigned char args[GDCA_OP_ARG_BYTES];
};

struct gdca_recorder {
    hipStream_t stream = nullptr;        // the stream the recorded launches are meant for (and will be issued on)
    std::vector<std::vector<gdca_op>> lists;  // one per member
    int cur = -1;
    long launches = 0, ops = 0;          // what flush() issued / was handed (statistics)

    // this thread's active recorder (nullptr: launches go out at once)
    static gdca_recorder *&active();
    void begin(hipStream_t s, int members);
    void member(int k) { cur = k; }
    void add(const gdca_op &op);
    // issues everything recorded so far (lockstep over the members, one launch per run of equal kinds) and empties the lists; the
    // recorder stays active.  Returns the first HIP error of a launch, hipSuccess otherwise.
    hipError_t flush();
    hipError_t end();  // flush + deactivate
};

template <typename Args, void (*K1)(BatchArgs<Args, 1>), void (*KB)(BatchArgs<Args, GDCA_MAXB>)>
inline void gdca_issue(hipStream_t s, const gdca_op *const *ops, int n)
{
    static_assert(sizeof(Args) <= GDCA_OP_ARG_BYTES, "raise GDCA_OP_ARG_BYTES");
    static_assert(sizeof(BatchArgs<Args, GDCA_MAXB>) <= 4096, "a batched launch's arguments must fit the kernel-argument segment");
    unsigned lds = 0;
    for (int i = 0; i < n; ++i) lds = ops[i]->lds > lds ? ops[i]->lds : lds;
    if (n == 1) {
        BatchArgs<Args, 1> B;
        memcpy(&B.m[0], ops[0]->args, sizeof(Args));
        if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(K1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(K1, ops[0]->grid, ops[0]->block, lds, s, B);
        return;
    }
    BatchArgs<Args, GDCA_MAXB> B;
    unsigned total = 0;
    for (int i = 0; i < n; ++i) {
        memcpy(&B.m[i], ops[i]->args, sizeof(Args));
        B.first[i] = total;
        B.gx[i] = ops[i]->grid.x;
        B.gy[i] = ops[i]->grid.y;
        total += ops[i]->grid.x * ops[i]->grid.y;
    }
    for (int i = n; i < GDCA_MAXB; ++i) {
        memcpy(&B.m[i], ops[0]->args, sizeof(Args));
        B.gx[i] = 1;
        B.gy[i] = 1;
    }
    for (int i = n; i <= GDCA_MAXB; ++i) B.first[i] = total;
    B.n = n;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(KB), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(KB, dim3(total), ops[0]->block, lds, s, B);
}

// K1 / KB: the kernel's instantiations for one member and for GDCA_MAXB members
template <typename Args, void (*K1)(BatchArgs<Args, 1>), void (*KB)(BatchArgs<Args, GDCA_MAXB>)>
inline void gdca_launch(dim3 grid, dim3 block, size_t lds, hipStream_t s, const Args &a)
{
    static_assert(std::is_trivially_copyable<Args>::value, "kernel arguments are copied as bytes");
    gdca_op op;
    op.launch = &gdca_issue<Args, K1, KB>;
    op.grid = grid;
    op.block = block;
    op.lds = (unsigned)lds;
    memcpy(op.args, &a, sizeof(Args));
    gdca_recorder *r = gdca_recorder::active();
    // recordable: the recorder's stream, a two-dimensional non-empty grid
    if (r && r->cur >= 0 && s == r->stream && grid.z == 1 && grid.x > 0 && grid.y > 0) {
        r->add(op);
        return;
    }
    if (r) (void)r->flush();
    const gdca_op *one = &op;
    gdca_issue<Args, K1, KB>(s, &one, 1);
}

// a launch (or any other stream operation) that is not recordable: everything recorded so far goes out first
#define GDCA_FLUSH_RECORDED()                                 \
    do {                                                      \
        if (gdca_recorder *r_ = gdca_recorder::active()) (void)r_->flush(); \
    } while (0)
#define GDCA_LAUNCH_DIRECT(...)          \
    do {                                 \
        GDCA_FLUSH_RECORDED();           \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

// stream operations of the host code in their recordable forms (gdca_api.hip): a fill of 32-bit words, a device time stamp
void gdca_fill_async(hipStream_t s, void *p, int byte_value, size_t bytes);
struct gdca_dev_scalars;
void gdca_launch_stamp(hipStream_t s, gdca_dev_scalars *sc, int slot, int slot2);  // slot2 < 0: one slot
