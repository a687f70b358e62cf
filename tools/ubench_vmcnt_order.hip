// Does gfx950 complete vector-memory operations of one wave IN ORDER as far as vmcnt is concerned?
//
// hipcc's waitcnt insertion assumes so for this target (one counter for loads, stores, atomics, scratch accesses and cache
// maintenance: `s_waitcnt vmcnt(N)` = "everything but the N youngest operations has completed").  If some class of operation can
// be counted off before an OLDER load has delivered its data, a wait with N > 0 lets the wave read that load's destination
// register too early.  Test, per class X:
//       v_mov   dst, SENTINEL
//       global_load_dword dst, [a cold address: HBM miss]          (A, slow)
//       X                                                           (B, fast: hot address / scratch / cache maintenance)
//       s_waitcnt vmcnt(1)                                          (in-order completion => A has landed)
//       v_mov   seen, dst
//       s_waitcnt vmcnt(0)
// and count lanes where `seen` is not A's value.  Class 0 (a second, hot global load) is the control.
//
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_vmcnt_order.hip -o tools/_bin/ubench_vmcnt_order && tools/_bin/ubench_vmcnt_order
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define SENTINEL 0xDEADBEEFu

template <int X>
__global__ __launch_bounds__(256) void k_order(const unsigned *__restrict__ big, size_t nbig, unsigned *__restrict__ hot, int iters,
                                               unsigned long long *bad, unsigned long long *sentinel_seen)
{
    // a private array with a run-time index: the kernel gets a scratch allocation and flat-scratch is set up
    volatile unsigned priv[8];
    for (int k = 0; k < 8; ++k) priv[k] = k + threadIdx.x;
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    unsigned *myhot = hot + gid;
    unsigned long long nbad = 0, nsent = 0;
    unsigned tmp = priv[(gid >> 3) & 7], ret = 0;
    const unsigned zero_off = 0;
    for (int it = 0; it < iters; ++it) {
        // cold address: a different 128-byte line per lane and iteration, far apart
        const size_t idx = ((size_t)gid * 2654435761ull + (size_t)it * 40503ull * 1048583ull) % (nbig / 32) * 32;
        const unsigned *pa = big + idx;
        unsigned dst, seen, junk = 0;
        if constexpr (X == 0)
            asm volatile("v_mov_b32 %0, %4\n\tglobal_load_dword %0, %3, off\n\tglobal_load_dword %2, %5, off\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(dst), "=&v"(seen), "=&v"(junk) : "v"(pa), "v"(SENTINEL), "v"(myhot) : "memory");
        else if constexpr (X == 1)
            asm volatile("v_mov_b32 %0, %4\n\tglobal_load_dword %0, %3, off\n\tscratch_load_dword %2, %5, off\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(dst), "=&v"(seen), "+v"(tmp) : "v"(pa), "v"(SENTINEL), "v"(zero_off) : "memory");
        else if constexpr (X == 2)
            asm volatile("v_mov_b32 %0, %3\n\tglobal_load_dword %0, %2, off\n\tscratch_store_dword %5, %4, off\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(dst), "=&v"(seen) : "v"(pa), "v"(SENTINEL), "v"(tmp), "v"(zero_off) : "memory");
        else if constexpr (X == 3)
            asm volatile("v_mov_b32 %0, %3\n\tglobal_load_dword %0, %2, off\n\tglobal_store_dword %5, %4, off\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(dst), "=&v"(seen) : "v"(pa), "v"(SENTINEL), "v"(tmp), "v"(myhot) : "memory");
        else if constexpr (X == 4)
            asm volatile("v_mov_b32 %0, %3\n\tglobal_load_dword %0, %2, off\n\tglobal_store_dword %5, %4, off sc1\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(dst), "=&v"(seen) : "v"(pa), "v"(SENTINEL), "v"(tmp), "v"(myhot) : "memory");
        else if constexpr (X == 5)
            asm volatile("v_mov_b32 %0, %3\n\tglobal_load_dword %0, %2, off\n\tglobal_store_dword %5, %4, off sc0 sc1\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(dst), "=&v"(seen) : "v"(pa), "v"(SENTINEL), "v"(tmp), "v"(myhot) : "memory");
        else if constexpr (X == 6)
            asm volatile("v_mov_b32 %0, %3\n\tglobal_load_dword %0, %2, off\n\tglobal_atomic_add %5, %4, off\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(dst), "=&v"(seen) : "v"(pa), "v"(SENTINEL), "v"(tmp), "v"(myhot) : "memory");
        else if constexpr (X == 7)
            asm volatile("v_mov_b32 %0, %4\n\tglobal_load_dword %0, %3, off\n\tglobal_atomic_add %2, %6, %5, off sc0\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(dst), "=&v"(seen), "=&v"(junk) : "v"(pa), "v"(SENTINEL), "v"(tmp), "v"(myhot) : "memory");
        else if constexpr (X == 8)
            asm volatile("v_mov_b32 %0, %4\n\tglobal_load_dword %0, %3, off\n\tglobal_load_dword %2, %5, off sc1\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(dst), "=&v"(seen), "=&v"(junk) : "v"(pa), "v"(SENTINEL), "v"(myhot) : "memory");
        else if constexpr (X == 9)
            asm volatile("v_mov_b32 %0, %3\n\tglobal_load_dword %0, %2, off\n\tbuffer_inv sc1\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(dst), "=&v"(seen) : "v"(pa), "v"(SENTINEL) : "memory");
        else if constexpr (X == 10)
            asm volatile("v_mov_b32 %0, %3\n\tglobal_load_dword %0, %2, off\n\tbuffer_wbl2 sc1\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(dst), "=&v"(seen) : "v"(pa), "v"(SENTINEL) : "memory");
        else if constexpr (X == 11)  // a 16-byte cold load followed by a hot 8-byte store (the tile loop's mix)
            asm volatile("v_mov_b32 %0, %3\n\tglobal_load_dword %0, %2, off\n\tglobal_store_dwordx2 %5, %4, off sc1\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(dst), "=&v"(seen) : "v"(pa), "v"(SENTINEL), "v"((unsigned long long)tmp), "v"(hot + 2 * (size_t)gid) : "memory");
        const unsigned expect = (unsigned)(idx * 2654435761ull) ^ 0x5bd1e995u;
        nbad += seen != expect;
        nsent += seen == SENTINEL;
        tmp += dst + junk;
    }
    priv[gid & 7] = tmp + ret;
    if (priv[(gid + 1) & 7] == 0x12345678u) hot[0] = tmp;  // keep everything alive
    atomicAdd(bad, nbad);
    atomicAdd(sentinel_seen, nsent);
}

__global__ void k_fill(unsigned *big, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        big[i] = (unsigned)(i * 2654435761ull) ^ 0x5bd1e995u;
}

template <int X>
static void run(const char *name, const unsigned *big, size_t nbig, unsigned *hot, int iters, unsigned long long *cnt)
{
    (void)hipMemset(cnt, 0, 16);
    hipLaunchKernelGGL(k_order<X>, dim3(2048), dim3(256), 0, 0, big, nbig, hot, iters, cnt, cnt + 1);
    unsigned long long h[2] = {0, 0};
    hipError_t e = hipDeviceSynchronize();
    (void)hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost);
    printf("%-58s %12llu of %llu lane-reads saw something else than A's value (%llu the sentinel)%s\n", name, h[0],
           2048ull * 256 * iters, h[1], e == hipSuccess ? "" : "  [HIP ERROR]");
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const size_t nbig = (size_t)1 << 30;  // 4 GiB of u32
    unsigned *big, *hot;
    unsigned long long *cnt;
    if (hipMalloc(&big, nbig * 4) != hipSuccess || hipMalloc(&hot, (size_t)2048 * 256 * 16) != hipSuccess || hipMalloc(&cnt, 16) != hipSuccess) {
        printf("allocation failed\n");
        return 1;
    }
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, big, nbig);
    (void)hipMemset(hot, 0, (size_t)2048 * 256 * 16);
    (void)hipDeviceSynchronize();
    printf("slow global load A, then X, then s_waitcnt vmcnt(1): has A landed?  (%d iterations x 2048 x 256 lanes)\n", iters);
    run<0>("X = global_load (hot)                    [control]", big, nbig, hot, iters, cnt);
    run<1>("X = scratch_load", big, nbig, hot, iters, cnt);
    run<2>("X = scratch_store", big, nbig, hot, iters, cnt);
    run<3>("X = global_store", big, nbig, hot, iters, cnt);
    run<4>("X = global_store sc1            (write-through, agent)", big, nbig, hot, iters, cnt);
    run<5>("X = global_store sc0 sc1        (system scope)", big, nbig, hot, iters, cnt);
    run<6>("X = global_atomic_add           (no return)", big, nbig, hot, iters, cnt);
    run<7>("X = global_atomic_add sc0       (returns)", big, nbig, hot, iters, cnt);
    run<8>("X = global_load sc1             (agent-scope load)", big, nbig, hot, iters, cnt);
    run<9>("X = buffer_inv sc1", big, nbig, hot, iters, cnt);
    run<10>("X = buffer_wbl2 sc1", big, nbig, hot, iters, cnt);
    run<11>("X = global_store_dwordx2 sc1", big, nbig, hot, iters, cnt);
    return 0;
}
