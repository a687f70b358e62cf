// VALU instruction issue-rate probe (cycles per wave-instruction per SIMD at 1/2/4 waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k_op(unsigned *out, int iters)
{
    unsigned x[16], y = threadIdx.x * 2654435761u, z = threadIdx.x ^ 0x9e3779b9u;
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 40503u + i * 7919u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) x[i] = x[i] ^ y;
            if (OP == 1) x[i] = __builtin_amdgcn_bitop3_b32(x[i], y, z, 0xBE);
            if (OP == 2) x[i] = __builtin_popcount(y ^ (unsigned)i) + x[i];        // v_bcnt_u32_b32 with accumulate (+ xor folded? see asm)
            if (OP == 3) x[i] = x[i] + y;
            if (OP == 4) x[i] = (x[i] | y) | z;                                      // v_or3_b32
            if (OP == 5) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(x[i]) : "v"(y));
            if (OP == 6) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xbe" : "+v"(x[i]) : "v"(y), "v"(z));
            if (OP == 7) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
            if (OP == 8) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
            if (OP == 9) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
            if (OP == 10) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
            if (OP == 11) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(x[i]) : "v"(y), "v"(z));
        }
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
int run(const char *name, unsigned *out)
{
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int nb = 256 * wps, iters = 20000;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_op<OP>, dim3(nb), dim3(256), 0, 0, out, 100);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_op<OP>, dim3(nb), dim3(256), 0, 0, out, iters);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s waves/SIMD=%d: %.2f clk@2.4GHz per wave-instr per SIMD\n", name, wps, ms * 1e-3 * 2.4e9 / ((double)iters * 16 * wps));
    }
    return 0;
}

int main()
{
    unsigned *out;
    CK(hipMalloc(&out, 4 * 2048 * 256));
    run<7>("v_xor_b32", out);
    run<6>("v_bitop3_b32 (asm)", out);
    run<5>("v_bcnt_u32_b32 acc (asm)", out);
    run<3>("v_add_u32", out);
    run<4>("v_or3_b32", out);
    run<8>("v_sad_u8", out);
    run<9>("v_xad_u32", out);
    run<10>("v_and_or_b32", out);
    run<11>("v_dot4_u32_u8", out);
    return 0;
}
