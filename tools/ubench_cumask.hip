// CU-mask experiment: f64 MFMA loop on one CU subset, integer VALU loop on the complement.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_mfma(double *out, int iters, unsigned *xcc)
{
    double4_t acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (double4_t){0, 0, 0, 0};
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && xcc) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        xcc[blockIdx.x * 2] = id;
        xcc[blockIdx.x * 2 + 1] = hw;
    }
}

__global__ __launch_bounds__(256) void k_int(unsigned *out, int iters)
{
    unsigned x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 2654435761u + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_popcount(x[i] ^ (x[(i + 1) & 15] | it)) + x[i] * 3u;
    unsigned s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    double *o1;
    unsigned *o2, *xcc;
    CK(hipMalloc(&o1, 8 * 4096 * 256));
    CK(hipMalloc(&o2, 4 * 4096 * 256));
    CK(hipMalloc(&xcc, 8 * 4096));
    const int it_m = 10000, it_i = 60000;
    // reference: each alone on the full chip
    hipEvent_t e0, e1, e2, e3;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
    hipLaunchKernelGGL(k_mfma, dim3(512), dim3(256), 0, 0, o1, 100, nullptr);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_mfma, dim3(512), dim3(256), 0, 0, o1, it_m, nullptr); CK(hipEventRecord(e1));
    hipLaunchKernelGGL(k_int, dim3(1024), dim3(256), 0, 0, o2, 100);
    CK(hipEventRecord(e2)); hipLaunchKernelGGL(k_int, dim3(1024), dim3(256), 0, 0, o2, it_i); CK(hipEventRecord(e3));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("alone: mfma 512 blocks full chip: %.3f ms = %.1f TF\n", ms, 512.0 * 4 * it_m * 4 * 2048 / ms / 1e9);
    CK(hipEventElapsedTime(&ms, e2, e3));
    printf("alone: int 1024 blocks full chip: %.3f ms\n", ms);

    for (int ncu_m = 128; ncu_m <= 224; ncu_m += 32) {
        // masks: first try "low bits = MFMA partition"
        std::vector<uint32_t> mA(8, 0), mB(8, 0);
        for (int c = 0; c < 256; ++c) {
            // interleave so that every group of 8 consecutive CU ids contributes proportionally
            const bool toA = ((c * ncu_m) / 256) != (((c + 1) * ncu_m) / 256);
            (toA ? mA : mB)[c / 32] |= 1u << (c % 32);
        }
        hipStream_t sA, sB;
        CK(hipExtStreamCreateWithCUMask(&sA, 8, mA.data()));
        CK(hipExtStreamCreateWithCUMask(&sB, 8, mB.data()));
        const int nbA = ncu_m * 2, nbB = (256 - ncu_m) * 4;
        hipLaunchKernelGGL(k_mfma, dim3(nbA), dim3(256), 0, sA, o1, 100, xcc);
        hipLaunchKernelGGL(k_int, dim3(nbB), dim3(256), 0, sB, o2, 100);
        CK(hipDeviceSynchronize());
        std::vector<unsigned> hx(nbA * 2);
        CK(hipMemcpy(hx.data(), xcc, 8 * nbA, hipMemcpyDeviceToHost));
        int perx[8] = {0};
        for (int b = 0; b < nbA; ++b) perx[hx[2 * b] & 7]++;
        // alone on partition
        CK(hipEventRecord(e0, sA)); hipLaunchKernelGGL(k_mfma, dim3(nbA), dim3(256), 0, sA, o1, it_m, nullptr); CK(hipEventRecord(e1, sA));
        CK(hipDeviceSynchronize());
        float msA_alone; CK(hipEventElapsedTime(&msA_alone, e0, e1));
        CK(hipEventRecord(e2, sB)); hipLaunchKernelGGL(k_int, dim3(nbB), dim3(256), 0, sB, o2, it_i); CK(hipEventRecord(e3, sB));
        CK(hipDeviceSynchronize());
        float msB_alone; CK(hipEventElapsedTime(&msB_alone, e2, e3));
        // concurrent
        CK(hipEventRecord(e0, sA)); hipLaunchKernelGGL(k_mfma, dim3(nbA), dim3(256), 0, sA, o1, it_m, nullptr); CK(hipEventRecord(e1, sA));
        CK(hipEventRecord(e2, sB)); hipLaunchKernelGGL(k_int, dim3(nbB), dim3(256), 0, sB, o2, it_i); CK(hipEventRecord(e3, sB));
        CK(hipDeviceSynchronize());
        float msA, msB;
        CK(hipEventElapsedTime(&msA, e0, e1)); CK(hipEventElapsedTime(&msB, e2, e3));
        printf("mask %3d CUs mfma / %3d CUs int: mfma alone %.3f ms (%.1f TF) concurrent %.3f ms (%.1f TF); int alone %.3f concurrent %.3f ms; mfma blocks per XCC:", ncu_m, 256 - ncu_m,
               msA_alone, nbA * 4.0 * it_m * 4 * 2048 / msA_alone / 1e9, msA, nbA * 4.0 * it_m * 4 * 2048 / msA / 1e9, msB_alone, msB);
        for (int x = 0; x < 8; ++x) printf(" %d", perx[x]);
        printf("\n");
        CK(hipStreamDestroy(sA)); CK(hipStreamDestroy(sB));
    }
    return 0;
}
