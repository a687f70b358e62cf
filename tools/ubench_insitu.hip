// In-situ launch latency experiment: how long does a small, high-priority kernel take to get through the chip while a
// register-hungry long kernel (a stand-in for the trailing update: 256 threads, 2 workgroups per CU, ~80 us per
// workgroup, thousands of workgroups) occupies every CU -- and how do CU masks on the long kernel's stream change that?
// Also prints the CU-mask bit -> (XCC, SE, CU) mapping.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256, 2) void k_hog(double *out, int iters)
{
    double4_t acc[4][4];
    double a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = 1.0 + 1e-9 * (threadIdx.x + 64 * i);
        b[i] = 1.0 - 1e-9 * (threadIdx.x + 64 * i);
        for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    double s = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_small(unsigned *where, int spin)
{
    __shared__ int x;
    if (threadIdx.x == 0) {
        unsigned id, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        where[blockIdx.x * 2] = id;
        where[blockIdx.x * 2 + 1] = hw;
        x = 0;
    }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(10);
}

int main()
{
    double *o1;
    unsigned *wh;
    CK(hipMalloc(&o1, 8ull * 16384 * 256));
    CK(hipMalloc(&wh, 8 * 8192));
    // ---- 1. mask bit -> XCC / HW_ID ----
    printf("mask bit -> xcc, hw_id (se_id = bits 13..15? cu_id = bits 8..11; printed raw)\n");
    for (int b : {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 31, 32, 33, 63, 64, 128, 255}) {
        std::vector<uint32_t> m(8, 0);
        m[b / 32] = 1u << (b % 32);
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, 8, m.data()) != hipSuccess) { printf("bit %d: create failed\n", b); (void)hipGetLastError(); continue; }
        hipLaunchKernelGGL(k_small, dim3(4), dim3(256), 0, s, wh, 0);
        CK(hipStreamSynchronize(s));
        unsigned h[8];
        CK(hipMemcpy(h, wh, sizeof(h), hipMemcpyDeviceToHost));
        printf("bit %3d: xcc %u %u %u %u  hw_id %08x %08x\n", b, h[0] & 15, h[2] & 15, h[4] & 15, h[6] & 15, h[1], h[3]);
        CK(hipStreamDestroy(s));
    }
    // ---- 2. in-situ latency ----
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t sp;
    CK(hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, hi));
    hipEvent_t e0, e1, h0, h1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&h0)); CK(hipEventCreate(&h1));
    const int hog_iters = 620;   // 16 MFMAs x 64 clk x 620 x 2 co-resident waves ~ 80 us
    {
        CK(hipEventRecord(h0, 0));
        hipLaunchKernelGGL(k_hog, dim3(512), dim3(256), 0, 0, o1, hog_iters);
        CK(hipEventRecord(h1, 0));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, h0, h1));
        printf("hog: one round of 512 workgroups %.1f us\n", ms * 1e3);
    }
    struct Cfg { const char *name; int reserve; int spread; };
    for (Cfg cfg : {Cfg{"no mask", 0, 0}, Cfg{"reserve bits 0..0", 1, 0}, Cfg{"reserve bits 0..7", 8, 0}, Cfg{"reserve bits 0..15", 16, 0},
                    Cfg{"reserve bits 0..31", 32, 0}, Cfg{"reserve every 32nd bit (8)", 8, 32}, Cfg{"reserve every 16th bit (16)", 16, 16}}) {
        std::vector<uint32_t> m(8, 0xFFFFFFFFu);
        for (int r = 0; r < cfg.reserve; ++r) {
            const int bit = cfg.spread ? r * cfg.spread : r;
            m[bit / 32] &= ~(1u << (bit % 32));
        }
        hipStream_t sh;
        if (cfg.reserve == 0) CK(hipStreamCreateWithFlags(&sh, hipStreamNonBlocking));
        else CK(hipExtStreamCreateWithCUMask(&sh, 8, m.data()));
        for (int nsmall : {1, 8, 18, 64, 576}) {
            // hog for ~2.4 ms, small kernels launched into its middle
            hipLaunchKernelGGL(k_hog, dim3(512 * 30), dim3(256), 0, sh, o1, hog_iters);
            std::vector<float> lat;
            for (int rep = 0; rep < 12; ++rep) {
                CK(hipEventRecord(e0, sp));
                hipLaunchKernelGGL(k_small, dim3(nsmall), dim3(256), 0, sp, wh, 0);
                CK(hipEventRecord(e1, sp));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                lat.push_back(ms * 1e3f);
            }
            bool hog_running = hipStreamQuery(sh) == hipErrorNotReady;
            CK(hipDeviceSynchronize());
            std::sort(lat.begin(), lat.end());
            printf("%-32s small kernel %3d WGs: latency us min %.1f median %.1f max %.1f  (hog still running at the end: %d)\n", cfg.name, nsmall,
                   lat.front(), lat[lat.size() / 2], lat.back(), (int)hog_running);
        }
        // a dependent chain of 10 small kernels (18 WGs each)
        hipLaunchKernelGGL(k_hog, dim3(512 * 30), dim3(256), 0, sh, o1, hog_iters);
        CK(hipEventRecord(e0, sp));
        for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(k_small, dim3(18), dim3(256), 0, sp, wh, 0);
        CK(hipEventRecord(e1, sp));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-32s chain of 10 x 18 WGs: %.1f us total\n", cfg.name, ms * 1e3);
        CK(hipDeviceSynchronize());
        CK(hipStreamDestroy(sh));
    }
    return 0;
}
