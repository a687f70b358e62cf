// Which workgroup shapes get onto the reserved CUs while a persistent, register- and LDS-hungry kernel holds every other
// CU?  (stand-in for the persistent trailing update: 256 threads, 256 VGPRs, 74 KB LDS, 2 per CU, runs ~1 ms)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256, 2) void k_hog(double *out, int iters)
{
    __shared__ double lds[9000];   // 72 KB
    double4_t acc[6][5];
    double a[6], b[5];
    for (int i = 0; i < 6; ++i) a[i] = 1.0 + 1e-9 * (threadIdx.x + 64 * i);
    for (int j = 0; j < 5; ++j) b[j] = 1.0 - 1e-9 * (threadIdx.x + 64 * j);
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 5; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    lds[threadIdx.x] = a[0];
    __syncthreads();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    double s = lds[(threadIdx.x * 7) % 9000];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 5; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int THREADS, int NV, int LDSB>
__global__ __launch_bounds__(THREADS) void k_shape(double *out)
{
    __shared__ char lds[LDSB];
    double v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = threadIdx.x * 1e-3 + i;
    lds[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < 50; ++r)
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = v[i] * 1.0000001 + v[(i + 1) % NV] * 1e-9;
    double s = lds[(threadIdx.x + 1) % LDSB];
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

int main()
{
    double *o1, *o2;
    CK(hipMalloc(&o1, 8ull * 1024 * 256));
    CK(hipMalloc(&o2, 8ull * 1024 * 1024));
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t sp;
    CK(hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, greatest));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<uint32_t> m(8, 0xFFFFFFFFu);
    m[0] &= ~0xFFu;   // one CU per XCD (SE0.CU0)
    hipStream_t sh;
    CK(hipExtStreamCreateWithCUMask(&sh, 8, m.data()));
    for (int hog_grid : {496, 480, 448}) {
        auto run = [&](const char *name, auto launch) -> int {
            hipLaunchKernelGGL(k_hog, dim3(hog_grid), dim3(256), 0, sh, o1, 1600);   // ~1.5 ms
            std::vector<float> lat;
            for (int rep = 0; rep < 9; ++rep) {
                CK(hipEventRecord(e0, sp));
                launch();
                CK(hipEventRecord(e1, sp));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                lat.push_back(ms * 1e3f);
            }
            const bool running = hipStreamQuery(sh) == hipErrorNotReady;
            CK(hipDeviceSynchronize());
            printf("hog grid %d  %-44s latency us:", hog_grid, name);
            for (float x : lat) printf(" %.0f", x);
            printf("  (hog running at end: %d)\n", (int)running);
            return 0;
        };
        run("1 WG x 256 thr, 16 VGPR, 1 KB LDS", [&] { hipLaunchKernelGGL((k_shape<256, 8, 1024>), dim3(1), dim3(256), 0, sp, o2); });
        run("1 WG x 768 thr, 16 VGPR, 1 KB LDS", [&] { hipLaunchKernelGGL((k_shape<768, 8, 1024>), dim3(1), dim3(768), 0, sp, o2); });
        run("1 WG x 768 thr, ~140 VGPR, 1 KB LDS", [&] { hipLaunchKernelGGL((k_shape<768, 64, 1024>), dim3(1), dim3(768), 0, sp, o2); });
        run("1 WG x 768 thr, ~140 VGPR, 45 KB LDS", [&] { hipLaunchKernelGGL((k_shape<768, 64, 45000>), dim3(1), dim3(768), 0, sp, o2); });
        run("1 WG x 256 thr, ~250 VGPR, 37 KB LDS", [&] { hipLaunchKernelGGL((k_shape<256, 120, 37000>), dim3(1), dim3(256), 0, sp, o2); });
        run("8 WG x 256 thr, ~250 VGPR, 37 KB LDS", [&] { hipLaunchKernelGGL((k_shape<256, 120, 37000>), dim3(8), dim3(256), 0, sp, o2); });
        run("576 WG x 256 thr, 16 VGPR then 1 x 768/140/45K", [&] {
            hipLaunchKernelGGL((k_shape<256, 8, 1024>), dim3(576), dim3(256), 0, sp, o2);
            hipLaunchKernelGGL((k_shape<768, 64, 45000>), dim3(1), dim3(768), 0, sp, o2);
        });
    }
    return 0;
}
