// Is hipStreamWaitValue32 usable here, and what does a stream wait on a value written by a RUNNING kernel cost?
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_producer(unsigned *flag, double *data, int spin_before, int spin_after)
{
    for (int i = 0; i < spin_before; ++i) __builtin_amdgcn_s_sleep(100);
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        data[0] = 42.0;
        __threadfence();
        atomicAdd(flag, 1u);
    }
    for (int i = 0; i < spin_after; ++i) __builtin_amdgcn_s_sleep(100);
}
__global__ void k_consumer(const double *data, double *out, long long *stamp)
{
    out[0] = data[0];
    stamp[0] = wall_clock64();
}
__global__ void k_stamp(long long *stamp) { stamp[0] = wall_clock64(); }
int main()
{
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    unsigned *flag; double *data, *out; long long *st;
    CK(hipMalloc(&flag, 4)); CK(hipMalloc(&data, 8)); CK(hipMalloc(&out, 8)); CK(hipMalloc(&st, 64));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(flag, 0, 4)); CK(hipMemset(data, 0, 8)); CK(hipMemset(out, 0, 8));
        CK(hipDeviceSynchronize());
        // producer runs ~100 us before the flag and ~2000 us after it
        hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, a, st + 0);
        hipLaunchKernelGGL(k_producer, dim3(1), dim3(64), 0, a, flag, data, 60, 1200);
        hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, a, st + 2);
        hipError_t e = hipStreamWaitValue32(b, flag, 1, hipStreamWaitValueGte, 0xFFFFFFFFu);
        if (e != hipSuccess) { printf("hipStreamWaitValue32: %s\n", hipGetErrorString(e)); return 0; }
        hipLaunchKernelGGL(k_consumer, dim3(1), dim3(1), 0, b, data, out, st + 1);
        CK(hipDeviceSynchronize());
        double h; long long hs[3];
        CK(hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hs, st, 24, hipMemcpyDeviceToHost));
        printf("consumer saw %.1f; producer start -> consumer ran %.1f us; producer start -> producer end %.1f us (wall_clock64 at 100 MHz)\n", h,
               (hs[1] - hs[0]) / 100.0, (hs[2] - hs[0]) / 100.0);
    }
    return 0;
}
