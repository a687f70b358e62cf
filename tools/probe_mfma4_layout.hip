// Discover the lane layout of v_mfma_f64_4x4x4_4b_f64 by one-hot probing: for every (p, r) set A = e_p,
// B = e_r and record which output lanes become 1.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k(double *d)
{
    const int l = threadIdx.x;
    for (int p = 0; p < 64; ++p)
        for (int r = 0; r < 64; ++r) {
            const double a = (l == p) ? 1.0 : 0.0, b = (l == r) ? 1.0 : 0.0;
            d[((size_t)p * 64 + r) * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        }
}
int main()
{
    double *dd;
    CK(hipMalloc(&dd, 8ull * 64 * 64 * 64));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dd);
    std::vector<double> h(64 * 64 * 64);
    CK(hipMemcpy(h.data(), dd, 8ull * 64 * 64 * 64, hipMemcpyDeviceToHost));
    // for each output lane l: list of (p, r) pairs that contribute
    for (int l = 0; l < 64; ++l) {
        printf("D lane %2d <-", l);
        int cnt = 0;
        for (int p = 0; p < 64; ++p)
            for (int r = 0; r < 64; ++r)
                if (h[((size_t)p * 64 + r) * 64 + l] != 0.0) { printf(" (A%d,B%d)", p, r); ++cnt; }
        printf("  [%d]\n", cnt);
    }
    return 0;
}
