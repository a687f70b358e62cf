// Calibration only (not used by the product): the vendor library's SPD inverse, rocsolver_dpotrf + rocsolver_dpotri,
// on a covariance-like matrix of the headline size (n = 10000) and of config D (n = 20000).
// hipcc tools/ubench_rocsolver_ref.cpp -lrocsolver -lrocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { if ((x) != 0) { printf("error at line %d\n", __LINE__); return 1; } } while (0)

__global__ void k_fill(double *A, int n)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)n * n) return;
    const int r = (int)(e % n), c = (int)(e / n);
    const unsigned h = (unsigned)(r ^ c) * 2654435761u + (unsigned)(r + c) * 40503u;
    double v = 1e-3 * ((double)(h % 1000) / 1000.0 - 0.5);   // small symmetric-ish noise ((r^c),(r+c) are symmetric)
    A[e] = (r == c) ? 1.0 : v;
}

int main()
{
    rocblas_handle h;
    CK(rocblas_create_handle(&h));
    for (int n : {10000, 20000}) {
        double *A;
        int *info;
        CK(hipMalloc(&A, sizeof(double) * (size_t)n * n));
        CK(hipMalloc(&info, sizeof(int)));
        hipEvent_t e0, e1, e2;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventCreate(&e2));
        float best = 1e30f, bf = 0, bi = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k_fill, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, 0, A, n);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            CK(rocsolver_dpotrf(h, rocblas_fill_lower, n, A, n, info));
            CK(hipEventRecord(e1));
            CK(rocsolver_dpotri(h, rocblas_fill_lower, n, A, n, info));
            CK(hipEventRecord(e2));
            CK(hipDeviceSynchronize());
            float f, i;
            CK(hipEventElapsedTime(&f, e0, e1));
            CK(hipEventElapsedTime(&i, e1, e2));
            if (f + i < best) { best = f + i; bf = f; bi = i; }
        }
        int hi = -1;
        CK(hipMemcpy(&hi, info, sizeof(int), hipMemcpyDeviceToHost));
        const double flops = (double)n * n * n;  // n^3/3 + 2 n^3/3
        printf("n=%d: dpotrf %.1f ms + dpotri %.1f ms = %.1f ms  (%.1f TFLOP/s on the n^3 model), info=%d\n", n, bf, bi, best,
               flops / (best * 1e-3) / 1e12, hi);
        CK(hipFree(A));
        CK(hipFree(info));
    }
    return 0;
}
