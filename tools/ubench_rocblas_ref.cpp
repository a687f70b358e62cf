// Calibration only (not used by the product): what do the vendor library's f64 kernels reach on the shapes of the
// SPD inverse's trailing update?  C (n x n, lower triangle) += G (n x k) * H^T, k = 128 / 256 / 512, as
// rocblas_dsyrkx (triangle only) and as a full rocblas_dgemm.  hipcc tools/ubench_rocblas_ref.cpp -lrocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { if ((x) != 0) { printf("error at line %d\n", __LINE__); return 1; } } while (0)

int main()
{
    rocblas_handle h;
    CK(rocblas_create_handle(&h));
    const int n = 10112;
    double *C, *G, *H;
    CK(hipMalloc(&C, sizeof(double) * (size_t)n * n));
    CK(hipMalloc(&G, sizeof(double) * (size_t)n * 512));
    CK(hipMalloc(&H, sizeof(double) * (size_t)n * 512));
    CK(hipMemset(C, 0, sizeof(double) * (size_t)n * n));
    std::vector<double> hg((size_t)n * 512);
    for (size_t i = 0; i < hg.size(); ++i) hg[i] = 1e-3 * (double)((i * 2654435761u) % 1000) - 0.5;
    CK(hipMemcpy(G, hg.data(), sizeof(double) * hg.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(H, hg.data(), sizeof(double) * hg.size(), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double one = 1.0;
    for (int k : {128, 256, 512}) {
        for (int mode = 0; mode < 2; ++mode) {
            const int reps = 10;
            for (int r = 0; r < reps + 2; ++r) {
                if (r == 2) CK(hipEventRecord(e0));
                if (mode == 0)
                    CK(rocblas_dsyrkx(h, rocblas_fill_lower, rocblas_operation_none, n, k, &one, G, n, H, n, &one, C, n));
                else
                    CK(rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_transpose, n, n, k, &one, G, n, H, n, &one, C, n));
            }
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double flops = (mode == 0 ? 1.0 : 2.0) * (double)n * n * k;  // triangle: n^2 k; full: 2 n^2 k
            printf("n=%d k=%3d %-8s %.3f ms per call  %.1f TFLOP/s\n", n, k, mode == 0 ? "dsyrkx" : "dgemm", ms / reps, flops / (ms / reps * 1e-3) / 1e12);
        }
    }
    return 0;
}
