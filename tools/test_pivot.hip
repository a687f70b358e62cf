// Stand-alone check of the sweep kernel's 128 x 128 pivot (pivot_chain, csrc/k_inverse.hip) against a host Gauss-Jordan inverse, for
// several leading dimensions and block positions, its time on an idle chip and the shader-clock stamps of its phases.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include -I gaussdca.jl_amd/csrc tools/test_pivot.hip -o tools/_bin/test_pivot
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#define GDCA_PIVOT_STAMPS
#include "../gaussdca.jl_amd/csrc/k_inverse.hip"

// as the persistent sweep kernel runs it: one 256-thread workgroup of a two-workgroups-per-CU kernel
__global__ __launch_bounds__(256, 2) void k_pivot4(const double *Ain, size_t ldin, double *Aout, size_t ldout, double *__restrict__ P,
                                                   size_t pld, int *bad)
{
    __shared__ __attribute__((aligned(16))) double buf[4 * KC * LDS_LD];
    if (threadIdx.x == 0) *bad = 0;
    __syncthreads();
    pivot_chain(Ain, ldin, Aout, ldout, P, pld, buf, bad, [] {});
}

static void print_stamps()
{
    long long h[9 * 8 * 12];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pivot_stamps), sizeof(h));
    auto at = [&](int K, int ph, int w) { return h[((K + 1) * 8 + ph) * 12 + w]; };
    const long long t0 = at(-1, 0, 3);
    printf("# pivot_chain, shader-clock cycles since the chain wave's first stamp.  chain: top, barrier 1, products done, barrier 2, micro-pivot done;"
           "  worker 0: top, barrier 1, Ns done, barrier 2, tiles done and next column staged\n");
    for (int K = -1; K < 8; ++K) {
        printf("K %2d chain:", K);
        for (int ph = 0; ph < 5; ++ph) printf(" %6lld", (K < 0 && ph >= 1 && ph <= 3) ? -1 : at(K, ph, 3) - t0);
        printf("   worker 0:");
        for (int ph = 0; ph < 6; ++ph)
            if (ph != 4) printf(" %6lld", (K < 0 && ph >= 1 && ph <= 3) ? -1 : at(K, ph, 0) - t0);
        printf("\n");
    }
}

static void host_inverse(const std::vector<double> &A, std::vector<double> &X, int n)
{
    std::vector<long double> M((size_t)n * 2 * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            M[(size_t)i * 2 * n + j] = A[(size_t)i + (size_t)j * n];
            M[(size_t)i * 2 * n + n + j] = i == j;
        }
    for (int k = 0; k < n; ++k) {
        const long double p = 1.0L / M[(size_t)k * 2 * n + k];
        for (int j = 0; j < 2 * n; ++j) M[(size_t)k * 2 * n + j] *= p;
        for (int i = 0; i < n; ++i)
            if (i != k) {
                const long double f = M[(size_t)i * 2 * n + k];
                for (int j = 0; j < 2 * n; ++j) M[(size_t)i * 2 * n + j] -= f * M[(size_t)k * 2 * n + j];
            }
    }
    X.resize((size_t)n * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) X[(size_t)i + (size_t)j * n] = (double)M[(size_t)i * 2 * n + n + j];
}

int main()
{
    const int n = 128;
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd;
    std::vector<double> B((size_t)n * 300), A0((size_t)n * n), Xr;
    for (auto &x : B) x = nd(rng);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0;
            for (int k = 0; k < 300; ++k) s += B[(size_t)i * 300 + k] * B[(size_t)j * 300 + k];
            A0[(size_t)i + (size_t)j * n] = s / 300 + (i == j ? 0.3 : 0.0);
        }
    host_inverse(A0, Xr, n);
    double xmax = 0;
    for (double x : Xr) xmax = std::fmax(xmax, std::fabs(x));
    int *dbad;
    (void)hipMalloc(&dbad, 4);
    int fails = 0;
    // in place inside a larger matrix (only the lower triangle of the block is given), P to a buffer of its own
    for (int ld : {128, 256, 384})
        for (int k = 0; k * 128 + 128 <= ld; ++k) {
            std::vector<double> A((size_t)ld * ld, 7.0), P((size_t)n * n, -3.0);
            for (int i = 0; i < n; ++i)
                for (int j = 0; j <= i; ++j) A[(size_t)(k * n + i) + (size_t)(k * n + j) * ld] = A0[(size_t)i + (size_t)j * n];
            double *dA, *dP;
            (void)hipMalloc(&dA, A.size() * 8);
            (void)hipMalloc(&dP, P.size() * 8);
            (void)hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
            (void)hipMemcpy(dP, P.data(), P.size() * 8, hipMemcpyHostToDevice);
            double *Akk = dA + (size_t)k * n + (size_t)k * n * ld;
            hipLaunchKernelGGL(k_pivot4, dim3(1), dim3(256), 0, 0, (const double *)Akk, (size_t)ld, Akk, (size_t)ld, dP, (size_t)n, dbad);
            hipError_t e = hipDeviceSynchronize();
            int hb = -1;
            (void)hipMemcpy(&hb, dbad, 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(A.data(), dA, A.size() * 8, hipMemcpyDeviceToHost);
            (void)hipMemcpy(P.data(), dP, P.size() * 8, hipMemcpyDeviceToHost);
            double eA = 0, eP = 0, outside = 0, asym = 0;
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    eA = std::fmax(eA, std::fabs(-A[(size_t)(k * n + i) + (size_t)(k * n + j) * ld] - Xr[(size_t)i + (size_t)j * n]));
                    eP = std::fmax(eP, std::fabs(P[(size_t)i + (size_t)j * n] - Xr[(size_t)i + (size_t)j * n]));
                    asym = std::fmax(asym, std::fabs(P[(size_t)i + (size_t)j * n] - P[(size_t)j + (size_t)i * n]));
                }
            for (int i = 0; i < ld; ++i)
                for (int j = 0; j < ld; ++j)
                    if (i / n != k || j / n != k) outside = std::fmax(outside, std::fabs(A[(size_t)i + (size_t)j * ld] - 7.0));
            const bool ok = e == hipSuccess && hb == 0 && eA / xmax < 1e-13 && eP / xmax < 1e-13 && outside == 0.0 && asym == 0.0;
            fails += !ok;
            printf("ld %d k %d: err %s bad %d  relerr Akk %.2e  P %.2e  asymmetry %.1e  touched outside %.1e  %s\n", ld, k, hipGetErrorString(e), hb,
                   eA / xmax, eP / xmax, asym, outside, ok ? "ok" : "FAIL");
            (void)hipFree(dA);
            (void)hipFree(dP);
        }
    {
        // not positive definite: leading minor 38 fails; then the launch time on an idle chip and the phase stamps
        std::vector<double> A(A0);
        A[37 + 37 * (size_t)n] = -1.0;
        double *dA, *dP, *dP2;
        (void)hipMalloc(&dA, A.size() * 8);
        (void)hipMalloc(&dP, A.size() * 8);
        (void)hipMalloc(&dP2, A.size() * 8);
        (void)hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_pivot4, dim3(1), dim3(256), 0, 0, (const double *)dA, (size_t)n, dP, (size_t)n, dP2, (size_t)n, dbad);
        int hb = -1;
        (void)hipMemcpy(&hb, dbad, 4, hipMemcpyDeviceToHost);
        printf("non-PD at local index 38: bad %d (want 38)  %s\n", hb, hb == 38 ? "ok" : "FAIL");
        fails += hb != 38;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipMemcpy(dA, A0.data(), A0.size() * 8, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0, 0);
            for (int it = 0; it < 50; ++it)
                hipLaunchKernelGGL(k_pivot4, dim3(1), dim3(256), 0, 0, (const double *)dA, (size_t)n, dP, (size_t)n, dP2, (size_t)n, dbad);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("pivot_chain (256 threads): %.1f us per launch (50 back-to-back launches, idle chip)\n", ms * 1000 / 50);
        }
        print_stamps();
    }
    printf(fails ? "FAILED\n" : "all ok\n");
    return fails != 0;
}
