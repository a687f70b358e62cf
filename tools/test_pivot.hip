// Stand-alone check of k_pivot (csrc/k_inverse.hip) against a host Gauss-Jordan inverse, for several leading
// dimensions and block positions.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include -I gaussdca.jl_amd/csrc tools/test_pivot.hip
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#define GDCA_PIVOT_STAMPS
#include "../gaussdca.jl_amd/csrc/k_inverse.hip"

// the form the persistent sweep kernel runs: one 256-thread workgroup, the serial chain on a wave of its own
__global__ __launch_bounds__(256, 2) void k_pivot4(const double *Ain, size_t ldin, double *Aout, size_t ldout, double *__restrict__ P,
                                                   size_t pld, int *bad)
{
    __shared__ __attribute__((aligned(16))) double buf[4 * KC * LDS_LD];
    pivot_chain(Ain, ldin, Aout, ldout, P, pld, buf, bad);
}

static void print_stamps(int nw)
{
    long long h[9 * 8 * 12];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pivot_stamps), sizeof(h));
    auto at = [&](int K, int ph, int w) { return h[((K + 1) * 8 + ph) * 12 + w]; };
    if (nw == 12) {
        const long long t0 = at(-1, 0, 0);
        printf("# k_pivot (12 waves), shader-clock cycles: micro-block, top of the loop, end of its update phase (wave 0)\n");
        for (int K = 0; K < 8; ++K) printf("K %d  %6lld %6lld\n", K, at(K, 0, 0) - t0, at(K, 7, 0) - t0);
        return;
    }
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
    gdca_dev_scalars *sc;
    hipMalloc(&sc, sizeof(*sc));
    for (int ld : {128, 256, 384})
        for (int k = 0; k * 128 + 128 <= ld; ++k) {
            std::vector<double> A((size_t)ld * ld, 7.0), P((size_t)n * n, -3.0);
            for (int i = 0; i < n; ++i)
                for (int j = 0; j <= i; ++j) A[(size_t)(k * n + i) + (size_t)(k * n + j) * ld] = A0[(size_t)i + (size_t)j * n];  // lower only
            double *dA, *dP;
            hipMalloc(&dA, A.size() * 8);
            hipMalloc(&dP, P.size() * 8);
            hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
            hipMemcpy(dP, P.data(), P.size() * 8, hipMemcpyHostToDevice);
            hipMemset(sc, 0, sizeof(*sc));
            double *Akk = dA + (size_t)k * n + (size_t)k * n * ld;
            hipLaunchKernelGGL(k_pivot, dim3(1), dim3(PIVOT_THREADS), 0, 0, (const double *)Akk, (size_t)ld, Akk, (size_t)ld, dP,
                               (size_t)n, sc, k * n, ld);
            hipError_t e = hipDeviceSynchronize();
            gdca_dev_scalars h;
            hipMemcpy(&h, sc, sizeof(h), hipMemcpyDeviceToHost);
            hipMemcpy(A.data(), dA, A.size() * 8, hipMemcpyDeviceToHost);
            hipMemcpy(P.data(), dP, P.size() * 8, hipMemcpyDeviceToHost);
            double eA = 0, eP = 0, outside = 0;
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    eA = std::fmax(eA, std::fabs(-A[(size_t)(k * n + i) + (size_t)(k * n + j) * ld] - Xr[(size_t)i + (size_t)j * n]));
                    eP = std::fmax(eP, std::fabs(P[(size_t)i + (size_t)j * n] - Xr[(size_t)i + (size_t)j * n]));
                }
            for (int i = 0; i < ld; ++i)
                for (int j = 0; j < ld; ++j)
                    if (i / n != k || j / n != k) outside = std::fmax(outside, std::fabs(A[(size_t)i + (size_t)j * ld] - 7.0));
            printf("ld %d k %d: err %s info %d  relerr Akk %.2e  P %.2e  touched outside %.1e\n", ld, k, hipGetErrorString(e), h.info,
                   eA / xmax, eP / xmax, outside);
            hipFree(dA);
            hipFree(dP);
        }
    {
        // not positive definite: leading minor 38 fails; and the launch time on an idle chip
        std::vector<double> A(A0);
        A[37 + 37 * (size_t)n] = -1.0;
        double *dA, *dP;
        hipMalloc(&dA, A.size() * 8);
        hipMalloc(&dP, A.size() * 8);
        double *dP2;
        hipMalloc(&dP2, A.size() * 8);
        hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
        hipMemset(sc, 0, sizeof(*sc));
        hipLaunchKernelGGL(k_pivot, dim3(1), dim3(PIVOT_THREADS), 0, 0, (const double *)dA, (size_t)n, dA, (size_t)n, dP, (size_t)n, sc,
                           256, 100000);
        gdca_dev_scalars h;
        hipMemcpy(&h, sc, sizeof(h), hipMemcpyDeviceToHost);
        printf("non-PD at local index 38, index0 256: info %d (want 294)\n", h.info);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            hipMemcpy(dA, A0.data(), A0.size() * 8, hipMemcpyHostToDevice);
            hipEventRecord(e0, 0);
            for (int it = 0; it < 50; ++it)  // the block is overwritten by -inverse: values stay finite under repetition? use fresh copies
                hipLaunchKernelGGL(k_pivot, dim3(1), dim3(PIVOT_THREADS), 0, 0, (const double *)dA, (size_t)n, dP + 0, (size_t)n, dP2,
                                   (size_t)n, sc, 0, 0);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("k_pivot: %.1f us per launch (50 back-to-back launches, idle chip)\n", ms * 1000 / 50);
        }
        print_stamps(12);
        int *dbad;
        hipMalloc(&dbad, 4);
        hipMemset(dbad, 0, 4);
        for (int rep = 0; rep < 3; ++rep) {
            hipMemcpy(dA, A0.data(), A0.size() * 8, hipMemcpyHostToDevice);
            hipEventRecord(e0, 0);
            for (int it = 0; it < 50; ++it)
                hipLaunchKernelGGL(k_pivot4, dim3(1), dim3(256), 0, 0, (const double *)dA, (size_t)n, dP + 0, (size_t)n, dP2, (size_t)n, dbad);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("k_pivot4 (256 threads): %.1f us per launch (50 back-to-back launches, idle chip)\n", ms * 1000 / 50);
        }
        print_stamps(4);
        {
            // the 256-thread form is correct too
            hipMemcpy(dA, A0.data(), A0.size() * 8, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k_pivot4, dim3(1), dim3(256), 0, 0, (const double *)dA, (size_t)n, dP + 0, (size_t)n, dP2, (size_t)n, dbad);
            std::vector<double> P((size_t)n * n);
            hipMemcpy(P.data(), dP2, P.size() * 8, hipMemcpyDeviceToHost);
            double eP = 0;
            for (size_t i = 0; i < P.size(); ++i) eP = std::fmax(eP, std::fabs(P[i] - Xr[i]));
            std::vector<double> Am((size_t)n * n);
            hipMemcpy(Am.data(), dP, Am.size() * 8, hipMemcpyDeviceToHost);
            double eA = 0, asym = 0;
            for (size_t i = 0; i < Am.size(); ++i) eA = std::fmax(eA, std::fabs(-Am[i] - Xr[i]));
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) asym = std::fmax(asym, std::fabs(P[(size_t)i + (size_t)j * n] - P[(size_t)j + (size_t)i * n]));
            int hb = -1;
            hipMemcpy(&hb, dbad, 4, hipMemcpyDeviceToHost);
            printf("k_pivot4 relerr P %.2e  Aout %.2e  asymmetry %.1e  bad %d (want 0)\n", eP / xmax, eA / xmax, asym, hb);
            // not positive definite: leading minor 38 fails
            std::vector<double> Ab(A0);
            Ab[37 + 37 * (size_t)n] = -1.0;
            hipMemcpy(dA, Ab.data(), Ab.size() * 8, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k_pivot4, dim3(1), dim3(256), 0, 0, (const double *)dA, (size_t)n, dP + 0, (size_t)n, dP2, (size_t)n, dbad);
            hipMemcpy(&hb, dbad, 4, hipMemcpyDeviceToHost);
            printf("k_pivot4 non-PD at local index 38: bad %d (want 38)\n", hb);
            // in place, inside a larger matrix (ld 384, block 1)
            const int ld = 384;
            std::vector<double> Big((size_t)ld * ld, 7.0);
            for (int i = 0; i < n; ++i)
                for (int j = 0; j <= i; ++j) Big[(size_t)(n + i) + (size_t)(n + j) * ld] = A0[(size_t)i + (size_t)j * n];
            double *dBig;
            hipMalloc(&dBig, Big.size() * 8);
            hipMemcpy(dBig, Big.data(), Big.size() * 8, hipMemcpyHostToDevice);
            hipMemset(dbad, 0, 4);
            double *Akk = dBig + n + (size_t)n * ld;
            hipLaunchKernelGGL(k_pivot4, dim3(1), dim3(256), 0, 0, (const double *)Akk, (size_t)ld, Akk, (size_t)ld, dP2, (size_t)n, dbad);
            hipMemcpy(Big.data(), dBig, Big.size() * 8, hipMemcpyDeviceToHost);
            double e2 = 0, outside = 0;
            for (int i = 0; i < ld; ++i)
                for (int j = 0; j < ld; ++j) {
                    if (i / n == 1 && j / n == 1)
                        e2 = std::fmax(e2, std::fabs(-Big[(size_t)i + (size_t)j * ld] - Xr[(size_t)(i - n) + (size_t)(j - n) * n]));
                    else
                        outside = std::fmax(outside, std::fabs(Big[(size_t)i + (size_t)j * ld] - 7.0));
                }
            printf("k_pivot4 in place (ld 384, block 1): relerr %.2e  touched outside %.1e\n", e2 / xmax, outside);
        }
    }
    return 0;
}
