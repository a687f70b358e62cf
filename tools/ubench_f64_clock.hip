// What clock does the chip hold under a saturating v_mfma_f64_16x16x4_f64 load?  Each workgroup (4 waves, 16
// independent accumulators per wave, register operands) reads the shader clock counter (s_memtime) and the
// constant 100 MHz counter (s_memrealtime) before and after its loop: shader clock = d(memtime) / d(realtime) * 100 MHz.
// Printed next to the achieved TFLOP/s and the TFLOP/s the same issue rate would give at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void k_load(double *out, unsigned long long *clk, const double *in, int iters)
{
    double4_t acc[4][4];
    double a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = in[threadIdx.x + 64 * i];
        b[i] = in[2048 + threadIdx.x + 64 * i];
        for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0, 0, 0, 0};
    }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = c1 - c0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

// The same loop with 8 accumulators and a 4-waves-per-SIMD register budget, to see what 3 and 4 MFMA-issuing waves per
// SIMD do to the matrix pipe (the trailing update shares its SIMDs with the waves of the look-ahead chain).
__global__ __launch_bounds__(256, 4) void k_load_occ(double *out, unsigned long long *clk, const double *in, int iters)
{
    double4_t acc[2][4];
    double a[2], b[4];
    for (int i = 0; i < 4; ++i) b[i] = in[2048 + threadIdx.x + 64 * i];
    for (int i = 0; i < 2; ++i) {
        a[i] = in[threadIdx.x + 64 * i];
        for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0, 0, 0, 0};
    }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = c1 - c0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

// Same MFMA stream, operands re-read from LDS every k4 step exactly as the trailing-update kernel does (8 ds_read_b64,
// wait, 16 MFMAs), optional barrier every 4 steps (one 16-deep chunk); still no global traffic.
template <bool BARRIER>
__global__ __launch_bounds__(256, 2) void k_load_lds(double *out, unsigned long long *clk, const double *in, int iters)
{
    __shared__ __attribute__((aligned(16))) double Gs[16][144];
    __shared__ __attribute__((aligned(16))) double Hs[16][144];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wr = wv & 1, wc = wv >> 1, l15 = lane & 15, lq = lane >> 4;
    for (int e = tid; e < 16 * 144; e += 256) {
        (&Gs[0][0])[e] = in[e & 4095];
        (&Hs[0][0])[e] = in[(e * 7) & 4095];
    }
    __syncthreads();
    double4_t acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0, 0, 0, 0};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int k4 = 0; k4 < 16; k4 += 4) {
            double a[4], b[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] = Hs[k4 + lq][wc * 64 + t * 16 + l15];
#pragma unroll
            for (int t = 0; t < 4; ++t) b[t] = Gs[k4 + lq][wr * 64 + t * 16 + l15];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (BARRIER) __syncthreads();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = c1 - c0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <class K>
int run_variant(const char *name, K kern, double *out, unsigned long long *clk, const double *in)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int wgs = 1; wgs <= 2; ++wgs) {
        const int nb = 256 * wgs, iters = 40000;
        hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, 0, out, clk, in, 1000);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, 0, out, clk, in, iters);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long hc[1024];
        CK(hipMemcpy(hc, clk, sizeof(unsigned long long) * 2 * nb, hipMemcpyDeviceToHost));
        double ghz = 0;
        for (int b = 0; b < nb; ++b) ghz += (double)hc[2 * b] / (double)hc[2 * b + 1] * 0.1;
        ghz /= nb;
        const double tf = (double)nb * 4 * iters * 16 * 2048.0 / (ms * 1e-3) / 1e12;
        printf("%-34s %d workgroup(s) per CU: %.1f TFLOP/s at %.2f GHz\n", name, wgs, tf, ghz);
    }
    return 0;
}

int main()
{
    double *out, *in;
    unsigned long long *clk;
    CK(hipMalloc(&out, sizeof(double) * 256 * 1024));
    CK(hipMalloc(&in, sizeof(double) * 4096));
    CK(hipMalloc(&clk, sizeof(unsigned long long) * 2048));
    double h[4096];
    // GDCA_UBENCH_RANDOM=1: full-entropy mantissas (a 64-bit LCG) instead of 1000 distinct three-digit values: the
    // f64 matrix pipe draws more power on them
    const bool rnd = getenv("GDCA_UBENCH_RANDOM") != nullptr;
    unsigned long long st = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < 4096; ++i) {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        h[i] = rnd ? ((double)(st >> 11) / 9007199254740992.0 - 0.5) : (1e-3 * ((i * 2654435761u) % 1000) - 0.5);
    }
    printf("operands: %s\n", rnd ? "random mantissas" : "three-digit values");
    CK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int wgs = 1; wgs <= 2; ++wgs)
        for (int cus = 32; cus <= 256; cus *= 2) {
            const int nb = cus * wgs, iters = 40000;
            hipLaunchKernelGGL(k_load, dim3(nb), dim3(256), 0, 0, out, clk, in, 1000);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_load, dim3(nb), dim3(256), 0, 0, out, clk, in, iters);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long hc[1024];
            CK(hipMemcpy(hc, clk, sizeof(unsigned long long) * 2 * nb, hipMemcpyDeviceToHost));
            double ghz = 0;
            for (int b = 0; b < nb; ++b) ghz += (double)hc[2 * b] / (double)hc[2 * b + 1] * 0.1;
            ghz /= nb;
            const double flops = (double)nb * 4 * iters * 16 * 2048.0;
            const double tf = flops / (ms * 1e-3) / 1e12;
            // busy clocks per MFMA per SIMD at the measured clock (waves per SIMD = wgs when cus == 256)
            printf("workgroups=%4d (%d per CU on %3d CUs): %.2f ms  %.1f TFLOP/s  shader clock %.2f GHz  -> %.1f clk per MFMA per SIMD\n",
                   nb, wgs, cus, ms, tf, ghz, ghz * 1e9 * (ms * 1e-3) / ((double)iters * 16 * wgs));
        }
    for (int wgs = 1; wgs <= 4; ++wgs) {
        const int nb = 256 * wgs, iters = 40000;
        hipLaunchKernelGGL(k_load_occ, dim3(nb), dim3(256), 0, 0, out, clk, in, 1000);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_load_occ, dim3(nb), dim3(256), 0, 0, out, clk, in, iters);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("8 accumulators per wave, %d wave(s) per SIMD: %.1f TFLOP/s\n", wgs, (double)nb * 4 * iters * 8 * 2048.0 / (ms * 1e-3) / 1e12);
    }
    run_variant("LDS-fed k4 steps, no barrier", k_load_lds<false>, out, clk, in);
    run_variant("LDS-fed k4 steps, barrier per chunk", k_load_lds<true>, out, clk, in);
    return 0;
}
