#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double double4_t __attribute__((ext_vector_type(4)));
// MODE 0: 4x4x4, constant operands; 1: 4x4x4, 16 A x 4 B distinct random operands (64 accumulators, like the tile
// kernel); 2: 16x16x4 with 4 A x 4 B distinct random operands (16 accumulators)
template <int MODE>
__global__ __launch_bounds__(256) void k4(double *out, const double *in, int iters)
{
    double s = 0;
    if (MODE == 0) {
        double acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = 0.0;
        double a = in[threadIdx.x], b = in[threadIdx.x + 256];
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < 8; ++i) s += acc[i];
    } else if (MODE == 1) {
        double acc[16][4], a[16], b[4];
        for (int i = 0; i < 16; ++i) { a[i] = in[threadIdx.x + 64 * i]; for (int j = 0; j < 4; ++j) acc[i][j] = 0.0; }
        for (int j = 0; j < 4; ++j) b[j] = in[2048 + threadIdx.x + 64 * j];
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
    } else {
        double4_t acc[4][4];
        double a[4], b[4];
        for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x + 64 * i]; b[i] = in[2048 + threadIdx.x + 64 * i]; for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0, 0, 0, 0}; }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
int run(double *out, const double *in, int wps, const char *name)
{
    const int nb = 256 * wps;
    const int per_iter = MODE == 0 ? 8 : (MODE == 1 ? 64 : 16);
    const double fl_per = MODE == 2 ? 2048.0 : 512.0;
    const int iters = (MODE == 2 ? 40000 : 160000) / per_iter;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k4<MODE>, dim3(nb), dim3(256), 0, 0, out, in, 10);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k4<MODE>, dim3(nb), dim3(256), 0, 0, out, in, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double fl = (double)nb * 4 * iters * per_iter * fl_per;
    printf("%-44s waves/SIMD=%d: %.3f ms %.1f TF\n", name, wps, ms, fl / ms / 1e9);
    return 0;
}
int main()
{
    double *out, *in;
    CK(hipMalloc(&out, 8 * 4096 * 256));
    CK(hipMalloc(&in, 8 * 8192));
    double h[8192];
    unsigned long long x = 88172645463325252ull;
    for (int i = 0; i < 8192; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = ((double)(x >> 11) / 9007199254740992.0 - 0.5) * 1e-3; }
    CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    for (int wps = 1; wps <= 2; ++wps) {
        run<0>(out, in, wps, "4x4x4 constant operands, 8 acc");
        run<1>(out, in, wps, "4x4x4 random 16 A x 4 B operands, 64 acc");
        run<2>(out, in, wps, "16x16x4 random 4 A x 4 B operands, 16 acc");
    }
    return 0;
}
