// Feasibility probe for an int8-sliced (Ozaki-style) K = 256 trailing-update tile on MI355X: the MFMA and LDS
// operand-read pattern of one 128 x 128 tile (4 waves, 64 x 64 per wave = 2 x 2 blocks of v_mfma_i32_32x32x32_i8),
// order loop outside: for order d = 0..7, for k-step = 0..7, for s = 0..d: read A slice s (2 blocks) and B slice
// d - s (2 blocks) from LDS, 4 MFMAs; after each order 64 int32 -> f64 conversions + FMAs.  No global traffic and
// no meaningful numbers (LDS holds arbitrary bytes): this measures whether the LDS feed keeps the int8 pipe busy.
// Prints microseconds per tile per CU with 2 workgroups per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// LDS per workgroup: one k-step (32 deep) of all 8 slices of A (128 rows) and B (128 rows): 2 x 8 x 128 x 32 B = 64 KB
__global__ __launch_bounds__(256, 2) void k_tile(double *out, int tiles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wr = wv & 1, wc = wv >> 1;
    for (int e = tid; e < 65536 / 4; e += 256) reinterpret_cast<int *>(lds)[e] = e * 2654435761u;
    __syncthreads();
    // operand addressing: slice s, row block rb (32 rows), lane -> 16 bytes: [s][rb][lane] -> 1 KB per block
    auto opA = [&](int s, int rb) { return *reinterpret_cast<const v4i *>(lds + (((s * 4 + rb) * 64 + lane) << 4)); };
    auto opB = [&](int s, int rb) { return *reinterpret_cast<const v4i *>(lds + 32768 + (((s * 4 + rb) * 64 + lane) << 4)); };
    double facc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) facc[i] = 0.0;
    for (int t = 0; t < tiles; ++t) {
#pragma unroll 1
        for (int d = 0; d < 8; ++d) {
            v16i acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
#pragma unroll 1
            for (int ks = 0; ks < 8; ++ks) {
                __syncthreads();  // stands for the per-k-step staging barrier
                for (int s = 0; s <= d; ++s) {
                    const v4i a0 = opA(s, wr * 2), a1 = opA(s, wr * 2 + 1);
                    const v4i b0 = opB(d - s, wc * 2), b1 = opB(d - s, wc * 2 + 1);
                    acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, acc[1][1], 0, 0, 0);
                }
            }
            const double w = 1.0 / (double)(1 << (d + 1));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) facc[(i * 2 + j) * 16 + e] = fma((double)acc[i][j][e], w, facc[(i * 2 + j) * 16 + e]);
        }
    }
    double ssum = 0;
#pragma unroll
    for (int i = 0; i < 64; ++i) ssum += facc[i];
    out[(size_t)blockIdx.x * 256 + tid] = ssum;
}

int main()
{
    double *out;
    CK(hipMalloc(&out, sizeof(double) * 256 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tile), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
        const int nb = 256 * wgs_per_cu, tiles = 20;
        hipLaunchKernelGGL(k_tile, dim3(nb), dim3(256), 65536, 0, out, 2);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_tile, dim3(nb), dim3(256), 65536, 0, out, tiles);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us_per_tile_per_cu = ms * 1e3 / tiles / wgs_per_cu;
        const double macs = 36.0 * 8 * 4 * 32768.0 * 4;  // per tile: pairs x k-steps x blocks x MACs x waves
        printf("workgroups/CU=%d: %.3f ms for %d tiles per workgroup -> %.1f us per tile per CU (f64 tile today: 42.5), int8 pipe %.2f PMAC/s chip-wide\n",
               wgs_per_cu, ms, tiles, us_per_tile_per_cu, macs * nb * tiles / (ms * 1e-3) / 1e15);
    }
    return 0;
}
