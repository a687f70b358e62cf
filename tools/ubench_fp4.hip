// what does the fp4 matrix pipe of MI355X sustain when its operands come out of LDS?  The question behind it: the
// all-pairs reweighting as a class-one-hot lower bound -- symbols mapped to 8 classes (their three low bits), a one-hot fp4 image of the
// alignment with K = 8 N columns, X X^T tile products whose entries are the numbers of positions where two sequences' classes AGREE --
// is 5.0e12 multiply-accumulates at config C (M = 50 000, N = 500); the VALU bound form does it in 2.39 ms today.  The product would
// have to sustain >= 3.5 PMAC/s to be worth a kernel.
// v_mfma_scale_f32_32x32x64_f8f6f4 with both operands fp4 (E2M1; 1.0 = 0b0010): 32 x 32 x 64 MACs per instruction; 0/1 operands and
// counts <= 2^24 are exact in its f32 accumulators.  Variants: a wave's tile of BA x BB blocks of 32 x 32 (2 x 2, 2 x 4, 4 x 4:
// 64 .. 256 accumulator registers), operands in registers (the pipe's own rate) or re-read from LDS every 64-deep k step (16 bytes per
// lane and block, one barrier per k step: the staging barrier of a real kernel), one or two workgroups of four waves per compute unit.
// Checked: all-ones operands give 64 per k step in every accumulator.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int BA, int BB, bool LDS>
__global__ __launch_bounds__(256, (BA * BB > 8) ? 1 : 2) void k_fp4(float *out, unsigned long long *clk, int ksteps)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // a wave's operand blocks of one k step: (BA + BB) x 1 KB, [block][lane] x 16 bytes
    unsigned char *mine = lds + (size_t)wv * (BA + BB) * 1024;
    if (LDS) {
        for (int e = lane; e < (BA + BB) * 64; e += 64) reinterpret_cast<v4i *>(mine)[e] = (v4i){0x22222222, 0x22222222, 0x22222222, 0x22222222};  // 32 x fp4 1.0
        __syncthreads();
    }
    v16f acc[BA][BB];
#pragma unroll
    for (int i = 0; i < BA; ++i)
#pragma unroll
        for (int j = 0; j < BB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    v8i a[BA], b[BB];
#pragma unroll
    for (int i = 0; i < BA; ++i) a[i] = (v8i){0x22222222, 0x22222222, 0x22222222, 0x22222222, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < BB; ++j) b[j] = (v8i){0x22222222, 0x22222222, 0x22222222, 0x22222222, 0, 0, 0, 0};
    const int one = 0x7f7f7f7f;  // E8M0 scale 2^0
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int ks = 0; ks < ksteps; ++ks) {
        if (LDS) {
#pragma unroll
            for (int i = 0; i < BA; ++i) {
                const v4i x = *reinterpret_cast<const v4i *>(mine + ((i * 64 + lane) << 4));
                a[i] = (v8i){x[0], x[1], x[2], x[3], 0, 0, 0, 0};
            }
#pragma unroll
            for (int j = 0; j < BB; ++j) {
                const v4i x = *reinterpret_cast<const v4i *>(mine + (((BA + j) * 64 + lane) << 4));
                b[j] = (v8i){x[0], x[1], x[2], x[3], 0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int i = 0; i < BA; ++i)
#pragma unroll
            for (int j = 0; j < BB; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[j], acc[i][j], 4, 4, 0, one, 0, one);
        if (LDS) __syncthreads();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f, mn = 1e30f, mx = -1e30f;
#pragma unroll
    for (int i = 0; i < BA; ++i)
#pragma unroll
        for (int j = 0; j < BB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s += acc[i][j][e];
                mn = fminf(mn, acc[i][j][e]);
                mx = fmaxf(mx, acc[i][j][e]);
            }
    out[(size_t)blockIdx.x * 256 + tid] = s;
    if (tid == 0) {
        clk[4 * blockIdx.x] = c1 - c0;
        clk[4 * blockIdx.x + 1] = r1 - r0;
        reinterpret_cast<float *>(clk + 4 * blockIdx.x + 2)[0] = mn;
        reinterpret_cast<float *>(clk + 4 * blockIdx.x + 2)[1] = mx;
    }
}

template <class K>
int run(const char *name, K kern, int ba, int bb, bool lds, int wgs, float *out, unsigned long long *clk)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int nb = 256 * wgs, ksteps = 100000;
    const size_t shm = lds ? (size_t)4 * (ba + bb) * 1024 : 0;
    hipLaunchKernelGGL(kern, dim3(nb), dim3(256), shm, 0, out, clk, 1000);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(nb), dim3(256), shm, 0, out, clk, ksteps);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    static unsigned long long hc[4096];
    CK(hipMemcpy(hc, clk, sizeof(unsigned long long) * 4 * nb, hipMemcpyDeviceToHost));
    double ghz = 0;
    for (int b = 0; b < nb; ++b) ghz += (double)hc[4 * b] / (double)hc[4 * b + 1] * 0.1;
    ghz /= nb;
    const float *mm = reinterpret_cast<const float *>(hc + 2);
    const double macs = (double)nb * 4 * ksteps * ba * bb * 32.0 * 32.0 * 64.0;
    printf("%-52s %d wg/CU: %.2f PMAC/s (%.2f PFLOP/s) at %.2f GHz, %.1f clk per MFMA per SIMD; accumulators %s (%.0f .. %.0f, expected %.0f)\n", name, wgs,
           macs / (ms * 1e-3) / 1e15, 2 * macs / (ms * 1e-3) / 1e15, ghz, ghz * 1e9 * (ms * 1e-3) / ((double)ksteps * ba * bb * wgs),
           (mm[0] == 64.0f * ksteps && mm[1] == 64.0f * ksteps) ? "exact" : (ksteps * 64.0 > 16777216.0 ? "beyond 2^24 (as expected for this many steps)" : "WRONG"), mm[0], mm[1], 64.0 * ksteps);
    return 0;
}

int main()
{
    float *out;
    unsigned long long *clk;
    CK(hipMalloc(&out, sizeof(float) * 512 * 256));
    CK(hipMalloc(&clk, sizeof(unsigned long long) * 4096));
    // exactness first: 1000 steps of all-ones = 64 000 in every accumulator
    hipLaunchKernelGGL((k_fp4<2, 2, false>), dim3(1), dim3(256), 0, 0, out, clk, 1000);
    CK(hipDeviceSynchronize());
    unsigned long long h[4];
    CK(hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost));
    printf("check: 1000 k steps of all-ones fp4 operands: accumulators %.0f .. %.0f (expected 64000)\n", reinterpret_cast<float *>(h + 2)[0], reinterpret_cast<float *>(h + 2)[1]);
    for (int wgs = 1; wgs <= 2; ++wgs) {
        run("registers, wave tile 64 x 64 (2 x 2 blocks)", k_fp4<2, 2, false>, 2, 2, false, wgs, out, clk);
        run("registers, wave tile 64 x 128 (2 x 4 blocks)", k_fp4<2, 4, false>, 2, 4, false, wgs, out, clk);
        run("LDS-fed, wave tile 64 x 64 (2 x 2 blocks)", k_fp4<2, 2, true>, 2, 2, true, wgs, out, clk);
        run("LDS-fed, wave tile 64 x 128 (2 x 4 blocks)", k_fp4<2, 4, true>, 2, 4, true, wgs, out, clk);
    }
    run("registers, wave tile 128 x 128 (4 x 4 blocks)", k_fp4<4, 4, false>, 4, 4, false, 1, out, clk);
    run("LDS-fed, wave tile 128 x 128 (4 x 4 blocks)", k_fp4<4, 4, true>, 4, 4, true, 1, out, clk);
    return 0;
}
