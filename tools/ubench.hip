// Micro-benchmarks that informed the kernel designs (diagnostic; not part of libgdca.so).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench.hip -o gpurun_out/ubench && gpurun_out/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// ---- LDS atomics: 64 lanes, lane -> distinct consecutive slot (conflict-free), random row ----
template <typename T>
__global__ __launch_bounds__(512) void k_lds_atomic(T *out, int iters, int rows)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    T *h = reinterpret_cast<T *>(sm);
    for (int e = threadIdx.x; e < rows * 32; e += blockDim.x) h[e] = T(0);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int jl = lane & 31;
    unsigned rng = threadIdx.x * 2654435761u + 12345u + blockIdx.x;
    for (int it = 0; it < iters; ++it) {
        rng = rng * 1664525u + 1013904223u;
        const int row = (rng >> 8) % rows;
        atomicAdd(&h[row * 32 + jl], T(1));
    }
    __syncthreads();
    T acc = T(0);
    for (int e = threadIdx.x; e < rows * 32; e += blockDim.x) acc += h[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// ---- int8 MFMA issue-rate probe (32x32x32, i32 accumulate), NACC independent accumulators ----
typedef int int16_t_v __attribute__((ext_vector_type(16)));
typedef int int4_t_v __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k_probe_i8(int *out, int iters)
{
    int16_t_v acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    int4_t_v a = {(int)(threadIdx.x * 0x01010101u), 0x7f3c21e5, (int)0x81c3a55a, 0x12345678};
    int4_t_v b = {(int)(threadIdx.x * 0x03050709u), 0x0badf00d, (int)0xdeadbeef, 0x7e6d5c4b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
int run_probe_i8(int wps)
{
    const int nb = 256 * wps, iters = 40000 / NACC;
    int *out;
    CK(hipMalloc(&out, sizeof(int) * nb * 256));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_probe_i8<NACC>, dim3(nb), dim3(256), 0, 0, out, 100);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_probe_i8<NACC>, dim3(nb), dim3(256), 0, 0, out, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double macs = (double)nb * 4 * iters * NACC * 32768.0;
    printf("probe_i8 32x32x32 NACC=%d waves/SIMD=%d: %.3f ms, %.1f TMAC/s (= %.1f TOP/s), %.1f clk@2.4GHz per MFMA per SIMD\n", NACC, wps, ms,
           macs / ms / 1e9, 2 * macs / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)iters * NACC * wps));
    CK(hipFree(out));
    return 0;
}

// ---- tally-loop replica: per wave-instruction 2 LDS reads (b64 broadcast per half-wave + u8) and
// one ds_add_u64; `same_pct` = percent of instructions whose two half-waves use the same row set ----
__global__ __launch_bounds__(1024) void k_tally_like(unsigned long long *out, int iters, int same_pct, int with_reads)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    unsigned long long *h = reinterpret_cast<unsigned long long *>(sm);       // [400][32]
    unsigned long long *meta = h + 400 * 32 + 64;                              // [1024]
    unsigned char *zs = reinterpret_cast<unsigned char *>(meta + 1024);       // [1024][32]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int e = tid; e < 400 * 32 + 64; e += blockDim.x) h[e] = 0;
    unsigned rng = tid * 2654435761u + 12345u;
    // per-sequence a (1..20) and weight; per (seq, col) b (1..20)
    {
        unsigned r2 = (tid >> (same_pct >= 100 ? 9 : 0)) * 747796405u + 2891336453u;
        r2 ^= r2 >> 15;
        unsigned a = 1 + (r2 % 20);
        if (same_pct < 0 && ((r2 >> 12) % 100) < (unsigned)(-same_pct)) a = 3;  // skewed: dominant symbol
        if (same_pct > 0 && same_pct < 100 && (tid & 1) && ((r2 >> 8) % 100) < (unsigned)same_pct) a = 0x80;  // marker: copy partner
        meta[tid] = ((unsigned long long)a << 56) | (rng & 0xffffff);
        for (int c = 0; c < 32; ++c) {
            rng = rng * 1664525u + 1013904223u;
            zs[tid * 32 + c] = 1 + ((rng >> 10) % 20);
            if (same_pct < 0 && ((rng >> 20) % 100) < (unsigned)(-same_pct)) zs[tid * 32 + c] = 1 + (c % 20);  // conserved column
        }
    }
    __syncthreads();
    if (same_pct > 0 && same_pct < 100 && (tid & 1)) {
        if ((meta[tid] >> 56) == 0x80) {  // make sequence tid identical to tid-1 (same a, same b's)
            meta[tid] = meta[tid - 1];
            for (int c = 0; c < 32; ++c) zs[tid * 32 + c] = zs[(tid - 1) * 32 + c];
        }
    }
    __syncthreads();
    const int jl = lane & 31, sub = lane >> 5;
    for (int it = 0; it < iters; ++it) {
        const int kk = wv * 64 + ((it * 2) & 63) + sub;
        unsigned long long m;
        int b;
        if (with_reads) {
            m = meta[kk];
            b = zs[kk * 32 + jl];
        } else {
            m = ((unsigned long long)(1 + (it + sub * 7) % 20) << 56) | 5;
            b = 1 + ((it * 3 + lane) % 20);
        }
        const unsigned a = (unsigned)(m >> 56);
        const unsigned idx = ((a - 1) * 20 + (b - 1)) * 32 + jl;
        atomicAdd(&h[idx], m & 0xffffffull);
    }
    __syncthreads();
    unsigned long long acc = 0;
    for (int e = tid; e < 400 * 32; e += blockDim.x) acc += h[e];
    out[blockIdx.x * blockDim.x + tid] = acc;
}

int run_tally_like(int same_pct, int with_reads, int threads = 512)
{
    const int blocks = 256, iters = 20000;
    unsigned long long *out;
    CK(hipMalloc(&out, 8 * blocks * threads));
    size_t lds = 8 * (400 * 32 + 64) + 8 * 1024 + 1024 * 32;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tally_like), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_tally_like, dim3(blocks), dim3(threads), lds, 0, out, 100, same_pct, with_reads);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_tally_like, dim3(blocks), dim3(threads), lds, 0, out, iters, same_pct, with_reads);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("tally-like threads=%d same_pct=%3d reads=%d: %.3f ms, %.2f clk@2.4GHz per wave-instr per CU\n", threads, same_pct, with_reads, ms,
           ms * 1e-3 * 2.4e9 / ((double)iters * threads / 64));
    CK(hipFree(out));
    return 0;
}

// ---- f64 MFMA / VALU probes with clocks ----
__global__ __launch_bounds__(256) void k_probe(double *out, long long *clk, int iters, int mode)
{
    double4_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.5 + i;
    const long long t0 = clock64();
    const long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (mode == 0 || mode == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        if (mode == 1 || mode == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fma(v[i], a, b);
        }
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        clk[blockIdx.x * 2] = t1 - t0;
        clk[blockIdx.x * 2 + 1] = w1 - w0;
    }
}

// clean MFMA-only probe: NACC independent accumulators, no branches in the loop
template <int NACC>
__global__ __launch_bounds__(256) void k_probe2(double *out, long long *clk, int iters)
{
    double4_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = clock64();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int NACC>
int run_probe2(double *out, long long *clk, int wps, int ncu = 256)
{
    const int nb = ncu * wps, iters = 40000 / NACC;
    std::vector<long long> h(nb);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_probe2<NACC>, dim3(nb), dim3(256), 0, 0, out, clk, 100);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_probe2<NACC>, dim3(nb), dim3(256), 0, 0, out, clk, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), clk, sizeof(long long) * nb, hipMemcpyDeviceToHost));
    double cyc = 0;
    for (int i = 0; i < nb; ++i) cyc += h[i];
    cyc /= nb;
    const double fl = (double)nb * 4 * iters * NACC * 2048.0;
    printf("probe2 CUs=%d NACC=%d waves/SIMD=%d: %.3f ms %.1f TF; %.1f cycles per MFMA per wave => %.1f cycles per MFMA per SIMD\n", ncu, NACC, wps, ms,
           fl / ms / 1e9, cyc / ((double)iters * NACC), cyc / ((double)iters * NACC) / wps);
    return 0;
}

template <typename T>
int run_atomic(const char *name, int rows)
{
    const int blocks = 256, threads = 512, iters = 20000;
    T *out;
    CK(hipMalloc(&out, sizeof(T) * blocks * threads));
    size_t lds = sizeof(T) * rows * 32;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lds_atomic<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_lds_atomic<T>, dim3(blocks), dim3(threads), lds, 0, out, 100, rows);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_lds_atomic<T>, dim3(blocks), dim3(threads), lds, 0, out, iters, rows);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double adds = (double)blocks * threads * iters;
    printf("LDS atomicAdd %-8s rows=%4d: %.3f ms, %.2f G adds/s, %.2f clk@2.4GHz per wave-instr per CU\n", name, rows, ms,
           adds / ms / 1e6, ms * 1e-3 * 2.4e9 / ((double)iters * threads / 64));
    CK(hipFree(out));
    return 0;
}

int main()
{
    run_probe_i8<1>(1);
    run_probe_i8<2>(1);
    run_probe_i8<4>(1);
    run_probe_i8<2>(2);
    run_probe_i8<4>(2);
    run_probe_i8<2>(4);
    run_tally_like(0, 1, 1024);
    run_tally_like(-60, 1, 1024);
    run_tally_like(0, 0, 1024);
    run_tally_like(0, 0);
    run_tally_like(0, 1);
    run_tally_like(30, 1);
    run_tally_like(100, 1);
    run_tally_like(-30, 1);
    run_tally_like(-60, 1);
    run_tally_like(-90, 1);
    run_atomic<unsigned long long>("u64", 400);
    run_atomic<unsigned int>("u32", 400);
    run_atomic<unsigned int>("u32", 1200);
    run_atomic<double>("f64", 400);
    run_atomic<float>("f32", 400);
    run_atomic<int>("i32", 400);

    const int blocks = 1024;
    double *out;
    long long *clk;
    CK(hipMalloc(&out, sizeof(double) * 2048 * 256));
    CK(hipMalloc(&clk, sizeof(long long) * 2048 * 2));
    std::vector<long long> h(blocks * 2);
    for (int ncu = 16; ncu <= 256; ncu *= 2) run_probe2<4>(out, clk, 2, ncu);
    for (int wps = 1; wps <= 8; wps *= 2) {
        run_probe2<1>(out, clk, wps);
        run_probe2<2>(out, clk, wps);
        run_probe2<4>(out, clk, wps);
        if (wps <= 4) run_probe2<8>(out, clk, wps);
    }
    for (int wpc = 1; wpc <= 4; wpc *= 2) {
        for (int mode = 0; mode < 3; ++mode) {
            const int nb = 256 * wpc;
            const int iters = 20000;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            hipLaunchKernelGGL(k_probe, dim3(nb), dim3(256), 0, 0, out, clk, 100, mode);
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_probe, dim3(nb), dim3(256), 0, 0, out, clk, iters, mode);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), clk, sizeof(long long) * nb * 2, hipMemcpyDeviceToHost));
            double cyc = 0, wall = 0;
            for (int i = 0; i < nb; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
            cyc /= nb; wall /= nb;
            const double mf = (mode != 1) ? (double)nb * 4 * iters * 8.0 * 2048 : 0.0;
            const double vf = (mode != 0) ? (double)nb * 4 * iters * 32.0 * 128 : 0.0;
            printf("probe blocks/CU=%d mode=%d (0 mfma,1 valu,2 both): %.3f ms  mfma %.1f TF + valu %.1f TF; clk64 %.0f cycles, wall %.0f ticks(100MHz) => %.3f GHz; cycles per loop-iter %.1f\n",
                   wpc, mode, ms, mf / ms / 1e9, vf / ms / 1e9, cyc, wall, cyc / (wall * 10.0) , cyc / iters);
        }
    }
    return 0;
}
