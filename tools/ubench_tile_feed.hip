// "settle the tile question with a measurement, not prose".  What would a 256 x 128 tile buy the sweep's tile items?
// Per flop it moves 0.75 x the panel bytes of the 128 x 128 tile through L2 -> registers -> LDS (the C tile and the fragment reads of
// a wave are the same).  This loop is the tile item's chunk loop stripped to what matters for that question: per 16-deep chunk every wave
// issues 64 v_mfma_f64_16x16x4_f64 on fragments read from LDS (8 ds_read_b64 per k4 step, one barrier per chunk, LDS double-buffered),
// and the workgroup STAGES Q/4 x 32 KB of panel data per chunk: global_load_dwordx4 from a 64-MB region that lives in L2 / Infinity Cache
// (as the sweep's panels do), ds_write_b128 into the other LDS buffer.  Q = 4: the 128 x 128 tile's traffic; Q = 3: the 256 x 128 tile's;
// Q = 0: none.  Two workgroups per compute unit, every compute unit, 0.5 s per variant; printed: TFLOP/s, the shader clock the chip held
// (s_memtime / s_memrealtime) and the matrix-pipe clocks per MFMA and SIMD.  If Q = 3 does not beat Q = 4 by more than a per cent in
// clock or rate, the bigger tile cannot win through operand movement.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int Q>
__global__ __launch_bounds__(256, 2) void k_feed(double *out, unsigned long long *clk, const double *panels, size_t panel_doubles, int chunks)
{
    __shared__ __attribute__((aligned(16))) double Gs[2][16][144];
    __shared__ __attribute__((aligned(16))) double Hs[2][16][144];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wr = wv & 1, wc = wv >> 1, l15 = lane & 15, lq = lane >> 4;
    for (int e = tid; e < 2 * 16 * 144; e += 256) {
        (&Gs[0][0][0])[e] = panels[e & 4095];
        (&Hs[0][0][0])[e] = panels[(e * 7) & 4095];
    }
    __syncthreads();
    double4_t acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0, 0, 0, 0};
    // this workgroup's stream of panel data: 32 KB per chunk at Q = 4, contiguous 16-byte pieces per thread
    size_t off = ((size_t)blockIdx.x * 977u * 4096u) % panel_doubles;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int ch = 0; ch < chunks; ++ch) {
        const int cur = ch & 1, nxt = cur ^ 1;
        double2 st[2 * Q > 0 ? 2 * Q : 1];
        if (Q > 0) {
#pragma unroll
            for (int u = 0; u < 2 * Q; ++u) st[u] = *reinterpret_cast<const double2 *>(panels + off + (size_t)(u * 256 + tid) * 2);
            off += 4096;
            if (off + 4096 > panel_doubles) off = 0;
        }
#pragma unroll
        for (int k4 = 0; k4 < 16; k4 += 4) {
            double a[4], b[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] = Hs[cur][k4 + lq][wc * 64 + t * 16 + l15];
#pragma unroll
            for (int t = 0; t < 4; ++t) b[t] = Gs[cur][k4 + lq][wr * 64 + t * 16 + l15];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            if (Q > 0 && k4 == 8) {
                // the staged pieces go to the other buffer in the shadow of the MFMAs (rows of 128 doubles, padded to 144)
#pragma unroll
                for (int u = 0; u < 2 * Q; ++u) {
                    const int e = u * 256 + tid, row = (e >> 6) & 15, col = (e & 63) * 2;
                    double *dst = (u & 1) ? &Hs[nxt][row][col] : &Gs[nxt][row][col];
                    *reinterpret_cast<double2 *>(dst) = st[u];
                }
            }
        }
        __syncthreads();
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
int run(const char *name, K kern, double *out, unsigned long long *clk, const double *panels, size_t pd, int q)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int nb = 512, chunks = 60000;  // ~0.5 s
    hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, 0, out, clk, panels, pd, 2000);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, 0, out, clk, panels, pd, chunks);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    static unsigned long long hc[1024];
    CK(hipMemcpy(hc, clk, sizeof(unsigned long long) * 2 * nb, hipMemcpyDeviceToHost));
    double ghz = 0;
    for (int b = 0; b < nb; ++b) ghz += (double)hc[2 * b] / (double)hc[2 * b + 1] * 0.1;
    ghz /= nb;
    const double tf = (double)nb * 4 * chunks * 64 * 2048.0 / (ms * 1e-3) / 1e12;
    const double gbs = (double)nb * chunks * q * 8192.0 / (ms * 1e-3) / 1e9;
    printf("%-44s %.1f TFLOP/s  clock %.3f GHz  %.1f clk per MFMA per SIMD  staged %.0f GB/s\n", name, tf, ghz,
           ghz * 1e9 * (ms * 1e-3) / ((double)chunks * 64 * 2), gbs);
    return 0;
}

int main()
{
    double *out, *panels;
    unsigned long long *clk;
    const size_t pd = (size_t)8 << 20;  // 64 MB of panels
    CK(hipMalloc(&out, sizeof(double) * 512 * 256));
    CK(hipMalloc(&panels, sizeof(double) * pd));
    CK(hipMalloc(&clk, sizeof(unsigned long long) * 2048));
    double *h = (double *)malloc(sizeof(double) * pd);
    unsigned long long st = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < pd; ++i) {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        h[i] = ((double)(st >> 11) / 9007199254740992.0 - 0.5) * 1e-3;  // random mantissas (the pipe draws more power on them), small values
    }
    CK(hipMemcpy(panels, h, sizeof(double) * pd, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; ++rep) {
        run("no staging (Q = 0)", k_feed<0>, out, clk, panels, pd, 0);
        run("128 x 128 tile's panel traffic (Q = 4)", k_feed<4>, out, clk, panels, pd, 4);
        run("256 x 128 tile's panel traffic (Q = 3)", k_feed<3>, out, clk, panels, pd, 3);
        run("half the panel traffic (Q = 2)", k_feed<2>, out, clk, panels, pd, 2);
    }
    return 0;
}
