// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access mix of the sweep kernel's tile items
// (MI355X_MICROARCH.md: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) --
// other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Three copy kernels over buffers far larger than the Infinity Cache, each moving a KNOWN number of bytes:
//   k_copy8    8 B per lane loads and stores, 128 contiguous bytes per 16 lanes (the C-tile loads/stores of a tile item)
//   k_copy16   16 B per lane loads (the panel staging loads: double2), 8 B stores
//   k_mix      the tile item's mix: per 128 KB of 8-B C-tile loads, 786 KB of 16-B panel loads; 128 KB of 8-B stores
// Run under  rocprofv3 --pmc FETCH_SIZE  and  --pmc WRITE_SIZE  (separate passes); the program prints the bytes each kernel
// moved so that the counter / bytes factors can be read off (tools/prof_summary.py calib).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_copy8(const double *__restrict__ src, double *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i] + 1.0;
}
__global__ __launch_bounds__(256) void k_copy16(const double2 *__restrict__ src, double *__restrict__ dst, size_t n2)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        const double2 v = src[i];
        dst[i] = v.x + v.y;
    }
}
// per "tile": 16384 doubles of C read 8 B/lane and written back, 6 x 16384 doubles of panel data read as double2
__global__ __launch_bounds__(256) void k_mix(const double *__restrict__ c_in, double *__restrict__ c_out, const double2 *__restrict__ pan,
                                              size_t ntile)
{
    for (size_t t = blockIdx.x; t < ntile; t += gridDim.x) {
        const double *ci = c_in + t * 16384;
        double *co = c_out + t * 16384;
        const double2 *pp = pan + t * (6 * 8192);
        double acc[64];
#pragma unroll
        for (int u = 0; u < 64; ++u) acc[u] = ci[threadIdx.x + 256 * u];
#pragma unroll
        for (int u = 0; u < 6 * 32; ++u) {  // fully unrolled: acc[] must stay in registers (a rolled loop indexes it dynamically -> scratch)
            const double2 v = pp[threadIdx.x + 256 * u];
            acc[u & 63] += v.x * 1e-9 + v.y * 1e-9;
        }
#pragma unroll
        for (int u = 0; u < 64; ++u) co[threadIdx.x + 256 * u] = acc[u];
    }
}

int main()
{
    const size_t n = (size_t)1 << 28;  // 2 GiB of doubles per buffer
    double *a, *b, *c;
    CK(hipMalloc(&a, n * 8));
    CK(hipMalloc(&b, n * 8));
    const size_t ntile = 4096, pan_bytes = ntile * 6 * 16384 * 8;  // 3 GiB of panel data for k_mix
    CK(hipMalloc(&c, pan_bytes));
    CK(hipMemset(a, 0, n * 8));
    CK(hipMemset(b, 0, n * 8));
    CK(hipMemset(c, 0, pan_bytes));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_copy8, dim3(4096), dim3(256), 0, 0, a, b, n);
        hipLaunchKernelGGL(k_copy16, dim3(4096), dim3(256), 0, 0, (const double2 *)a, b, n / 2);
        hipLaunchKernelGGL(k_mix, dim3(2048), dim3(256), 0, 0, a, b, (const double2 *)c, ntile);
        CK(hipDeviceSynchronize());
    }
    printf("k_copy8:  read %zu B (8 B/lane), wrote %zu B\n", n * 8, n * 8);
    printf("k_copy16: read %zu B (16 B/lane), wrote %zu B\n", n * 8, n * 4);
    printf("k_mix:    read %zu B (8 B/lane) + %zu B (16 B/lane), wrote %zu B\n", ntile * 16384 * 8, pan_bytes, ntile * 16384 * 8);
    return 0;
}
