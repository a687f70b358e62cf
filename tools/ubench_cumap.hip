// CU-mask bit -> physical CU mapping on MI355X (hipExtStreamCreateWithCUMask): for single-bit masks, the set of
// (XCC, SE, CU) the workgroups of a 256-block launch land on; then the in-situ latency of small high-priority kernels
// beside a realistic stand-in for the trailing update (256 threads, 256 VGPRs -> 2 workgroups per CU, ~80 us each,
// thousands queued) under several reservations.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <vector>
#include <set>
#include <map>
#include <algorithm>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256, 2) void k_hog(double *out, int iters)
{
    double4_t acc[6][5];   // 120 VGPR pairs... keeps the kernel above 170 VGPRs: two workgroups per CU
    double a[6], b[5];
    for (int i = 0; i < 6; ++i) a[i] = 1.0 + 1e-9 * (threadIdx.x + 64 * i);
    for (int j = 0; j < 5; ++j) b[j] = 1.0 - 1e-9 * (threadIdx.x + 64 * j);
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 5; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    double s = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 5; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_small(unsigned *where)
{
    if (threadIdx.x == 0) {
        unsigned id, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        where[blockIdx.x] = ((id & 15) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
    }
    __builtin_amdgcn_s_sleep(100);
}

static std::vector<unsigned> where_set(hipStream_t s, unsigned *wh, int nblk)
{
    hipLaunchKernelGGL(k_small, dim3(nblk), dim3(256), 0, s, wh);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(nblk);
    hipMemcpy(h.data(), wh, 4 * nblk, hipMemcpyDeviceToHost);
    std::set<unsigned> st(h.begin(), h.end());
    return std::vector<unsigned>(st.begin(), st.end());
}

int main()
{
    double *o1;
    unsigned *wh;
    CK(hipMalloc(&o1, 8ull * 32768 * 256));
    CK(hipMalloc(&wh, 4 * 65536));
    {
        auto all = where_set(0, wh, 8192);
        std::map<unsigned, int> perx;
        for (unsigned w : all) perx[w >> 8]++;
        printf("no mask: %zu distinct (xcc,se,sh,cu); per xcc:", all.size());
        for (auto &kv : perx) printf(" %u:%d", kv.first, kv.second);
        printf("\n");
    }
    for (int b : {0, 1, 2, 7, 8, 31, 32, 33, 64, 100, 255}) {
        std::vector<uint32_t> m(8, 0);
        m[b / 32] = 1u << (b % 32);
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, 8, m.data()) != hipSuccess) { printf("bit %d: create failed\n", b); (void)hipGetLastError(); continue; }
        auto st = where_set(s, wh, 512);
        printf("bit %3d -> %zu CUs:", b, st.size());
        for (unsigned w : st) printf(" x%u.se%u.sh%u.cu%u", w >> 8, (w >> 5) & 7, (w >> 4) & 1, w & 15);
        printf("\n");
        CK(hipStreamDestroy(s));
    }
    {
        std::vector<uint32_t> m(8, 0);
        m[0] = 0xFF;
        hipStream_t s;
        CK(hipExtStreamCreateWithCUMask(&s, 8, m.data()));
        auto st = where_set(s, wh, 2048);
        printf("bits 0..7 -> %zu CUs\n", st.size());
        CK(hipStreamDestroy(s));
        m[0] = 0xFFFFFFFF;
        CK(hipExtStreamCreateWithCUMask(&s, 8, m.data()));
        st = where_set(s, wh, 4096);
        printf("bits 0..31 -> %zu CUs\n", st.size());
        CK(hipStreamDestroy(s));
        for (int w = 0; w < 8; ++w) m[w] = 0xFFFFFFFF;
        m[0] = 0xFFFFFFFE;
        CK(hipExtStreamCreateWithCUMask(&s, 8, m.data()));
        st = where_set(s, wh, 8192);
        printf("all but bit 0 -> %zu CUs\n", st.size());
        CK(hipStreamDestroy(s));
        m[0] = 0xFFFF0000;
        CK(hipExtStreamCreateWithCUMask(&s, 8, m.data()));
        st = where_set(s, wh, 8192);
        printf("all but bits 0..15 -> %zu CUs\n", st.size());
        CK(hipStreamDestroy(s));
    }
    // ---- in-situ latency with a realistic hog ----
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t sp;
    CK(hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, greatest));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int hog_iters = 21;   // 30 MFMAs x 64 clk x 21 x 2 co-resident waves ~ 80 us
    {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_hog, dim3(512), dim3(256), 0, 0, o1, hog_iters);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("hog: one round of 512 workgroups %.1f us (priority range least %d greatest %d)\n", ms * 1e3, least, greatest);
    }
    for (int reserve : {0, 1, 2, 4, 8, 16, 32}) {
        std::vector<uint32_t> m(8, 0xFFFFFFFFu);
        for (int r = 0; r < reserve; ++r) m[r / 32] &= ~(1u << (r % 32));
        hipStream_t sh;
        if (reserve == 0) CK(hipStreamCreateWithFlags(&sh, hipStreamNonBlocking));
        else CK(hipExtStreamCreateWithCUMask(&sh, 8, m.data()));
        for (int nsmall : {1, 8, 18, 64, 456}) {
            hipLaunchKernelGGL(k_hog, dim3(512 * 40), dim3(256), 0, sh, o1, hog_iters);   // ~3.3 ms
            std::vector<float> lat;
            for (int rep = 0; rep < 20; ++rep) {
                CK(hipEventRecord(e0, sp));
                hipLaunchKernelGGL(k_small, dim3(nsmall), dim3(256), 0, sp, wh);
                CK(hipEventRecord(e1, sp));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                lat.push_back(ms * 1e3f);
            }
            bool hog_running = hipStreamQuery(sh) == hipErrorNotReady;
            CK(hipDeviceSynchronize());
            std::sort(lat.begin(), lat.end());
            printf("reserve %2d bits: small kernel %3d WGs: latency us min %.1f median %.1f max %.1f (hog running at end: %d)\n", reserve, nsmall,
                   lat.front(), lat[lat.size() / 2], lat.back(), (int)hog_running);
        }
        CK(hipStreamDestroy(sh));
    }
    return 0;
}
