// Stand-alone check of k_hamming_fp4.hip (the fp4-MFMA lower bound of the all-pairs Hamming distances): random clustered bit planes,
// the candidate list of the kernel against the pairs with D < 3 thresh counted on the CPU, as sets.  Validates the operand and
// accumulator layouts the kernel assumes (a wrong one gives wrong candidates at once).
//   hipcc -O2 --offload-arch=gfx950 -std=c++17 -Iinclude -Igaussdca.jl_amd/csrc tools/test_fp4_gram.hip -o tools/_bin/test_fp4_gram
#include "../gaussdca.jl_amd/csrc/k_hamming_fp4.hip"

#include <set>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
gdca_recorder *&gdca_recorder::active()
{
    static thread_local gdca_recorder *r = nullptr;
    return r;
}
hipError_t gdca_recorder::flush() { return hipSuccess; }
void gdca_recorder::add(const gdca_op &) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int run(int N, int M, int thresh, unsigned long long seed)
{
    const int NW = (N + 31) / 32, Mt128 = (M + 127) / 128;
    std::vector<uint32_t> Zb((size_t)Mt128 * 5 * NW * 128, 0u);
    std::vector<uint8_t> Z((size_t)M * N);
    unsigned long long st = seed;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(st >> 33); };
    const int ncl = std::max(1, M / 20);
    std::vector<uint8_t> cen((size_t)ncl * N);
    for (auto &c : cen) c = 1 + rnd() % 21;
    for (int k = 0; k < M; ++k) {
        const int c = rnd() % ncl, mu = rnd() % 50;
        for (int i = 0; i < N; ++i) Z[(size_t)k * N + i] = (rnd() % 100 < (unsigned)mu) ? 1 + rnd() % 21 : cen[(size_t)c * N + i];
    }
    for (int k = 0; k < M; ++k)
        for (int i = 0; i < N; ++i)
            for (int p = 0; p < 5; ++p)
                if ((Z[(size_t)k * N + i] >> p) & 1) Zb[(((size_t)(k >> 7) * 5 + p) * NW + (i >> 5)) * 128 + (k & 127)] |= 1u << (i & 31);
    std::set<std::pair<int, int>> want;
    for (int k = 0; k < M; ++k)
        for (int l = k + 1; l < M; ++l) {
            int D = 0;
            for (int i = 0; i < N; ++i) D += __builtin_popcount((Z[(size_t)k * N + i] ^ Z[(size_t)l * N + i]) & 7);
            if (D < 3 * thresh) want.insert({k, l});
        }
    uint32_t *dZb;
    void *img;
    gdca_dev_scalars *sc, hsc;
    int2 *list;
    const unsigned cap = 1u << 20;
    CK(hipMalloc(&dZb, Zb.size() * 4));
    CK(hipMemcpy(dZb, Zb.data(), Zb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&img, gdca_fp4_image_bytes(N, M)));
    CK(hipMemset(img, 0xFF, gdca_fp4_image_bytes(N, M)));
    CK(hipMalloc(&sc, sizeof(gdca_dev_scalars)));
    memset(&hsc, 0, sizeof hsc);
    hsc.thresh = thresh;
    hsc.ham_mode = 2;
    CK(hipMemcpy(sc, &hsc, sizeof hsc, hipMemcpyHostToDevice));
    CK(hipMalloc(&list, (size_t)cap * 8));
    gdca_launch_hamming_fp4(0, dZb, img, N, M, sc, list, cap);
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    CK(hipMemcpy(&hsc, sc, sizeof hsc, hipMemcpyDeviceToHost));
    std::vector<int2> got((size_t)std::min<unsigned long long>(hsc.ham_ncand, cap));
    CK(hipMemcpy(got.data(), list, got.size() * 8, hipMemcpyDeviceToHost));
    std::set<std::pair<int, int>> have;
    size_t dup = 0, unordered = 0;
    for (auto &p : got) {
        if (p.x >= p.y) ++unordered;
        if (!have.insert({std::min(p.x, p.y), std::max(p.x, p.y)}).second) ++dup;
    }
    size_t missing = 0, extra = 0;
    for (auto &p : want) missing += !have.count(p);
    for (auto &p : have) extra += !want.count(p);
    printf("N %4d M %5d thresh %3d: CPU %zu pairs with D < 3 thresh, kernel listed %llu (%zu distinct, %zu duplicates, %zu with k >= l): missing %zu, extra %zu  %s\n", N, M, thresh,
           want.size(), (unsigned long long)hsc.ham_ncand, have.size(), dup, unordered, missing, extra, (missing || extra || dup || unordered) ? "FAILED" : "ok");
    (void)hipFree(dZb); (void)hipFree(img); (void)hipFree(sc); (void)hipFree(list);
    return (missing || extra || dup || unordered) ? 1 : 0;
}

int main()
{
    int bad = 0;
    bad += run(200, 700, 60, 1);
    bad += run(64, 300, 20, 2);
    bad += run(33, 1000, 12, 3);
    bad += run(500, 1500, 150, 4);
    bad += run(96, 256, 30, 5);
    bad += run(1, 10, 1, 6);
    printf(bad ? "FAILED\n" : "all ok\n");
    return bad;
}
