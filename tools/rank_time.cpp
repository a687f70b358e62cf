// Wall time of gdca_ranking (compute_ranking, src/GaussDCA.jl:88-99) on a random symmetric N x N matrix: tools/_bin/rank_time N
// g++ -O2 -std=c++17 -I include tools/rank_time.cpp -o tools/_bin/rank_time -L gaussdca.jl_amd -lgdca -Wl,-rpath,$PWD/gaussdca.jl_amd
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "gdca.h"
int main(int argc, char **argv) {
    int N = argc > 1 ? atoi(argv[1]) : 500;
    std::vector<double> S((size_t)N * N);
    std::mt19937_64 g(1);
    std::normal_distribution<double> d;
    for (int i = 0; i < N; ++i) for (int j = 0; j <= i; ++j) { double x = d(g); S[(size_t)i + (size_t)j * N] = x; S[(size_t)j + (size_t)i * N] = x; }
    long long n = gdca_ranking_length(N, 5);
    std::vector<int> ii(n), jj(n); std::vector<double> sc(n);
    for (int rep = 0; rep < 6; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        gdca_ranking(S.data(), N, 5, ii.data(), jj.data(), sc.data());
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("N %d rep %d: %.2f ms\n", N, rep, ms);
    }
}
