// Host timing of the sparse direct route's analysis (csrc/slu_analyse.h) on a matrix file written by
// tools/slu_analyse_time.py:  g++ -O2 -std=c++17 -pthread tools/slu_analyse_time.cpp -o tools/slu_analyse_time
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <queue>
#include <cmath>
#include <numeric>
#include "../nodal_amd/csrc/slu_analyse.h"
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb");
    int64_t hdr[2];
    if (!f || fread(hdr, 8, 2, f) != 2) return 1;
    const int64_t n = hdr[0], nnz = hdr[1];
    std::vector<int32_t> ip((size_t)n + 1), idx((size_t)nnz);
    std::vector<double> val((size_t)nnz);
    if (fread(ip.data(), 4, (size_t)n + 1, f) != (size_t)n + 1 || fread(idx.data(), 4, (size_t)nnz, f) != (size_t)nnz ||
        fread(val.data(), 8, (size_t)nnz, f) != (size_t)nnz) return 1;
    fclose(f);
    for (int rep = 0; rep < 2; ++rep) {
        slu::Symbolic S;
        const auto t0 = std::chrono::steady_clock::now();
        const bool ok = slu::analyse(n, ip.data(), idx.data(), val.data(), S, true);
        fprintf(stderr, "analyse: %s, %.1f ms\n", ok ? "ok" : "structurally singular",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    return 0;
}
