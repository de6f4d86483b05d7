// nd_compare <matrix.bin> ...: the nested dissection of csrc/slu_analyse.h against the sequential form it replaced
// (tools/nd_reference.h) on the symmetrised graph of each matrix (file format of tools/slu_analyse_time.py:
// n, nnz, indptr, indices, data).  Also random graphs with several components and hubs when called with "random".
//   g++ -O2 -std=c++17 -pthread tools/nd_compare.cpp -o /tmp/nd_compare
#include "nd_reference.h"
#include <random>
#include <cstring>

static int compare(const char *name, int64_t n, const slu::Graph &g) {
    std::vector<int32_t> o1, s1, o2, s2;
    slu::nested_dissection(n, g, o1, s1);
    slu::nested_dissection_ref(n, g, o2, s2);
    const bool same = o1 == o2 && s1 == s2;
    printf("%s: n %lld, %zu supernodes: %s\n", name, (long long)n, s1.size() - 1, same ? "identical" : "DIFFERENT");
    return same ? 0 : 1;
}

int main(int argc, char **argv) {
    int bad = 0;
    for (int a = 1; a < argc; ++a) {
        if (!strcmp(argv[a], "random")) {
            std::mt19937 rng(7);
            for (int t = 0; t < 12; ++t) {
                // a few grids of different sizes side by side (components), random chords, one or two hubs
                const int parts = 1 + t % 4;
                std::vector<std::pair<int32_t, int32_t>> edges;
                int64_t n = 0;
                for (int p = 0; p < parts; ++p) {
                    const int w = 5 + (int)(rng() % 60), hgt = 3 + (int)(rng() % 50);
                    for (int y = 0; y < hgt; ++y)
                        for (int x = 0; x < w; ++x) {
                            const int32_t v = (int32_t)(n + y * w + x);
                            if (x + 1 < w) edges.push_back({v, v + 1});
                            if (y + 1 < hgt) edges.push_back({v, v + w});
                        }
                    const int64_t m = (int64_t)w * hgt;
                    for (int c = 0; c < (int)(m / 7); ++c)
                        edges.push_back({(int32_t)(n + rng() % m), (int32_t)(n + rng() % m)});
                    if (t % 3 == 0 && m > 200)
                        for (int c = 0; c < 150; ++c) edges.push_back({(int32_t)n, (int32_t)(n + 1 + rng() % (m - 1))});
                    n += m;
                }
                n += 3;  // isolated vertices
                std::vector<std::vector<int32_t>> adj((size_t)n);
                for (auto &e : edges)
                    if (e.first != e.second) {
                        adj[(size_t)e.first].push_back(e.second);
                        adj[(size_t)e.second].push_back(e.first);
                    }
                slu::Graph g;
                g.ptr.assign(1, 0);
                for (auto &l : adj) {
                    std::sort(l.begin(), l.end());
                    l.erase(std::unique(l.begin(), l.end()), l.end());
                    g.adj.insert(g.adj.end(), l.begin(), l.end());
                    g.ptr.push_back((int64_t)g.adj.size());
                }
                char nm[64];
                snprintf(nm, sizeof nm, "random %d (%d parts)", t, parts);
                bad += compare(nm, n, g);
            }
            continue;
        }
        FILE *f = fopen(argv[a], "rb");
        int64_t hdr[2];
        if (!f || fread(hdr, 8, 2, f) != 2) return 2;
        const int64_t n = hdr[0], nnz = hdr[1];
        std::vector<int32_t> ip((size_t)n + 1), idx((size_t)nnz);
        std::vector<double> val((size_t)nnz);
        if (fread(ip.data(), 4, (size_t)n + 1, f) != (size_t)n + 1 || fread(idx.data(), 4, (size_t)nnz, f) != (size_t)nnz ||
            fread(val.data(), 8, (size_t)nnz, f) != (size_t)nnz) return 2;
        fclose(f);
        std::vector<int32_t> rmatch;
        if (!slu::row_matching(n, ip.data(), idx.data(), val.data(), rmatch)) { printf("%s: structurally singular\n", argv[a]); continue; }
        slu::Graph g;
        slu::symmetrised_graph(n, ip.data(), idx.data(), rmatch, g);
        bad += compare(argv[a], n, g);
    }
    return bad ? 1 : 0;
}
