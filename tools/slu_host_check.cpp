// Host emulation of the sparse direct route's numeric phase (csrc/sparse_direct.hip) on the structures of
// csrc/slu_analyse.h: equilibration, scatter, extend-add, partial LU of the fronts with pivoting restricted
// to the fully summed rows, forward / backward substitution.  Debugging aid for the analysis (no GPU needed):
//   g++ -O2 -std=c++17 -o tools/slu_host_check tools/slu_host_check.cpp
//   tools/slu_host_check matrix.bin x.bin      (driver: tools/slu_host_check.py)
// matrix.bin: int64 n, int64 nnz, int32 indptr[n+1], int32 indices[nnz], double data[nnz], double b[n]
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../nodal_amd/csrc/slu_analyse.h"

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int64_t n = 0, nnz = 0;
    if (fread(&n, 8, 1, f) != 1 || fread(&nnz, 8, 1, f) != 1) return 2;
    std::vector<int32_t> indptr((size_t)n + 1), indices((size_t)nnz);
    std::vector<double> data((size_t)nnz), b((size_t)n);
    if (fread(indptr.data(), 4, (size_t)n + 1, f) != (size_t)n + 1 || fread(indices.data(), 4, (size_t)nnz, f) != (size_t)nnz ||
        fread(data.data(), 8, (size_t)nnz, f) != (size_t)nnz || fread(b.data(), 8, (size_t)n, f) != (size_t)n)
        return 2;
    fclose(f);
    slu::Symbolic S;
    if (!slu::analyse(n, indptr.data(), indices.data(), data.data(), S, true)) {
        fprintf(stderr, "structurally singular\n");
        return 3;
    }
    const int32_t nsn = (int32_t)S.sn_start.size() - 1;
    // equilibration
    std::vector<double> rs((size_t)n, 1.0), cs((size_t)n, 0.0);
    for (int64_t i = 0; i < n; ++i) {
        double m = 0;
        for (int32_t e = indptr[i]; e < indptr[i + 1]; ++e) m = std::max(m, std::fabs(data[e]));
        rs[i] = m > 0 ? 1.0 / m : 1.0;
    }
    for (int64_t i = 0; i < n; ++i)
        for (int32_t e = indptr[i]; e < indptr[i + 1]; ++e) cs[indices[e]] = std::max(cs[indices[e]], std::fabs(data[e]) * rs[i]);
    for (int64_t j = 0; j < n; ++j) cs[j] = cs[j] > 0 ? 1.0 / cs[j] : 1.0;
    std::vector<double> fronts((size_t)S.front_off[nsn], 0.0);
    for (int64_t i = 0; i < n; ++i)
        for (int32_t e = indptr[i]; e < indptr[i + 1]; ++e) fronts[(size_t)S.dest[e]] += data[e] * rs[i] * cs[indices[e]];
    std::vector<int32_t> lperm((size_t)n);
    const double tiny = 1.4901161193847656e-08;
    int64_t perturbed = 0, swaps = 0;
    auto dims = [&](int32_t t, int &s, int &dim) {
        s = S.sn_start[t + 1] - S.sn_start[t];
        dim = s + (int)(S.struct_ptr[t + 1] - S.struct_ptr[t]);
    };
    const int32_t nlev = (int32_t)S.lvl_ptr.size() - 1;
    for (int32_t l = 0; l < nlev; ++l)
        for (int32_t q = S.lvl_ptr[l]; q < S.lvl_ptr[l + 1]; ++q) {
            const int32_t t = S.lvl_sn[q];
            int s, dim;
            dims(t, s, dim);
            double *F = fronts.data() + S.front_off[t];
            for (int32_t cq = S.child_ptr[t]; cq < S.child_ptr[t + 1]; ++cq) {
                const int32_t c = S.child_idx[cq];
                int cs_, cdim;
                dims(c, cs_, cdim);
                const int cb = cdim - cs_;
                const double *C = fronts.data() + S.front_off[c];
                const int32_t *map = S.cmap.data() + S.struct_ptr[c];
                for (int j = 0; j < cb; ++j)
                    for (int i = 0; i < cb; ++i) F[map[i] + (int64_t)map[j] * dim] += C[(cs_ + i) + (int64_t)(cs_ + j) * cdim];
            }
            int32_t *perm = lperm.data() + S.sn_start[t];
            for (int i = 0; i < s; ++i) perm[i] = i;
            for (int k = 0; k < s; ++k) {
                double best = -1;
                int br = k;
                for (int i = k; i < s; ++i)
                    if (std::fabs(F[i + (int64_t)k * dim]) > best) { best = std::fabs(F[i + (int64_t)k * dim]); br = i; }
                const double d = F[k + (int64_t)k * dim];
                if (!(best >= tiny)) {
                    if (getenv("SLU_DEBUG")) {
                        fprintf(stderr, "  perturbed: supernode %d level %d k %d of s %d dim %d best %.3e; pivot-block rows (orig row / orig col):", t, l, k, s, dim, best);
                        for (int i = 0; i < s; ++i) fprintf(stderr, " %d/%d", S.rowof[S.sn_start[t] + perm[i]], S.colof[S.sn_start[t] + i]);
                        fprintf(stderr, "\n    column k entries over all rows:");
                        for (int i = 0; i < dim; ++i) if (F[i + (int64_t)k * dim] != 0.0) fprintf(stderr, " [%d]=%.3g", i, F[i + (int64_t)k * dim]);
                        fprintf(stderr, "\n    row k entries:");
                        for (int j = 0; j < dim; ++j) if (F[k + (int64_t)j * dim] != 0.0) fprintf(stderr, " [%d]=%.3g", j, F[k + (int64_t)j * dim]);
                        fprintf(stderr, "\n");
                    }
                    F[k + (int64_t)k * dim] = d < 0 ? -tiny : tiny; br = k; ++perturbed;
                }
                else if (std::fabs(d) >= 0.25 * best) br = k;
                if (br != k) {
                    ++swaps;
                    std::swap(perm[k], perm[br]);
                    for (int j = 0; j < dim; ++j) std::swap(F[k + (int64_t)j * dim], F[br + (int64_t)j * dim]);
                }
                const double rp = 1.0 / F[k + (int64_t)k * dim];
                for (int i = k + 1; i < dim; ++i) F[i + (int64_t)k * dim] *= rp;
                for (int j = k + 1; j < dim; ++j) {
                    const double u = F[k + (int64_t)j * dim];
                    if (u == 0.0) continue;
                    for (int i = k + 1; i < dim; ++i) F[i + (int64_t)j * dim] -= F[i + (int64_t)k * dim] * u;
                }
            }
        }
    fprintf(stderr, "[host check] factorised: %lld perturbed pivots, %lld row interchanges\n", (long long)perturbed, (long long)swaps);
    // solve
    std::vector<double> xb((size_t)n), vec((size_t)S.vec_off[nsn]);
    for (int64_t k = 0; k < n; ++k) xb[k] = b[S.rowof[k]] * rs[S.rowof[k]];
    for (int32_t l = 0; l < nlev; ++l)
        for (int32_t q = S.lvl_ptr[l]; q < S.lvl_ptr[l + 1]; ++q) {
            const int32_t t = S.lvl_sn[q];
            int s, dim;
            dims(t, s, dim);
            const double *F = fronts.data() + S.front_off[t];
            double *v = vec.data() + S.vec_off[t];
            for (int i = 0; i < dim; ++i) v[i] = i < s ? xb[S.sn_start[t] + i] : 0.0;
            for (int32_t cq = S.child_ptr[t]; cq < S.child_ptr[t + 1]; ++cq) {
                const int32_t c = S.child_idx[cq];
                int cs_, cdim;
                dims(c, cs_, cdim);
                const double *cv = vec.data() + S.vec_off[c] + cs_;
                const int32_t *map = S.cmap.data() + S.struct_ptr[c];
                for (int i = 0; i < cdim - cs_; ++i) v[map[i]] += cv[i];
            }
            const int32_t *perm = lperm.data() + S.sn_start[t];
            double *tmp = v + dim;
            for (int i = 0; i < s; ++i) tmp[i] = v[perm[i]];
            for (int i = 0; i < s; ++i) v[i] = tmp[i];
            for (int k = 0; k < s; ++k)
                for (int i = k + 1; i < dim; ++i) v[i] -= F[i + (int64_t)k * dim] * v[k];
        }
    for (int32_t l = nlev - 1; l >= 0; --l)
        for (int32_t q = S.lvl_ptr[l]; q < S.lvl_ptr[l + 1]; ++q) {
            const int32_t t = S.lvl_sn[q];
            int s, dim;
            dims(t, s, dim);
            const double *F = fronts.data() + S.front_off[t];
            double *v = vec.data() + S.vec_off[t];
            const int32_t *bidx = S.struct_idx.data() + S.struct_ptr[t];
            for (int i = s; i < dim; ++i) v[i] = xb[bidx[i - s]];
            for (int i = 0; i < s; ++i)
                for (int j = s; j < dim; ++j) v[i] -= F[i + (int64_t)j * dim] * v[j];
            for (int k = s - 1; k >= 0; --k) {
                v[k] /= F[k + (int64_t)k * dim];
                for (int i = 0; i < k; ++i) v[i] -= F[i + (int64_t)k * dim] * v[k];
            }
            for (int i = 0; i < s; ++i) xb[S.sn_start[t] + i] = v[i];
        }
    std::vector<double> x((size_t)n);
    for (int64_t k = 0; k < n; ++k) x[S.colof[k]] = xb[k] * cs[S.colof[k]];
    f = fopen(argv[2], "wb");
    fwrite(x.data(), 8, (size_t)n, f);
    fclose(f);
    return 0;
}
