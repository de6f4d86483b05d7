// General sparse LU for large non-symmetric MNA systems (branch equations with
// zero diagonals).  Not built yet: fails loudly instead of falling back.
#include "ctx.h"

int sparse_lu_solve(nodal_ctx *h, int32_t *info) {
    (void)info;
    return nodal_fail(h, NODAL_E_UNSUPPORTED,
                      "sparse LU for large non-symmetric systems is not implemented yet "
                      "(n above the densify limit with branch equations)");
}
