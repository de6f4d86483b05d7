/*
 * nodal_hip.h -- C ABI of libnodal_hip.so: MI355X (gfx950) assembly of the
 * modified-nodal-analysis system G x = A and its dense / sparse solve.
 *
 * The reference (EnricoMiccoli/nodal v1.3.0) has no FFI layer; its seam is the
 * Python pair Circuit.build_model / Circuit.solve.  Each entry point below
 * names the reference code it replaces (file:line into the reference tree).
 * The host-side mirror of the reference API that calls these through ctypes is
 * nodal_amd/circuit.py; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every function returns a nodal_status (0 = OK); no exception crosses;
 *   - the caller owns all host buffers; the opaque handle owns device memory;
 *   - one handle per (device, stream); thread-compatible, not thread-safe;
 *   - node indices are int32, -1 means "lead is the ground node";
 *   - unknown vector layout: x[0:K] node potentials in nodenum order,
 *     x[K:K+B] branch currents in anomnum order (reference nodal/nodal.py:405-408).
 */
#ifndef NODAL_HIP_H
#define NODAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nodal_ctx *nodal_handle;

typedef enum {
    NODAL_OK = 0,
    NODAL_E_INVALID = 1,          /* bad argument / call order                      */
    NODAL_E_HIP = 2,              /* a HIP runtime call failed (see nodal_last_error) */
    NODAL_E_ZERO_RESISTANCE = 3,  /* -> ValueError   (reference nodal/models.py:14-17) */
    NODAL_E_STAMP_COLLISION = 4,  /* -> AssertionError (reference nodal/models.py:43,47,
                                      66,70,168,172,176,193,197: `assert G[i, j] == 0`) */
    NODAL_E_SINGULAR = 5,         /* zero pivot / floating sub-network / non-finite x */
    NODAL_E_NOMEM = 6,
    NODAL_E_UNSUPPORTED = 7
} nodal_status;

/* component type codes of the `type` column (nodal_amd/constants.py TYPE_CODE).
 * VCCS rows carry NODAL_T_VCVS: the reference dispatches them to write_VCVS
 * (reference nodal/nodal.py:377-378). */
enum { NODAL_T_R = 0, NODAL_T_A = 1, NODAL_T_E = 2, NODAL_T_VCVS = 3,
       NODAL_T_CCVS = 4, NODAL_T_CCCS = 5,
       /* internal (never produced by the netlist front end): transconductance stamp
        * without a branch unknown -- a current value*(e_c - e_d) flows from lead a to
        * lead b.  Used by the presolve that eliminates branch equations (presolve.hip). */
       NODAL_T_GM = 6 };

/* sparse solver selection for nodal_solve_sparse */
enum { NODAL_SPARSE_AUTO = 0,
       NODAL_SPARSE_PCG = 1,      /* SPD: multigrid-preconditioned flexible CG (Jacobi-CG when small) */
       NODAL_SPARSE_DENSIFY = 2,  /* scatter to a dense panel, LU with pivoting                      */
       NODAL_SPARSE_LU = 3,       /* general: block-preconditioned flexible GMRES (historic name)   */
       NODAL_SPARSE_DIRECT = 4 }; /* multifrontal LU (static matching, nested dissection, pivoting inside the
                                     fronts) + fp64 refinement: what AUTO falls back on when an iteration
                                     gives up -- spsolve's "any non-singular G" (reference nodal/nodal.py:325) */

/* ---- lifetime ---------------------------------------------------------- */
int nodal_create(int device_id, nodal_handle *out);
int nodal_destroy(nodal_handle h);
const char *nodal_last_error(nodal_handle h);
/* library / build identification, e.g. "nodal_hip 0.1 gfx950" */
const char *nodal_version(void);

/* ---- component table (replaces the per-component Python objects read by
 *      Circuit.build_model, reference nodal/nodal.py:338-368) --------------
 * Copies the structure-of-arrays table to HBM.  K = nums["kcl"], B = nums["be"].
 * The range check of the rows (nodes < K, drivers < ncomp, branch types <=> k >= 0)
 * runs on the device behind the copies; NODAL_E_INVALID as before for a bad row.
 * c, d, drv, k may be NULL -- all four together -- when B == 0 (resistors and current
 * sources read none of them): the columns then hold -1 on the device.  In a table WITH those columns they are
 * read for the rows whose type uses them (E .. CCCS and the internal GM) and taken as -1 on resistor and current-source
 * rows, whatever the caller's arrays hold there: only the dependent rows' entries travel (round 5).
 * Columns that live in pinned memory (nodal_host_alloc) are copied by DMA at link
 * rate; pageable memory goes through the runtime's staging copies. */
int nodal_upload_components(nodal_handle h, int64_t ncomp,
                            const uint8_t *type, const double *value,
                            const int32_t *a, const int32_t *b,
                            const int32_t *c, const int32_t *d,
                            const int32_t *drv, const int32_t *k,
                            int32_t K, int32_t B);
/* Page-locked host memory for the table's columns (what the lowering of a large netlist
 * writes into: reference nodal/nodal.py:338-368 builds Python objects there).  Needs a
 * HIP device; NODAL_E_HIP otherwise (callers fall back to ordinary memory). */
int nodal_host_alloc(size_t bytes, void **out);
int nodal_host_free(void *p);

/* Replace the value column only (same topology): `batch` members, row-major
 * [batch][ncomp].  Used for value sweeps (BASELINE.json config 4). */
int nodal_upload_values(nodal_handle h, int32_t batch, const double *values);

/* ---- assembly (replaces Circuit.build_model + models.write_*, reference
 *      nodal/nodal.py:338-398, nodal/models.py:13-214) ---------------------
 * symbolic: sparsity pattern (CSR, sorted columns) + ordered contribution
 *           lists; depends on topology only, reusable across a value sweep.
 * numeric : folds every matrix / rhs entry's contributions in component order
 *           (bit-identical to the reference's sequential += / = stamping) for
 *           batch member `member` (0 when no batch was uploaded).
 * On NODAL_E_ZERO_RESISTANCE / NODAL_E_STAMP_COLLISION, *bad_component (may be
 * NULL) receives the table row of the first offending component. */
int nodal_assemble_symbolic(nodal_handle h);
int nodal_assemble_numeric(nodal_handle h, int32_t member, int64_t *bad_component);

/* sizes after symbolic assembly */
int nodal_get_sizes(nodal_handle h, int64_t *n, int64_t *nnz, int64_t *ncontrib);

/* ---- export for parity / debugging (what the reference exposes as
 *      Circuit.G, Circuit.A; reference nodal/nodal.py:311,396-398) --------- */
int nodal_export_csr(nodal_handle h, int32_t *indptr, int32_t *indices,
                     double *data, double *rhs);
/* row-major n x n, as numpy's Circuit.G */
int nodal_export_dense(nodal_handle h, double *G, double *rhs);

/* ---- solve (replaces Circuit.solve, reference nodal/nodal.py:313-336) ----
 * dense : direct solve of the dense system, replacing LAPACK dgesv behind
 *         np.linalg.solve (reference nodal/nodal.py:327): partial pivoting with dgesv's
 *         pivot order for small systems, tournament pivoting for large general ones,
 *         pivot-free block elimination for (presolved) conductance networks -- see
 *         DESIGN.md section 3.2.  *info > 0: singular matrix (an exactly zero pivot, or
 *         a floating sub-network of a passive system) -> status NODAL_E_SINGULAR (host
 *         maps it to LinAlgError / UnconnectedCircuitError as reference
 *         nodal/nodal.py:328-335).
 * sparse: replaces scipy.sparse.linalg.spsolve (reference nodal/nodal.py:325).
 *         On a singular system x is filled with NaN, *info > 0 and the status is
 *         NODAL_OK: the reference's sparse path warns and returns NaNs, it does
 *         not raise.
 * x may be NULL to leave the solution on the device (nodal_download_x). */
int nodal_solve_dense(nodal_handle h, double *x, int32_t *info);
int nodal_solve_sparse(nodal_handle h, int32_t method, double *x, int32_t *info,
                       int32_t *iters, double *resid);
int nodal_download_x(nodal_handle h, double *x);

/* ---- equivalent-resistance sweep (replaces one equivalent_resistance() call per
 *      pair, reference nodal/equiv.py:31-61: deepcopy + rebuild + re-solve) ---------
 * For every pair (ia[q], ib[q]) of node indices (-1 = ground) a 1 A probe enters ia and
 * leaves ib; resistance[q] = e(ia) - e(ib).  G is factorised (dense) or its multigrid
 * hierarchy built (sparse) once for all pairs.  The circuit's own rhs is ignored, as
 * it is zero for the resistive networks the reference accepts here.  *info > 0:
 * singular network (dense: status NODAL_E_SINGULAR; sparse: NaNs, status OK). */
int nodal_solve_pairs(nodal_handle h, int32_t dense, int32_t npairs, const int32_t *ia,
                      const int32_t *ib, double *resistance, int32_t *info);

/* scaled residual ||G x - A||_inf / (||G||_inf ||x||_inf + ||A||_inf) of the
 * solution currently on the device, computed on the device from the CSR form */
int nodal_residual(nodal_handle h, double *scaled_residual);

/* ---- whole-path entry for resident inputs --------------------------------
 * symbolic + numeric (member) + solve, nothing copied to the host.
 * dense != 0 selects the dense path.  Used by bench.py's timed region. */
int nodal_run(nodal_handle h, int32_t dense, int32_t member, int32_t reuse_symbolic,
              int32_t *info);

/* ---- batch variant (BASELINE.json config 4) -----------
 * Replaces a Python loop of `Circuit(netlist, sparse=True)` + `.solve()` (reference
 * nodal/nodal.py:306-336) over the members [first, first + count) of the value table
 * uploaded by nodal_upload_values: the members are assembled and solved on the device as
 * ONE block-diagonal system built from the single topology in HBM (sparse path).
 * x_out (may be NULL: results stay on the device, see nodal_batch_x_device) receives
 * count x n doubles, row m = unknown vector of member first + m.  info_out (may be NULL)
 * receives one int per member: 0 solved; > 0 singular network (row of NaNs, as the
 * reference's spsolve); < 0 minus the nodal_status the member's assembly failed with
 * (NODAL_E_ZERO_RESISTANCE, NODAL_E_STAMP_COLLISION; row of NaNs).  When the block system
 * cannot be solved as a whole the members are solved one by one, so only the offending
 * members are marked.  reuse_symbolic != 0 keeps the block pattern of the previous call
 * with the same topology and count. */
int nodal_run_batch(nodal_handle h, int32_t first, int32_t count, int32_t reuse_symbolic,
                    double *x_out, int32_t *info_out);
/* copy the count x n results of the last nodal_run_batch into DEVICE memory of the handle's
 * GPU (e.g. the send buffer of a collective); capacity_bytes is the size of that buffer */
int nodal_batch_x_device(nodal_handle h, void *device_dst, int64_t capacity_bytes);
/* the same for the n unknowns of the last single-circuit solve (nodal_solve_dense / nodal_solve_sparse /
 * nodal_run): x of an independent circuit into the send buffer of the all_gather that shares the ranks'
 * solutions (nodal_amd/batch.py ShardedCircuits; the D2D counterpart of nodal_download_x, which replaces the
 * host array the reference's Circuit.solve returns, reference nodal/nodal.py:336).  Returns when the copy is done. */
int nodal_x_device(nodal_handle h, void *device_dst, int64_t capacity_bytes);

/* ---- timing of the last call, measured with HIP events on the handle's
 *      stream: milliseconds spent in [symbolic, numeric, factor/solve] ----- */
int nodal_last_timings(nodal_handle h, double *ms3);
/* HIP-event duration (ms) and launch count of the dominant kernel class of the
 * last solve (SpMV for the iterative path, trailing GEMM update for dense LU) */
int nodal_last_kernel_stats(nodal_handle h, double *ms_total, int64_t *launches,
                            double *alg_bytes_or_flops);

/* iterations of the last sparse solve (0 for direct paths), multigrid levels used
 * and the solver's own relative residual estimate */
int nodal_last_solve_info(nodal_handle h, int32_t *iterations, int32_t *amg_levels,
                          double *relative_residual);

int nodal_synchronize(nodal_handle h);

/* ---- options -------------------------------------------------------------
 * NODAL_OPT_FORCE_PIVOTING (0/1): dense LU always searches pivots, even on
 *   passive networks (column diagonally dominant G) where it provably never swaps.
 * NODAL_OPT_GEPP_PANEL (0/1, default 1): partial pivoting factors a 32-column panel in one launch
 *   (registers) instead of two launches per column; both forms give the same bits (cross-check).
 * NODAL_OPT_EXTRA_STREAMS (0/1, default 0; environment NODAL_EXTRA_STREAMS): a handle that is used ALONE -- one
 *   handle in the process, one call at a time -- may spread independent pieces of a solve over streams of its own
 *   (the multigrid setup builds R beside A P, the direct factorisation runs the wide fronts of a level side by
 *   side).  Same kernels, same results.  Off by default: the runtime maps a process's streams onto a handful of
 *   hardware queues, and a second stream per handle makes the main streams of several handles share them.
 * NODAL_OPT_BORROW_TABLE (0/1, default 0): the caller promises that the columns it passes to
 *   nodal_upload_components stay valid and unchanged until the next upload or nodal_destroy.  The library then reads
 *   them in place where its host code needs the table (the presolve of systems with branch equations looks at the
 *   branch rows and at the rows touching an eliminated node) instead of keeping a copy of its own: the upload of a
 *   table with branches is DMA only.  nodal_amd/_ffi.py sets it (its Handle keeps the arrays alive). */
enum { NODAL_OPT_FORCE_PIVOTING = 1, NODAL_OPT_GEPP_PANEL = 2, NODAL_OPT_EXTRA_STREAMS = 3, NODAL_OPT_BORROW_TABLE = 4 };
int nodal_set_option(nodal_handle h, int32_t option, int32_t value);

/* ---- testing hooks (not part of the reference-facing surface) -------------
 * C[M x N] -= A[M x K] * B[K x N] on the device with the LU's trailing-update
 * kernel; host buffers, column-major, leading dimensions M, K, M. */
int nodal_debug_gemm(nodal_handle h, int32_t M, int32_t N, int32_t K, const double *A,
                     const double *B, double *C);

#ifdef __cplusplus
}
#endif
#endif /* NODAL_HIP_H */
