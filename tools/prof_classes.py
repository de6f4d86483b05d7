"""Where the GPU time of a solve goes, by class of kernel (profiles/<round>_classes.json, read by
bench.py for `roofline.by_class` / `top_kernel_by_time`).

    rocprofv3 --kernel-trace -d DIR -- python3 bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu --no-also --concurrent 0
    python tools/prof_classes.py DIR cfg3 1000000 4995995 > profiles/r03_classes_cfg3.json

Dispatches are told apart by kernel name AND grid size: the level-0 launches of the cycle kernels
(k_restrict, k_prolong, ...) are the ones whose grid covers the 1e6 rows; the row kernels of level 0
run on a grid capped at 1024 workgroups and are recognised by their width class (<5, ...>).  Algorithmic bytes of the
level-0 passes: rows x bytes per row of what the kernel reads and writes.
"""
import glob, json, sqlite3, sys

W0 = 5  # padded ELL width of the grid's level 0
_F32ROW = W0 * 6          # column offset i16 (round 5: Ell::dcol) + val f32 per slot
_F64ROW = W0 * 10
# bytes per level-0 row of each pass (reads + writes; the outer iteration's vectors x, r, p, Ap are f64, the
# vectors inside the cycle -- x0, the cycle's residual, xp, z -- f32 since round 4: csrc/sagg.hip, cyc_t)
LEVEL0_BYTES = {
    "k_smooth_residual": _F32ROW + 8 + 4 + 4,            # A, b (f64: the outer residual), x0 gather, r
    "k_restrict": 4 * 8 + 4 + 8 / 7.0,                   # R entries (col + f32 val) of the 4 P slots per fine row, r, rc
    "k_prolong": 4 * 8 + 4 + 4 + 8 / 7.0,                # P (col + f32 val) x 4, x, xp, coarse gathers
    "k_post": _F32ROW + 8 + 4 + 4 + 4 + 8 + 8,           # A, b (f64), xp gather + own, out (z), u (Ap, f64), dinv
    "f_spmv": _F64ROW + 8 + 8 + 8,                       # A (f64), p gather + own, Ap
    "f_dir_spmv": _F64ROW + 8 + 4 + 8 + 8,               # A (f64), p_old and z gathered, the new direction, Ap
    "f_direction": 4 + 8 + 8,                            # z, p, p
    "f_update": 8 * 4 + 8 * 2 + 4 + 8,                   # x r p Ap in, x r out, x0 out, dinv
    "f_init": 8 * 2 + 8 * 3 + 4,
}
SETUP_NAMES = ("mis_", "assign_", "build_P", "r_count", "r_fill", "r_sort", "r_to_ell", "r_refresh", "ap_rows",
               "galerkin", "coarsest_inverse", "flags_up", "last_level", "any_unflagged", "k_tail_pack", "row_stats",
               "csr_to_ell", "reduce_bstat", "scan_", "grounded_flags", "select_nodes")
STAMP_NAMES = ("grp::", "fold_matrix", "fold_rhs", "set_tail")


def short(name):
    nm = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return nm.split("(")[0]


def class_of(nm, grid_threads, wg_threads, n):
    """(class, base kernel name) of one dispatch: kernel name AND grid size (see the module docstring).  `nm` is the
    short kernel name (no namespace, no argument list), the sizes are in threads."""
    base = nm.split("<")[0]
    wgs = (grid_threads / wg_threads) if (grid_threads and wg_threads) else 0
    big = n / 256 * 0.9  # a grid that covers the level-0 rows (one thread per row, 256 per workgroup)
    if any(base.startswith(p) or p in nm for p in STAMP_NAMES):
        return "stamping", base
    if any(base.startswith(p) for p in SETUP_NAMES):
        return "hierarchy_setup", base
    if base in LEVEL0_BYTES and (wgs >= big or nm.startswith(base + "<%d" % W0) or
                                 base in ("f_spmv", "f_dir_spmv", "f_direction", "f_update", "f_init")):
        return "level0_passes", base
    if base.startswith("k_") or base.startswith("f_"):
        return "coarse_levels", base
    return "other", base


def classify(path, workload, n, nnz, verbose=False):
    """Classes of kernels of one rocprofv3 --kernel-trace results.db (a directory is searched for it)."""
    dbs = glob.glob(path + "/**/*.db", recursive=True) if not path.endswith(".db") else [path]
    c = sqlite3.connect(dbs[0])
    tables = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    kt = "kernels" if "kernels" in tables else [t for t in tables if "kernel" in t.lower()][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({kt})")]
    if verbose:
        print(kt, cols, file=sys.stderr)
    gcol = next((x for x in ("grid_size_x", "grid_x", "grid_size") if x in cols), None)
    wcol = next((x for x in ("workgroup_size_x", "workgroup_x", "workgroup_size") if x in cols), None)
    q = f"select name, start, end, {gcol or 0}, {wcol or 1} from {kt} order by start"
    rows = list(c.execute(q))

    classes = {}
    per_kernel = {}
    total = 0

    def add(cls, dur, nbytes=0.0):
        a = classes.setdefault(cls, {"us": 0.0, "launches": 0, "alg_bytes": 0.0})
        a["us"] += dur / 1e3
        a["launches"] += 1
        a["alg_bytes"] += nbytes

    for name, start, end, grid, wg in rows:
        nm = short(name)
        dur = end - start
        total += dur
        pk = per_kernel.setdefault(nm, [0, 0])
        pk[0] += 1
        pk[1] += dur
        cls, base = class_of(nm, grid if gcol else 0, wg if wcol else 0, n)
        if cls == "level0_passes":
            add(cls, dur, LEVEL0_BYTES[base] * n)
        elif cls == "coarse_levels":
            # (a coarse level's row kernel: one thread per row; 64 B per row is what a 5-to-16-entry f32 ELL row plus
            # its vectors comes to -- an ESTIMATE, the class is latency-bound whatever the bytes; the counter figure
            # is in profiles/<round>_pmc_by_class.json, tools/pmc_by_class.py)
            add(cls, dur, 64.0 * (grid if gcol else 0))
        else:
            add(cls, dur)
    out_classes = {}
    for cls, a in classes.items():
        e = {"share_of_gpu_time": a["us"] * 1e3 / total, "us": a["us"], "launches": a["launches"]}
        if a["alg_bytes"]:
            e["alg_bytes"] = a["alg_bytes"]
            e["GB_per_s"] = a["alg_bytes"] / (a["us"] * 1e-6) / 1e9
            e["frac_of_hbm_peak"] = e["GB_per_s"] / 8000.0
        out_classes[cls] = e
    top = max(per_kernel.items(), key=lambda kv: kv[1][1])
    res = {workload: {"top_kernel_by_time": {"kernel": top[0], "launches": top[1][0],
                                             "share_of_gpu_time": top[1][1] / total,
                                             "avg_us": top[1][1] / top[1][0] / 1e3},
                      "by_class": out_classes, "total_kernel_ms": total / 1e6,
                      "source": "rocprofv3 --kernel-trace of bench.py --workload %s" % workload}}
    return res


def main():
    path, workload, n, nnz = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    print(json.dumps(classify(path, workload, n, nnz, "--columns" in sys.argv), indent=1))


if __name__ == "__main__":
    main()
