#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/<round>_pmc_traffic.json.

Usage: pmc_summary.py OUT.json  WORKLOAD:FETCH_DIR:WRITE_DIR [...]

Each DIR is the -d directory of one `rocprofv3 --pmc <COUNTER> --output-format csv` pass over
`python3 bench.py --workload WORKLOAD --no-cpu --no-also` (counters in separate passes, no
trace domains, as MI355X_MICROARCH.md prescribes).  FETCH_SIZE / WRITE_SIZE are reported in KB
per dispatch; on gfx950 FETCH_SIZE undercounts 8-byte-per-lane streaming reads by 2x
(calibrated on f_update: x, r, p, Ap read = 32.0 MB per launch at 1e6 rows -> ~16 MB counted), so reads are
doubled; writes are exact.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

DOMINANT = {  # workload -> (kernel name regex, minimum grid size in threads)
    "cfg2": (r"gemm_sub_kernel<0", 200000),  # (<0>: C -= A B; <0, true, true>: the symmetric C -= A^T B)
    "cfg3": (r"f_(dir_)?spmv<", 200000),
    "cfg4": (r"f_(dir_)?spmv<", 200000),
    "cfg5": (r"k_ell_spmv<", 200000),
}


CIRCUITS_PER_STEP = {"cfg3": 32, "cfg5": 16, "cfg2": 8, "cfg4": 128}  # bench.py's


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.split(r"\(", name)[0]


def read_pass(directory, counter):
    rows = []
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] == counter:
                    rows.append((short(r["Kernel_Name"]), int(r["Grid_Size"]), float(r["Counter_Value"]) * 1024.0))
    return rows


def main():
    out = sys.argv[1]
    result = {"note": __doc__.strip().split("\n\n")[-1].replace("\n", " "), "kernels": {}, "dominant": {}, "totals": {}}
    for spec in sys.argv[2:]:
        workload, fdir, wdir = spec.split(":")
        fetch, write = read_pass(fdir, "FETCH_SIZE"), read_pass(wdir, "WRITE_SIZE")
        acc = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
        for name, grid, val in fetch:
            a = acc[(name, grid)]
            a[0] += 1
            a[1] += 2.0 * val
        for name, grid, val in write:
            a = acc[(name, grid)]
            a[2] += 1
            a[3] += val
        per_kernel = {}
        for (name, grid), (nf, rd, nw, wr) in acc.items():
            if nf == 0 or nw == 0:
                continue
            per_kernel[(name, grid)] = (nf, rd / nf, wr / nw)
        top = sorted(per_kernel.items(), key=lambda kv: -kv[1][0] * (kv[1][1] + kv[1][2]))[:12]
        for (name, grid), (n, rd, wr) in top:
            result["kernels"][f"{workload}:{name}:{grid}"] = {
                "launches": n, "read_bytes_corrected": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr}
        # everything the workload moved, and per circuit (bench.py --steps 1 --warmup 1: two steps)
        total_bytes = sum(n * (rd + wr) for (n, rd, wr) in per_kernel.values())
        circuits = 2 * CIRCUITS_PER_STEP.get(workload, 1)
        result["totals"][workload] = {"hbm_bytes": total_bytes, "dispatches_counted": sum(n for n, _, _ in per_kernel.values()),
                                      "circuits": circuits, "hbm_bytes_per_circuit": total_bytes / circuits}
        pat, min_grid = DOMINANT.get(workload, (None, 0))
        if pat:
            sel = [(n, rd, wr) for (name, grid), (n, rd, wr) in per_kernel.items()
                   if re.search(pat, name) and grid >= min_grid]
            total = sum(n for n, _, _ in sel)
            if total:
                result["dominant"][workload] = {
                    "kernel": pat, "min_grid_threads": min_grid, "launches": total,
                    "hbm_bytes_per_launch": sum(n * (rd + wr) for n, rd, wr in sel) / total}
    with open(out, "w") as f:
        json.dump(result, f, indent=1)
    print(json.dumps(result["dominant"], indent=1))


if __name__ == "__main__":
    main()
