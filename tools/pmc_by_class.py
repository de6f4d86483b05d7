#!/usr/bin/env python3
"""HBM traffic of one circuit BY CLASS OF KERNEL, from counters (profiles/<round>_pmc_by_class.json).

    pmc_by_class.py OUT.json WORKLOAD N CIRCUITS FETCH_DIR WRITE_DIR [KERNEL_TRACE_DIR]

FETCH_DIR / WRITE_DIR: the -d directories of two `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv`
passes over `python3 bench.py --workload WORKLOAD --steps 1 --warmup 1 ...` (one counter per pass, no trace
domains: MI355X_MICROARCH.md).  Every dispatch is sorted into the classes of tools/prof_classes.py (kernel name AND
grid size, the same rule bench.py's `roofline.by_class` uses for the TIMES) and its bytes are added up: FETCH_SIZE and
WRITE_SIZE come in KB per dispatch; on gfx950 FETCH_SIZE undercounts 8-byte-per-lane streaming reads by 2x (calibrated
on f_update, tools/pmc_summary.py), so reads are doubled -- generous for the gathering kernels --, writes are exact.
CIRCUITS = circuits the profiled run solved (bench.py --steps 1 --warmup 1: two steps of CIRCUITS_PER_STEP).
With a kernel-trace directory of the same command the classes' times are added, and with them bytes per second.
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import prof_classes  # noqa: E402


def read_pass(directory, counter):
    rows = []
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] == counter:
                    wg = int(r.get("Workgroup_Size", 0) or 0)
                    rows.append((prof_classes.short(r["Kernel_Name"]), int(r["Grid_Size"]), wg,
                                 float(r["Counter_Value"]) * 1024.0))
    return rows


def main():
    out, workload, n, circuits, fdir, wdir = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
    trace = sys.argv[7] if len(sys.argv) > 7 else None
    acc = collections.defaultdict(lambda: {"dispatches": 0, "read_bytes_corrected": 0.0, "write_bytes": 0.0})
    per_kernel = collections.defaultdict(lambda: [0, 0.0])
    for name, grid, wg, val in read_pass(fdir, "FETCH_SIZE"):
        cls, base = prof_classes.class_of(name, grid, wg or 256, n)
        acc[cls]["dispatches"] += 1
        acc[cls]["read_bytes_corrected"] += 2.0 * val
        per_kernel[(cls, base)][0] += 1
        per_kernel[(cls, base)][1] += 2.0 * val
    for name, grid, wg, val in read_pass(wdir, "WRITE_SIZE"):
        cls, base = prof_classes.class_of(name, grid, wg or 256, n)
        acc[cls]["write_bytes"] += val
        per_kernel[(cls, base)][1] += val
    times = None
    if trace:
        times = prof_classes.classify(trace, workload, n, 0)[workload]["by_class"]
    total = sum(a["read_bytes_corrected"] + a["write_bytes"] for a in acc.values())
    classes = {}
    for cls, a in sorted(acc.items(), key=lambda kv: -(kv[1]["read_bytes_corrected"] + kv[1]["write_bytes"])):
        b = a["read_bytes_corrected"] + a["write_bytes"]
        e = {"dispatches_per_circuit": a["dispatches"] / circuits, "hbm_bytes_per_circuit": b / circuits,
             "read_bytes_corrected_per_circuit": a["read_bytes_corrected"] / circuits,
             "write_bytes_per_circuit": a["write_bytes"] / circuits, "share_of_bytes": b / total if total else None,
             "hbm_bytes_per_dispatch": b / a["dispatches"] if a["dispatches"] else None}
        if times and cls in times:
            t = times[cls]
            e["us_per_circuit_traced"] = t["us"] / circuits
            e["share_of_gpu_time"] = t["share_of_gpu_time"]
            e["counter_GB_per_s"] = b / (t["us"] * 1e-6) / 1e9 if t["us"] else None
            e["frac_of_hbm_peak"] = e["counter_GB_per_s"] / 8000.0 if e["counter_GB_per_s"] else None
            if t.get("alg_bytes"):
                e["alg_bytes_per_circuit"] = t["alg_bytes"] / circuits
        classes[cls] = e
    top = sorted(per_kernel.items(), key=lambda kv: -kv[1][1])[:14]
    result = {"workload": workload, "circuits": circuits, "hbm_bytes_per_circuit": total / circuits,
              "by_class": classes,
              "largest_kernels_by_bytes": [{"class": c, "kernel": k, "dispatches_per_circuit": v[0] / circuits,
                                            "hbm_bytes_per_circuit": v[1] / circuits} for (c, k), v in top],
              "how": __doc__.strip().split("\n\n")[1].replace("\n", " ")}
    with open(out, "w") as f:
        json.dump(result, f, indent=1)
    print(json.dumps({k: (round(v["hbm_bytes_per_circuit"] / 1e6, 1), "MB") for k, v in classes.items()}))


if __name__ == "__main__":
    main()
