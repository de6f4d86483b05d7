#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the counting atomics of the stamping's symbolic phase against the part's atomic
# rate (round 4's review: "close it with counters").  One --pmc pass per counter over a cfg3 run, the stamping kernels'
# rows kept, and the kernel trace of the same command for their durations.
set -e
O=${1:-gpurun_out/pmc_stamping}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $O
ARGS="bench.py --workload cfg3 --steps 1 --warmup 1 --per-step 2 --no-cpu --no-also --concurrent 0 --no-classes"
for c in TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum SQ_WAIT_ANY SQ_WAVE_CYCLES WRITE_SIZE FETCH_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/$c -- python3 $ARGS > $O/$c.log 2>&1 || echo "pass $c failed"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $ARGS > $O/trace.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, os, sys, collections
O = sys.argv[1]
keep = ("count_rows", "emit_tuples", "row_heads_short", "fill_rows_short", "fold_matrix_stream", "fold_rhs", "collect_tuples", "group_few")
def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for cdir in sorted(glob.glob(os.path.join(O, "*"))):
    for path in glob.glob(os.path.join(cdir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            nm = short(r["Kernel_Name"])
            for k in keep:
                if k in nm and int(r["Grid_Size"]) > 100000:
                    vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for path in glob.glob(os.path.join(O, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        nm = short(r["Name"])
        for k in keep:
            if k in nm:
                dur[k] = float(r["AverageNs"]) / 1e3
with open(os.path.join(O, "stamping_counters.txt"), "w") as f:
    def p(*a):
        print(*a); print(*a, file=f)
    p("stamping kernels of one cfg3 circuit (1e6 nodes, 1 998 001 components): rocprofv3 --pmc, one counter per pass")
    for k in keep:
        if k not in vals: continue
        c = {n: sum(v) / len(v) for n, v in vals[k].items()}
        us = dur.get(k)
        line = f"{k:22s} {us if us else float('nan'):7.1f} us"
        if "TCC_ATOMIC_sum" in c:
            line += f"  L2 atomics {c['TCC_ATOMIC_sum']:.3e}"
            if us: line += f" = {c['TCC_ATOMIC_sum'] / us / 1e3:.1f} G atomics/s"
        if "TCC_REQ_sum" in c: line += f"  L2 requests {c['TCC_REQ_sum']:.3e}"
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c['TCC_HIT_sum'] + c['TCC_MISS_sum'] > 0:
            line += f"  L2 hit {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.2f}"
        if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
            line += f"  waves waiting {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.2f}"
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            b = 2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024
            line += f"  HBM {b / 1e6:.1f} MB"
            if us: line += f" = {b / us / 1e3:.0f} GB/s"
        p(line)
PY
rm -rf $O/TCC_* $O/SQ_* $O/WRITE_SIZE $O/FETCH_SIZE $O/trace
