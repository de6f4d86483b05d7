# Run ON THE GPU BOX: ordered kernel sequence of one fresh config-5 circuit (presolve + FGMRES), with gaps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/c5_trace -- python3 bench.py --workload cfg5 --steps 1 --warmup 1 --per-step 2 --no-cpu --no-also --concurrent 0 --no-classes > gpurun_out/c5_bench.txt 2>&1
python3 - <<'PY' > gpurun_out/seq_cfg5.txt
import glob, sqlite3
c = sqlite3.connect(glob.glob("gpurun_out/c5_trace/**/*.db", recursive=True)[0])
rows = list(c.execute("select name, start, end from kernels order by start"))
i0 = max(i for i, r in enumerate(rows) if "count_rows" in r[0] and "Matrix" in r[0] and i + 50 < len(rows) and not any("count_rows" in q[0] and "Matrix" in q[0] for q in rows[i+1:i+8]))
# the last circuit starts at the last-but-(k) MatrixStamp count_rows that is followed by a presolve: take the last one whose next MatrixStamp count_rows is the presolved netlist's
cands = [i for i, r in enumerate(rows) if "count_rows" in r[0] and "Matrix" in r[0]]
i0 = cands[-2] if len(cands) >= 2 else cands[-1]
t0 = rows[i0][1]; prev = t0
out = rows[i0:]
print(len(out), "kernels,", (out[-1][2] - t0) / 1e3, "us")
gaps = 0
for k, r in enumerate(out):
    nm = r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]
    gap = (r[1]-prev)/1e3
    if gap > 0: gaps += gap
    if k < 150 or gap > 8:
        print(f"{(r[1]-t0)/1e3:10.1f} us  {nm:50s} {(r[2]-r[1])/1e3:8.1f} us  gap {gap:6.1f}")
    prev = max(prev, r[2])
print("gaps", gaps)
PY
rm -rf gpurun_out/c5_trace
