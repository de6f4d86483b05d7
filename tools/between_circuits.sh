# Run ON THE GPU BOX: what the GPU does between the last iteration of one config-3 circuit and the first stamping kernel of the next
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/bc_trace -- python3 bench.py --workload cfg3 --steps 1 --warmup 1 --per-step 3 --no-cpu --no-also --concurrent 0 --no-classes > gpurun_out/bc_bench.txt 2>&1
python3 - <<'PY' > gpurun_out/seq_between.txt
import glob, sqlite3
c = sqlite3.connect(glob.glob("gpurun_out/bc_trace/**/*.db", recursive=True)[0])
rows = list(c.execute("select name, start, end from kernels order by start"))
starts = [i for i, r in enumerate(rows) if "count_rows" in r[0] and "Matrix" in r[0]]
i1 = starts[-1]
i0 = max(i for i in range(i1) if "f_spmv" in rows[i][0]) - 40
t0 = rows[i0][1]; prev = t0
for r in rows[i0:i1 + 3]:
    nm = r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]
    print(f"{(r[1]-t0)/1e3:10.1f} us  {nm:50s} {(r[2]-r[1])/1e3:8.1f} us  gap {(r[1]-prev)/1e3:6.1f}")
    prev = max(prev, r[2])
PY
rm -rf gpurun_out/bc_trace
