# Run ON THE GPU BOX: ordered kernel sequence of one config-4 step (128 members as one block system), gaps above 8 us
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/c4_trace -- python3 bench.py --workload cfg4 --steps 2 --warmup 1 --no-cpu --no-also --concurrent 0 --no-classes > gpurun_out/c4_bench.txt 2>&1
python3 - <<'PY' > gpurun_out/seq_cfg4.txt
import glob, sqlite3
c = sqlite3.connect(glob.glob("gpurun_out/c4_trace/**/*.db", recursive=True)[0])
rows = list(c.execute("select name, start, end from kernels order by start"))
cands = [i for i, r in enumerate(rows) if "row_stats" in r[0]]
i1 = cands[-1]
# step start: walk back to the previous f_update (end of the previous step's solve) + 1
i0 = max(i for i in range(i1) if "f_update" in rows[i][0]) + 1
t0 = rows[i0][1]; prev = t0
out = rows[i0:]
print(len(out), "kernels,", (out[-1][2] - t0) / 1e3, "us")
gaps = 0
for k, r in enumerate(out):
    nm = r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]
    gap = (r[1]-prev)/1e3
    if gap > 0: gaps += gap
    if k < 70 or gap > 8:
        print(f"{(r[1]-t0)/1e3:10.1f} us  {nm:50s} {(r[2]-r[1])/1e3:8.1f} us  gap {gap:6.1f}")
    prev = max(prev, r[2])
print("gaps", gaps)
PY
rm -rf gpurun_out/c4_trace
