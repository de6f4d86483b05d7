# Run ON THE GPU BOX: kernel statistics and the ordered sequence of two diagonal-block steps of config 2's dense solve.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/c2_trace -- python3 bench.py --workload cfg2 --steps 1 --warmup 1 --per-step 1 --no-cpu --no-also --concurrent 0 --no-classes > gpurun_out/c2_bench.txt 2>&1
python3 tools/prof_db.py gpurun_out/c2_trace 40 > gpurun_out/c2_kernels.txt
python3 - <<'PY' > gpurun_out/c2_seq.txt
import glob, sqlite3
c = sqlite3.connect(glob.glob("gpurun_out/c2_trace/**/*.db", recursive=True)[0])
rows = list(c.execute("select name, start, end from kernels order by start"))
# the last solve: from the last fold_matrix* on
i0 = max(i for i, r in enumerate(rows) if "fold_matrix" in r[0])
t0 = rows[i0][1]; prev = t0
out = rows[i0:]
print(len(out), "kernels,", (out[-1][2] - t0) / 1e3, "us")
for r in out[:120] + out[len(out)//2:len(out)//2 + 60]:
    nm = r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]
    print(f"{(r[1]-t0)/1e3:10.1f} us  {nm:50s} {(r[2]-r[1])/1e3:8.1f} us  gap {(r[1]-prev)/1e3:6.1f}")
    prev = max(prev, r[2])
PY
rm -rf gpurun_out/c2_trace
