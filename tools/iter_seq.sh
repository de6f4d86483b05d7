# Run ON THE GPU BOX: the ordered kernel sequence of ONE outer iteration of a config-3 solve (between two f_update launches).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/it_trace -- python3 bench.py --workload cfg3 --steps 1 --warmup 1 --per-step 2 --no-cpu --no-also --concurrent 0 --no-classes > gpurun_out/it_bench.txt 2>&1 || exit 1
python3 - <<'PY' > gpurun_out/iter_seq.txt
import glob, sqlite3
c = sqlite3.connect(glob.glob("gpurun_out/it_trace/**/*.db", recursive=True)[0])
rows = list(c.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "f_update" in r[0]]
# iterations 9..12 of the last solve (a frozen and an adaptive cycle among them)
last = idx[-14:-9]
for a, b in zip(last[:-1], last[1:]):
    t0 = rows[a][1]; prev = t0
    print(f"--- one iteration: {b - a} launches, {(rows[b][1] - t0) / 1e3:.1f} us")
    for r in rows[a:b]:
        nm = r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
        print(f"{(r[1]-t0)/1e3:8.1f} us  {nm:48s} {(r[2]-r[1])/1e3:6.1f} us  gap {(r[1]-prev)/1e3:5.1f}")
        prev = max(prev, r[2])
PY
rm -rf gpurun_out/it_trace
tail -3 gpurun_out/iter_seq.txt
