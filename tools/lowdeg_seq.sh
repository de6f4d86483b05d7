# Run ON THE GPU BOX: ordered kernel sequence of one repeated solve through the low-degree elimination (a ladder of 1e5 sections)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/ld_trace -- python3 -c "
import sys
sys.path.insert(0, '.')
from nodal_amd import _ffi, generators as gen
t = gen.ladder_table(100000)
h = _ffi.Handle(0); h.upload(t); h.assemble_symbolic(); h.assemble_numeric()
for _ in range(3):
    x, info, it, rr = h.solve_sparse()
h.close()
" > gpurun_out/ld_run.txt 2>&1
python3 - <<'PY' > gpurun_out/seq_lowdeg.txt
import glob, sqlite3
c = sqlite3.connect(glob.glob("gpurun_out/ld_trace/**/*.db", recursive=True)[0])
rows = list(c.execute("select name, start, end from kernels order by start"))
# the last solve: from the last schur_values that is preceded by a recover_x (i.e. the last solve's first)
firsts = [i for i, r in enumerate(rows) if "schur_values" in r[0] and (i == 0 or "schur_values" not in rows[i-1][0])]
# find start of last solve: the first schur_values after the last recover_x of the previous solve
rec = [i for i, r in enumerate(rows) if "recover_x" in r[0]]
# group recover_x runs
last_end = rec[-1]
k = len(rec) - 1
while k > 0 and rec[k-1] == rec[k] - 1: k -= 1
# previous solve's last recover
prev_last = rec[k-1] if k > 0 else -1
i0 = prev_last + 1
t0 = rows[i0][1]; prev = t0
out = rows[i0:last_end+1]
print(len(out), "kernels,", (out[-1][2] - t0) / 1e3, "us")
for r in out[:60]:
    nm = r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]
    print(f"{(r[1]-t0)/1e3:10.1f} us  {nm:50s} {(r[2]-r[1])/1e3:8.1f} us  gap {(r[1]-prev)/1e3:6.1f}")
    prev = max(prev, r[2])
PY
rm -rf gpurun_out/ld_trace
