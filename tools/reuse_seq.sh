# Run ON THE GPU BOX: ordered kernel sequence of one config-3 circuit with the symbolic phases kept (values-only refresh), up to f_init
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/ru_trace -- python3 -c "
import sys
sys.path.insert(0, '.')
from nodal_amd import _ffi, generators as gen
t = gen.grid_table(1000)
h = _ffi.Handle(0); h.set_option(_ffi.OPT_EXTRA_STREAMS, 1); h.upload(t)
for _ in range(4):
    assert h.run(False, 0, True) == 0
h.close()
" > gpurun_out/ru_run.txt 2>&1
python3 - <<'PY' > gpurun_out/seq_reuse.txt
import glob, sqlite3
c = sqlite3.connect(glob.glob("gpurun_out/ru_trace/**/*.db", recursive=True)[0])
rows = list(c.execute("select name, start, end from kernels order by start"))
i1 = max(i for i, r in enumerate(rows) if "f_init" in r[0])
i0 = max(i for i in range(i1) if "init_numeric" in rows[i][0])
t0 = rows[i0][1]; prev = t0
gaps = 0
for r in rows[i0:i1 + 1]:
    nm = r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]
    gap = (r[1]-prev)/1e3
    if gap > 0: gaps += gap
    print(f"{(r[1]-t0)/1e3:10.1f} us  {nm:50s} {(r[2]-r[1])/1e3:8.1f} us  gap {gap:6.1f}")
    prev = max(prev, r[2])
print(i1 - i0 + 1, "kernels,", (rows[i1][2]-t0)/1e3, "us, gaps", gaps)
PY
rm -rf gpurun_out/ru_trace
