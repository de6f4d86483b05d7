#!/bin/bash
# kernel-time profile of the direct route on cfg5(N) (run through gpurun): prof_direct.sh N OUT
set -e
N=${1:-1000}; O=$GRAFT_REPO_ROOT/gpurun_out/${2:-prof_direct}
cd /tmp && export TMPDIR=/tmp
rm -rf $O && mkdir -p $O
cat > $O/run.py <<PY
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from nodal_amd import _ffi, generators as gen
table = gen.cfg5_table($N)
h = _ffi.Handle(0)
h.set_option(_ffi.OPT_EXTRA_STREAMS, 1)
h.upload(table); h.assemble_symbolic(); h.assemble_numeric()
for rep in range(3):
    t0 = time.time(); x, info, iters, rr = h.solve_sparse(method=_ffi.SPARSE_DIRECT); print("solve %.1f ms info %d iters %d" % ((time.time() - t0) * 1e3, info, iters), flush=True)
h.close()
PY
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o d -- python3 $O/run.py > $O/run.log 2>&1
grep "solve" $O/run.log
cd $GRAFT_REPO_ROOT && python3 tools/prof_db.py $O/prof 30 > $O/kernels.txt && cat $O/kernels.txt
