cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
NODAL_TRACE=1 timeout -k 10 120 python tools/sa_probe.py ${1:-1000} 3 2>&1 | grep -E "sagg\] (levels|[0-9]|decl)|run 2|normwise" | tail -4
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_sa8 -- python tools/sa_probe.py ${1:-1000} 4 > gpurun_out/r2_p9.log 2>&1; python tools/prof_db.py gpurun_out/prof_sa8 ${2:-40}
