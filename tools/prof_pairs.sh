#!/bin/bash
# kernel-time profile of a pair sweep on grid(N) through the factor-once route (run through gpurun): prof_pairs.sh N NPAIRS OUT
set -e
N=${1:-1000}; P=${2:-64}; O=$GRAFT_REPO_ROOT/gpurun_out/${3:-prof_pairs}
cd /tmp && export TMPDIR=/tmp
rm -rf $O && mkdir -p $O
NODAL_PAIRS_DIRECT=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o pairs -- python3 $GRAFT_REPO_ROOT/tools/pairs_probe.py $N $P direct > $O/run.log 2>&1
grep "pairs in" $O/run.log
cd $GRAFT_REPO_ROOT && python3 tools/prof_db.py $O/prof 28 > $O/kernels.txt && cat $O/kernels.txt
