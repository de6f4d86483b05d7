#!/bin/bash
# kernel-time profile of fresh config-3 circuits (run through gpurun): per-kernel totals + the ordered setup sequence
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof_cfg3}
cd /tmp && export TMPDIR=/tmp
rm -rf $O && mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/prof -- python3 bench.py --workload cfg3 --steps 1 --warmup 1 --per-step 4 --no-cpu --no-also --concurrent 0 --no-classes > $O/run.log 2>&1
python3 tools/prof_db.py $O/prof 60 > $O/kernels.txt
python3 tools/prof_sequence.py $O/prof count_rows f_init 200 > $O/setup_sequence.txt || true
head -64 $O/kernels.txt
