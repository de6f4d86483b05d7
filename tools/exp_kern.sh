#!/bin/bash
# Run ON THE GPU BOX: kernel averages of fresh config-3 circuits under one environment setting
# usage: tools/exp_kern.sh NAME "ENV=VAL ..." [pattern]
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1
rm -rf $O && mkdir -p $O
cd $GRAFT_REPO_ROOT
for kv in $2; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/prof -- python3 bench.py --workload ${WORKLOAD:-cfg3} --steps 1 --warmup 1 --per-step 4 --no-cpu --no-also --concurrent 0 --no-classes > $O/run.log 2>&1
python3 tools/prof_db.py $O/prof ${TOPN:-80} > $O/kernels.txt
rm -rf $O/prof
echo "== $1 [$2]"; grep -m1 "levels (rows" $O/run.log
grep -E "${3:-k_|f_}" $O/kernels.txt
