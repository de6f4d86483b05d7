#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: HBM traffic of one circuit by class of kernel, from counters.
#   tools/pmc_classes.sh WORKLOAD N OUT     e.g.  tools/pmc_classes.sh cfg3 999999 gpurun_out/pmc_cfg3
# Two --pmc passes (one counter each, no trace domains) and one kernel trace of the same command.
set -e
W=${1:-cfg3}; N=${2:-999999}; O=${3:-gpurun_out/pmc_$W}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $O
ARGS="bench.py --workload $W --steps 1 --warmup 1 --per-step 4 --no-cpu --no-also --concurrent 0 --no-classes"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/$c -- python3 $ARGS > $O/$c.log 2>&1
done
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/trace -- python3 $ARGS > $O/trace.log 2>&1
python3 tools/pmc_by_class.py $O/pmc_by_class.json $W $N 8 $O/FETCH_SIZE $O/WRITE_SIZE $O/trace
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE $O/trace
