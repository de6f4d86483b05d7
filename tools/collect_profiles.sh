#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: kernel-trace summaries of the default
# bench and PMC passes (one counter per pass, no trace domains) for cfg2 / cfg3 / cfg4 / cfg5.
# Outputs land under gpurun_out/prof_final/; tools/pmc_summary.py turns the PMC passes into
# profiles/<round>_pmc_traffic.json afterwards.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_final
mkdir -p $O
# where the GPU time of a config-3 solve goes, by class of kernel (read back by bench.py: roofline.by_class)
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/classes -- python3 bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu --no-also --concurrent 0 --no-classes > $O/classes.log 2>&1
python3 tools/prof_classes.py $O/classes cfg3 999999 4995995 > $O/classes.json
rm -rf $O/classes
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/default -o default -- python3 bench.py --concurrent 0 --no-classes > $O/bench_default.log 2>&1
for w in cfg2 cfg3 cfg4 cfg5; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${w}_$c -- python3 bench.py --workload $w --steps 1 --warmup 1 --no-cpu --no-also --concurrent 0 --no-classes > $O/pmc_${w}_$c.log 2>&1
  done
done
tail -1 $O/bench_default.log
# summaries only travel back (gpurun merges at most 64 MiB): PMC passes -> one JSON, the per-dispatch
# traces and counter files stay on the box
python3 tools/pmc_summary.py $O/pmc_traffic.json cfg2:$O/pmc_cfg2_FETCH_SIZE:$O/pmc_cfg2_WRITE_SIZE \
  cfg3:$O/pmc_cfg3_FETCH_SIZE:$O/pmc_cfg3_WRITE_SIZE cfg4:$O/pmc_cfg4_FETCH_SIZE:$O/pmc_cfg4_WRITE_SIZE \
  cfg5:$O/pmc_cfg5_FETCH_SIZE:$O/pmc_cfg5_WRITE_SIZE > $O/pmc_summary.log 2>&1
cp $O/default/default_kernel_stats.csv $O/bench_default_kernel_stats.csv
rm -rf $O/default $O/pmc_cfg*_FETCH_SIZE $O/pmc_cfg*_WRITE_SIZE
# (a plain `python bench.py > gpurun_out/bench_plain.log` gives profiles/<round>_bench_default.json)
