# Run ON THE GPU BOX: a round's closing records -- counters by class (config 3), GPU suite, plain default bench line, kernel
# statistics of the same command.  -> gpurun_out/fin/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/fin
O=gpurun_out/fin
bash tools/pmc_classes.sh cfg3 999999 $O/pmc_cfg3 > $O/pmc_classes.log 2>&1 || { tail $O/pmc_classes.log; exit 1; }
cp $O/pmc_cfg3/pmc_by_class.json profiles/r05_pmc_by_class.json   # (read by bench.py on this box for roofline.traffic)
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1; rc=$?; tail -3 $O/gpu_suite.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; rc=$?; [ $rc -eq 0 ] || { tail $O/bench_default.err; exit $rc; }
python3 -c "
import json;d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print('cfg3', d['value'], d['ms_per_solve'], d['solver'], 'conc', d['concurrent']['circuits_per_sec'], 'reuse', d['reuse_symbolic']['ms_per_solve'])
a=d['also']; print('cfg2', a['cfg2']['ms_per_solve'], 'cfg4', a['cfg4']['circuits_per_sec'], 'cfg5', a['cfg5']['ms_per_solve'], 'direct', a['sparse_direct']['repeated_ms_analysis_kept'], 'sweep', a['resistance_sweep']['repeated_s'])
print('roofline', d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('hbm_bytes_per_circuit'))
"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/default -o default -- python3 bench.py --concurrent 0 --no-classes > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err; rc=$?
find $O/default -name "*kernel_stats.csv" -exec cp {} $O/bench_default_kernel_stats.csv \;
rm -rf $O/default
exit $rc
