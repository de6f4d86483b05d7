# Run ON THE GPU BOX: the third session's closing records -- GPU suite, plain default bench line, kernel statistics of the same command.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/fin
O=gpurun_out/fin
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1; rc=$?; tail -3 $O/gpu_suite.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; rc=$?; [ $rc -eq 0 ] || { tail $O/bench_default.err; exit $rc; }
python3 -c "
import json;d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print('cfg3', d['value'], d['ms_per_solve'], d['solver'], 'conc', d['concurrent']['circuits_per_sec'], 'reuse', d['reuse_symbolic']['ms_per_solve'])
a=d['also']; print('cfg2', a['cfg2']['ms_per_solve'], 'cfg4', a['cfg4']['circuits_per_sec'], 'cfg5', a['cfg5']['ms_per_solve'], 'direct', a['sparse_direct'], 'sweep', a['resistance_sweep'])
print('roofline', d['roofline']['frac'], d['roofline']['traffic'])
"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/default -o default -- python3 bench.py --concurrent 0 --no-classes > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err; rc=$?
cp $O/default/default_kernel_stats.csv $O/bench_default_kernel_stats.csv 2>/dev/null || find $O/default -name "*kernel_stats.csv" -exec cp {} $O/bench_default_kernel_stats.csv \;
rm -rf $O/default
head -5 $O/bench_default_kernel_stats.csv
exit $rc
