# Run ON THE GPU BOX: the plain default bench line with its wall time -> gpurun_out/bench_s4.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
t0=$(date +%s)
timeout -k 10 600 python3 bench.py > gpurun_out/bench_s4.json 2> gpurun_out/bench_s4.err; rc=$?
echo "wall $(( $(date +%s) - t0 )) s, rc $rc"
[ $rc -eq 0 ] || { tail gpurun_out/bench_s4.err; exit $rc; }
python3 -c "
import json;d=json.loads(open('gpurun_out/bench_s4.json').read().strip().splitlines()[-1])
print('cfg3', d['value'], d['ms_per_solve'], d['solver'], 'conc', d['concurrent']['circuits_per_sec'], 'reuse', d['reuse_symbolic']['ms_per_solve'])
a=d['also']; print('cfg2', a['cfg2']['ms_per_solve'], 'cfg4', a['cfg4']['circuits_per_sec'], 'cfg5', a['cfg5']['ms_per_solve'], 'direct', a['sparse_direct']['repeated_ms_analysis_kept'], 'sweep', a['resistance_sweep']['repeated_s'])
print('roofline', d['roofline']['frac'], d['roofline']['traffic'], 'cpu', d['cpu_baseline'])
"
