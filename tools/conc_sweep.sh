cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for c in 4 5 6 8; do
  timeout -k 10 200 python3 bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu --no-also --concurrent $c --no-classes > gpurun_out/conc_$c.json 2>gpurun_out/conc_err.txt || { tail gpurun_out/conc_err.txt; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/conc_$c.json').read().strip().splitlines()[-1]);print($c, [(x['streams'],x['symbolic_phases_kept'],round(x['circuits_per_sec'],1)) for x in d['concurrent']['all']])"
done
