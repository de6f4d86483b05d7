#!/bin/bash
# Run ON THE GPU BOX: config 3 alone under a list of environment settings (one bench line each).
# usage: tools/exp_cfg3.sh "NAME=VAL NAME2=VAL" "..." ; an empty string = the defaults
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for e in "$@"; do
  echo "== [$e]"
  env $e timeout -k 10 200 python3 bench.py --workload cfg3 --steps 8 --warmup 2 --no-cpu --no-also --concurrent 0 --no-classes 2> gpurun_out/exp.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_solve %.3f  value %.1f  phases %s  it %s  reuse %s' % (d['ms_per_solve'], d['value'], {k: round(v,3) for k,v in d['phase_ms'].items()}, d['solver']['iterations'], d.get('reuse_symbolic', {}).get('ms_per_solve')))
"
  grep "\[sagg\] [0-9]* iterations" gpurun_out/exp.err | sort | uniq -c | head -3
done
