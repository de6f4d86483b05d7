for cfg in "0 0" "1 8" "1 16" "1 32" "1 64"; do
  set -- $cfg
  if [ "$1" = "1" ]; then export NODAL_BI_MASKED=1 NODAL_PANEL_CUS=$2; else unset NODAL_BI_MASKED NODAL_PANEL_CUS; fi
  timeout -k 10 200 python bench.py --workload cfg2 --steps 4 --warmup 1 --no-cpu --no-also --concurrent 0 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('masked $1 reserved CUs $2:', round(d['ms_per_step']/8,3), 'ms per solve')"
done
