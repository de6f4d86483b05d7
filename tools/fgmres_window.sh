for w in 99 16 12 8 4; do
  export NODAL_FGMRES_WINDOW=$w
  timeout -k 10 200 python bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu --no-also --concurrent 0 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('window $w:', round(d['ms_per_step']/16,3), 'ms', d.get('solver'), d.get('scaled_residual'))"
done
