for m in 6 5 4 3; do
  NODAL_SA_MIS=$m timeout -k 10 200 python bench.py --workload cfg3 --steps 10 --warmup 2 --no-cpu --no-also --concurrent 0 > gpurun_out/b_mis$m.log 2>&1
  python3 -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/b_mis$m.log') if l.startswith('{')][0])
print('MIS rounds $m:', round(d['ms_per_step']/32,3), 'ms', d.get('phase_ms'), d.get('iterations'), d.get('solver'))"
done
