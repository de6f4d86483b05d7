# Run ON THE GPU BOX: smoothing sweeps at the levels below the first coarse one (NODAL_SA_NU = level 0 / 1 / deeper), A/B/A/B
for nu in 111 112 111 112; do
  for w in cfg3 cfg4; do
  NODAL_SA_NU=$nu timeout -k 10 200 python bench.py --workload $w --steps 8 --warmup 2 --no-cpu --no-also --concurrent 0 --no-classes 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$nu $w', round(d['value'],1),round(d['ms_per_solve'],3),d['solver'])"
  done
done
