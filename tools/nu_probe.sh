# Run ON THE GPU BOX: Jacobi sweeps per level (NODAL_SA_NU) and inside the tail (NODAL_SA_TAIL_NU) over the shapes, the
# topologies and the bench's legs -- which default serves all of them.   usage: tools/nu_probe.sh "ENV..." "ENV..."
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/nu
i=0
for e in "$@"; do
  i=$((i+1))
  echo "==== [$e]"
  env $e timeout -k 10 300 python3 tools/shape_probe.py grid:100 grid:316 grid:562 grid:1000 grid:1200 grid:1600 grid:2000 rgrid:1000 cfg5:700 cfg5:1000 batch:64x140 batch:1024x35 batch:128x100 grid3:80 > gpurun_out/nu/shapes_$i.txt 2>&1 || { tail gpurun_out/nu/shapes_$i.txt; exit 1; }
  cat gpurun_out/nu/shapes_$i.txt
  env $e timeout -k 10 300 python3 tools/topologies.py > gpurun_out/nu/topo_$i.txt 2>&1 || { tail gpurun_out/nu/topo_$i.txt; exit 1; }
  cat gpurun_out/nu/topo_$i.txt
  env $e timeout -k 10 400 python3 bench.py --no-cpu --no-classes --steps 6 > gpurun_out/nu/bench_$i.json 2> gpurun_out/nu/bench_$i.err || { tail gpurun_out/nu/bench_$i.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/nu/bench_$i.json').read().strip().splitlines()[-1])
print('cfg3', round(d['value'],1), round(d['ms_per_solve'],3), d['solver'], 'conc', [(c['streams'],c['symbolic_phases_kept'],round(c['circuits_per_sec'],1)) for c in d['concurrent']['all']], 'reuse', round(d['reuse_symbolic']['ms_per_solve'],3))
a=d['also']; print('cfg4', round(a['cfg4']['circuits_per_sec']), a['cfg4']['solver'], 'cfg5', round(a['cfg5']['ms_per_solve'],3), a['cfg5']['solver'], 'sweep', round(a['resistance_sweep']['repeated_s'],4))
"
done
