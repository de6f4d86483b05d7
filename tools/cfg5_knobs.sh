cd "$GRAFT_REPO_ROOT"
for v in "" "NODAL_FGMRES_WINDOW=4" "NODAL_FGMRES_WINDOW=6" "NODAL_FGMRES_WINDOW=12" "NODAL_FGMRES_WINDOW=99" "NODAL_SA_TAIL_NU=2" "NODAL_SA_TAIL_NU=4" ""; do
  echo "== [$v]"
  env $v timeout -k 10 200 python3 bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu --no-also --concurrent 0 --no-classes 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms_per_solve', round(d['ms_per_solve'],3), d['solver'], d['scaled_residual'])"
done
