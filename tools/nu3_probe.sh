# Run ON THE GPU BOX: a THIRD Jacobi sweep at level 0 (NODAL_SA_NU=312; until this script the knob's digits stopped at 2 and
# "312" silently meant the default) against the default 212, A/B/A/B; config 3 fresh, then config 5 and config 4.
cd "$GRAFT_REPO_ROOT"
run() {  # workload, env
  env $2 timeout -k 10 200 python3 bench.py --workload $1 --steps 3 --warmup 1 --no-cpu --no-also --concurrent 0 --no-classes 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  $1 [$2] value', round(d['value'],1), 'ms_per_solve', round(d.get('ms_per_solve') or 0,3), d['solver'], d.get('scaled_residual'))"
}
for v in "NODAL_SA_NU=212" "NODAL_SA_NU=312" "NODAL_SA_NU=212" "NODAL_SA_NU=312" "NODAL_SA_NU=313" "NODAL_SA_NU=322" "NODAL_SA_NU=323"; do run cfg3 "$v" || exit 1; done
for v in "NODAL_SA_NU=212" "NODAL_SA_NU=312"; do run cfg5 "$v" || exit 1; run cfg4 "$v" || exit 1; done
