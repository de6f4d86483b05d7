# Run ON THE GPU BOX: the row count at which config 2's block elimination drops from 512- to 256-wide blocks (NODAL_BI_SWITCH,
# default 5632) and the CUs left to the chain (NODAL_PANEL_CUS, default 32), re-scanned after the chain got shorter.
cd "$GRAFT_REPO_ROOT"
for v in "" "NODAL_BI_SWITCH=3584" "NODAL_BI_SWITCH=4608" "NODAL_BI_SWITCH=6656" "NODAL_BI_SWITCH=7680" "NODAL_BI_SWITCH=9728" "NODAL_PANEL_CUS=16" "NODAL_PANEL_CUS=24" "NODAL_PANEL_CUS=48" ""; do
  echo "== [$v]"
  env $v timeout -k 10 200 python3 bench.py --workload cfg2 --steps 4 --warmup 1 --no-cpu --no-also --concurrent 0 --no-classes 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms_per_solve', round(d['ms_per_solve'],3), 'resid', d['scaled_residual'])"
done
