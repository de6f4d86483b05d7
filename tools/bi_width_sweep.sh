for v in "" "NODAL_BI_WIDTH=128" "NODAL_BI_SWITCH2=2048" "NODAL_BI_SWITCH2=4096" "NODAL_BI_SWITCH2=6144" "NODAL_BI_SWITCH2=8192"; do
  echo "== $v"
  env $v timeout -k 10 200 python bench.py --workload cfg2 --steps 3 --warmup 1 --no-cpu --no-also --concurrent 0 --no-classes 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ms_per_solve', round(d['ms_per_solve'],2), 'resid', d['scaled_residual'])"
done
