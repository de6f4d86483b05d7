# Run ON THE GPU BOX: the in-wave 16 x 16 inverse with DPP / readlane exchanges against ds_bpermute --
# the probe (bits, cycles), then config 2 A/B/A/B on this box.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 60 tools/gj16_dpp_probe > gpurun_out/gj16_probe.txt 2>&1 || exit 1
cat gpurun_out/gj16_probe.txt
for v in 0 1 0 1; do
  NODAL_GJ_DPP=$v timeout -k 10 200 python3 bench.py --workload cfg2 --steps 6 --warmup 2 --no-cpu --no-also --concurrent 0 --no-classes > gpurun_out/gj_ab_$v.json 2>gpurun_out/gj_ab_err.txt || exit 1
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/gj_ab_$v.json").read().strip().splitlines()[-1])
print("NODAL_GJ_DPP=$v", d["ms_per_step"], "ms per step", d["value"], d["unit"])
PY
done
