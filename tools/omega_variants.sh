# Run ON THE GPU BOX: config 3 / batch / config 5 under library variants built with other Jacobi weights
# (-DNODAL_SA_OMEGA=..., build/variants/libnodal_hip_omXXX.so; the box's copy of the library is swapped between runs).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for tag in 085 075 080 090 095 085; do
  cp build/variants/libnodal_hip_om$tag.so nodal_amd/libnodal_hip.so
  echo "==== OMEGA 0.$tag"
  bash tools/exp_cfg3.sh ""
  timeout -k 10 200 python3 tools/shape_probe.py grid:316 batch:128x100 cfg5:1000 rgrid:1000 2>&1 | tail -4
done
cp build/variants/libnodal_hip_om085.so nodal_amd/libnodal_hip.so
echo "==== prolongator weights (OMEGA 0.85)"
bash tools/exp_cfg3.sh "NODAL_SA_OMEGA_P=0.65,0.85" "NODAL_SA_OMEGA_P=0.75,0.85" "NODAL_SA_OMEGA_P=0.70,0.80" "NODAL_SA_OMEGA_P=0.70,0.90" "NODAL_SA_OMEGA_P=0.75,0.90" "NODAL_SA_KLEVELS=2" ""
