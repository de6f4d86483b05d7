# Builds tools/panel_probe.hip here (cross-compiled): the register panel's bodies are cut out of csrc/sparse_direct.hip.
# Run the binary on the GPU box: gpurun -- 'timeout -k 5 60 build/panel_probe'
cd "$(dirname "$0")/.." && mkdir -p build
python3 - <<'PY'
s = open("nodal_amd/csrc/sparse_direct.hip").read()
i = s.index("constexpr int PNB = 16;")
j = s.index("template <int RPT>\n__global__ __launch_bounds__(256) void panel_factor_regs(")
open("build/panel_bodies.inc", "w").write(s[i:j])
PY
FLAGS=""
grep -q panel_factor_regs_body1 build/panel_bodies.inc && FLAGS="-DPANEL_ONE_BARRIER"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -mllvm -pragma-unroll-threshold=1000000 \
  -Wno-unused-value -Wno-unused-function -Wno-pass-failed $FLAGS -o build/panel_probe tools/panel_probe.hip
