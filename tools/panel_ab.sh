# Run ON THE GPU BOX: the register panel with one barrier per column + DPP arg-max (default) against the two-barrier
# shuffle panel (NODAL_DIRECT_PANEL_1B=0): bits of the solution, factorisation times A/B/A/B, then the direct route's tests.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for v in 0 1 0 1; do
  NODAL_DIRECT_PANEL_1B=$v timeout -k 10 250 python3 tools/direct_time.py 1000 > gpurun_out/panel_ab_$v.txt 2>&1 || { tail -20 gpurun_out/panel_ab_$v.txt; exit 1; }
  echo "NODAL_DIRECT_PANEL_1B=$v"; grep -E "all levels|numeric factorisation|cfg5\(" gpurun_out/panel_ab_$v.txt | tail -4
done
python3 - <<'PY'
import hashlib, numpy as np, os, subprocess, sys
code = r'''
import os, sys, hashlib, numpy as np
sys.path.insert(0, ".")
from nodal_amd import _ffi, generators as gen
from nodal_amd.lowering import lower_generated
os.environ["NODAL_GENERAL_ROUTE"] = "direct"
for n in (300, 700):
    h = _ffi.Handle(0); h.upload(lower_generated(gen.cfg5_components(n)))
    for it in range(2):
        info = h.run(False, sparse=True); h.synchronize()
    print(n, info, hashlib.sha256(np.ascontiguousarray(h.download_x()).tobytes()).hexdigest()[:16])
    h.close()
'''
out = []
for v in ("0", "1"):
    env = dict(os.environ, NODAL_DIRECT_PANEL_1B=v)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    print("NODAL_DIRECT_PANEL_1B=" + v, r.stdout.strip().replace("\n", " | "), r.stderr.strip()[-300:])
    out.append(r.stdout)
print("solutions bit-identical:", out[0] == out[1] and out[0] != "")
PY
timeout -k 10 600 python3 -m pytest tests/test_gpu_direct.py -m gpu -x -q 2>&1 | tail -5
