# Run ON THE GPU BOX: kernel statistics of the sparse direct route on config 5's matrix (two solves: analysis + numeric, numeric only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/d_trace -- python3 -c "
import sys
sys.path.insert(0, 'tools'); sys.argv=['x']
import direct_probe as d
from nodal_amd import generators as gen
d.run('cfg5(1000)', gen.cfg5_table(1000), ref=False)
" > gpurun_out/d_run.txt 2>&1
python3 tools/prof_db.py gpurun_out/d_trace 30 > gpurun_out/d_kernels.txt
rm -rf gpurun_out/d_trace
