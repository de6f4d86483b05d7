cd $GRAFT_REPO_ROOT
NODAL_DIRECT_PANEL_SHORT=0 bash tools/prof_direct.sh 1000 pd0 > gpurun_out/pd0.txt 2>&1 || { tail gpurun_out/pd0.txt; exit 1; }
NODAL_DIRECT_PANEL_SHORT=1 bash tools/prof_direct.sh 1000 pd1 > gpurun_out/pd1.txt 2>&1 || { tail gpurun_out/pd1.txt; exit 1; }
rm -rf gpurun_out/pd0/prof gpurun_out/pd1/prof
grep -E "panel|total kernel" gpurun_out/pd0.txt; echo; grep -E "panel|total kernel" gpurun_out/pd1.txt
