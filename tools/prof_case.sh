#!/bin/bash
# kernel-time profile of one tools/topologies.py case (run through gpurun): prof_case.sh CASE
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_case
rm -rf $O && mkdir -p $O
NODAL_TOPO_NO_ORACLE=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o case -- python3 tools/topologies.py "$1" > $O/log.txt 2>&1
tail -1 $O/log.txt | cut -c1-120
