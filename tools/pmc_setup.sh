#!/bin/bash
# PMC passes over the kernels of a FRESH config-3 solve's setup and stamping (run on the GPU box)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_setup
mkdir -p $O
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/p1 -- python3 tools/sa_probe.py 1000 2 > $O/p1.log 2>&1
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/p2 -- python3 tools/sa_probe.py 1000 2 > $O/p2.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/p3 -- python3 tools/sa_probe.py 1000 2 > $O/p3.log 2>&1
PMC_KERNELS='galerkin|ap_rows|count_rows|emit_tuples|row_heads|fill_rows|fold_matrix|mis_max|mis_update|r_fill|r_count|build_P|csr_to_ell|assign_' python3 tools/pmc_coarse.py $O/summary.json $O/p1 $O/p2 $O/p3 > $O/summary.txt 2>&1
rm -rf $O/p1 $O/p2 $O/p3
cat $O/summary.txt
