#!/bin/bash
# PMC passes over the latency chains: the direct route's factorisation (cfg5(1000)) and config 2's dense chain
# (run on the GPU box; counters only, no trace domains).  -> gpurun_out/pmc_chains/{direct,cfg2}.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_chains
mkdir -p $O
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY"
P3="SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
n=0
for P in "$P1" "$P2" "$P3"; do
  n=$((n+1))
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $O/d$n -- python3 tools/direct_time.py 1000 > $O/d$n.log 2>&1 || { tail -5 $O/d$n.log; exit 1; }
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $O/c$n -- python3 bench.py --workload cfg2 --steps 1 --warmup 1 --per-step 1 --no-cpu --no-also --concurrent 0 --no-classes > $O/c$n.log 2>&1 || { tail -5 $O/c$n.log; exit 1; }
done
PMC_KERNELS='level_panel|level_swaps|level_rank|factor_fronts|extend_add|invert_diag|forward_level|backward_level|level_fwd|level_bwd' python3 tools/pmc_coarse.py $O/direct.json $O/d1 $O/d2 $O/d3 > $O/direct.txt 2>&1
PMC_KERNELS='gj128|gemm_small|gemm_sub|copy_block|transpose_block|bs_block' python3 tools/pmc_coarse.py $O/cfg2.json $O/c1 $O/c2 $O/c3 > $O/cfg2.txt 2>&1
rm -rf $O/d1 $O/d2 $O/d3 $O/c1 $O/c2 $O/c3
wc -l $O/direct.txt $O/cfg2.txt
