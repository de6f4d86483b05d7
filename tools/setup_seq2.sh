cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/seq_trace -- python3 bench.py --workload cfg3 --steps 1 --warmup 1 --per-step 2 --no-cpu --no-also --concurrent 0 --no-classes > gpurun_out/seq_bench.txt 2>&1 || exit 1
python3 tools/prof_sequence.py gpurun_out/seq_trace row_stats f_init 400 > gpurun_out/seq_fresh.txt
python3 tools/prof_sequence.py gpurun_out/seq_trace count_rows f_init 400 > gpurun_out/seq_fresh_all.txt
rm -rf gpurun_out/seq_trace
tail -1 gpurun_out/seq_fresh.txt
