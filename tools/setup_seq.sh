cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -q -x -k "sparse or sagg or grid or reuse or symbolic or batch" > gpurun_out/t_sp.txt 2>&1; tail -3 gpurun_out/t_sp.txt
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/seq_trace -- python3 bench.py --workload cfg3 --steps 1 --warmup 1 --per-step 2 --no-cpu --no-also --concurrent 0 --no-classes > gpurun_out/seq_bench.txt 2>&1
python3 tools/prof_sequence.py gpurun_out/seq_trace row_stats f_init 400 > gpurun_out/seq_fresh.txt
rm -rf gpurun_out/seq_trace
tail -1 gpurun_out/seq_fresh.txt
timeout -k 10 300 python bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu --no-also --concurrent 0 --no-classes > gpurun_out/b3.txt 2>&1; python3 -c "
import json;d=json.loads(open('gpurun_out/b3.txt').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_solve'],d['phase_ms'],d['solver'])"
