cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sgm in 1 2 4; do
NODAL_SA_GSEG=$sgm timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gal$sgm -o gal -- python3 bench.py --workload cfg3 --steps 3 --warmup 1 --no-cpu --no-also --concurrent 0 > gpurun_out/gal.log 2>&1
echo "segments $sgm:"; grep -i "galerkin" gpurun_out/gal$sgm/gal_kernel_stats.csv | sed 's/.*)",//'
rm -rf gpurun_out/gal$sgm
done
