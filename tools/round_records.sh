# Run ON THE GPU BOX: the round's auxiliary records (topologies, shapes, stubborn sources, config 4 at its full 1024
# members, the direct route, the pair sweep) -> gpurun_out/rec/*.txt; copy what is to be judged into profiles/.
mkdir -p gpurun_out/rec
O=gpurun_out/rec
timeout -k 10 300 python tools/topologies.py > $O/topologies.txt 2>&1; echo "topologies rc $?"
timeout -k 10 300 python tools/shape_probe.py grid:100 grid:316 grid:562 grid:1000 grid:1200 rgrid:1000 cfg5:700 cfg5:1000 batch:64x140 batch:1024x35 grid3:80 > $O/shapes.txt 2>&1; echo "shapes rc $?"
timeout -k 10 400 python tools/stubborn_probe.py 1000 > $O/stubborn_sources.txt 2>&1; echo "stubborn rc $?"
timeout -k 10 400 python tests/campaigns/cfg4_full.py > $O/cfg4_full_1024_on_one_gpu.txt 2>&1; echo "cfg4 rc $?"
timeout -k 10 300 python tools/direct_probe.py 40 100 316 1000 > $O/direct_route.txt 2>&1; echo "direct rc $?"
(timeout -k 10 120 python tools/pairs_probe.py 240 40; timeout -k 10 200 python tools/pairs_probe.py 1000 192) > $O/pair_sweep.txt 2>&1; echo "pairs rc $?"
