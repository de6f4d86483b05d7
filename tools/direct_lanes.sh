# Run ON THE GPU BOX: wide fronts of a level on 1 / 2 / 3 / 4 / 6 streams (csrc/sparse_direct.hip, NODAL_DIRECT_LANES)
for L in 1 2 3 4 6; do
NODAL_DIRECT_LANES=$L NODAL_TRACE=1 timeout -k 10 300 python -c "
import sys; sys.argv=['x']
sys.path.insert(0,'tools')
import direct_probe as d
from nodal_amd import generators as gen
d.run('cfg5(1000) lanes $L', gen.cfg5_table(1000), ref=False)
" 2>&1 | grep -E "first|kept"
done
