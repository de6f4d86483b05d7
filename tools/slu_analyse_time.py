"""Writes config 5's matrix (or grid(N)) for tools/slu_analyse_time and runs it: host time of the direct route's
analysis by phase.  python tools/slu_analyse_time.py [cfg5|grid] [N]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nodal_amd import generators as gen
from oracle import nodal_oracle as oracle
kind = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
table = gen.cfg5_table(N) if kind == "cfg5" else gen.grid_table(N)
G, _ = oracle.assemble_fast(table)
G = G.tocsr(); G.sort_indices()
path = "/tmp/slu_matrix.bin"
with open(path, "wb") as f:
    np.array([G.shape[0], G.nnz], dtype=np.int64).tofile(f)
    G.indptr.astype(np.int32).tofile(f); G.indices.astype(np.int32).tofile(f); G.data.astype(np.float64).tofile(f)
exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "slu_analyse_time")
subprocess.run([exe, path], check=True)
