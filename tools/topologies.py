#!/usr/bin/env python3
"""Sparse passive path on topologies other than the benchmark grids: iterations, time and the
scaled residual for ladders, trees, grids with dangling wires,
high-contrast grids.  Run on the GPU box:

    python tools/topologies.py [name ...]          # NODAL_LOWDEG=0 switches the elimination off
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nodal_amd import _ffi  # noqa: E402
from nodal_amd import generators as gen  # noqa: E402

if os.environ.get("NODAL_LIB"):  # a library built with other compile-time constants (experiments)
    _ffi.LIB_PATH = os.path.abspath(os.environ["NODAL_LIB"])


def contrast_grid(side, decades, seed=1):
    rng = np.random.default_rng(seed)
    vals = 10.0 ** rng.uniform(-decades / 2, decades / 2, gen.grid_resistor_count(side))
    return gen.grid_table(side, vals)


def anisotropic_grid(side, ratio):
    """vertical resistors `ratio` times the horizontal ones"""
    ga, gb, _ = gen._grid_arrays(side)
    return gen.grid_table(side, np.where(gb == ga + 1, 1.0, float(ratio)))


def general_contrast(side, decades, seed=1):
    """config 5 (grid + 1 % voltage / dependent sources) with the resistances spread over decades"""
    rng = np.random.default_rng(seed)
    table = gen.cfg5_table(side)
    res = table.type == 0
    table.value[res] = 10.0 ** rng.uniform(-decades / 2, decades / 2, int(res.sum()))
    return table


CASES = {
    "ladder5e3": lambda: gen.ladder_table(5000),
    "ladder2e4": lambda: gen.ladder_table(20000),
    "ladder1e5": lambda: gen.ladder_table(100000),
    "ladder1e6": lambda: gen.ladder_table(1000000),
    "chain1e5": lambda: gen.chain_table(100000),
    "tree1e5": lambda: gen.binary_tree_table(100000),
    "tree1e6": lambda: gen.binary_tree_table(1000000),
    "wires300x200": lambda: gen.grid_with_wires_table(300, 200),
    "grid300": lambda: gen.grid_table(300),
    "grid1000": lambda: gen.grid_table(1000),
    "contrast300d1": lambda: contrast_grid(300, 1),
    "contrast300d2": lambda: contrast_grid(300, 2),
    "contrast300d3": lambda: contrast_grid(300, 3),
    "contrast300d4": lambda: contrast_grid(300, 4),
    "contrast300d6": lambda: contrast_grid(300, 6),
    "aniso300r10": lambda: anisotropic_grid(300, 10),
    "aniso300r1000": lambda: anisotropic_grid(300, 1000),
    "contrast300d8": lambda: contrast_grid(300, 8),
    "contrast300d10": lambda: contrast_grid(300, 10),
    "contrast300d12": lambda: contrast_grid(300, 12),
    "contrast1000d4": lambda: contrast_grid(1000, 4),
    "contrast1000d6": lambda: contrast_grid(1000, 6),
    "contrast3000d4": lambda: contrast_grid(3000, 4),
    "general300d0": lambda: gen.cfg5_table(300),
    "general300d4": lambda: general_contrast(300, 4),
}


def run(name):
    table = CASES[name]()
    h = _ffi.Handle(0)
    h.upload(table)
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    # the first solve builds whatever depends on the topology only (elimination sets, reduced
    # patterns); repeated solves on the same handle reuse it
    times = []
    for _ in range(4):
        t0 = time.perf_counter()
        x, info, iters, relres = h.solve_sparse()
        times.append((time.perf_counter() - t0) * 1e3)
    first, best = times[0], min(times[1:])
    res = h.residual()  # (distances to SuperLU are the tests' business: tests/test_gpu_kernels.py)
    print(f"{name:16s} n={len(x):8d} info={info} iters={iters:5d} first {first:7.2f} ms, then {best:7.2f} ms  residual {res:.1e}",
          flush=True)
    h.close()


if __name__ == "__main__":
    names = sys.argv[1:] or [c for c in CASES if c not in ("grid1000", "contrast300d8", "contrast300d10", "contrast300d12", "contrast1000d4", "contrast1000d6", "contrast3000d4", "general300d4", "general300d0")]
    for nm in names:
        run(nm)
