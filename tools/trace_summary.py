#!/usr/bin/env python3
"""Average duration of the bulk trailing-update launches in a rocprofv3 kernel trace
(gemm_sub_kernel<0> with at least MIN_TILES 128x128 tiles) -- the launches bench.py times with
HIP events for roofline.achieved.  Usage: trace_summary.py KERNEL_TRACE.csv [MIN_TILES]"""
import csv
import sys

path = sys.argv[1]
min_tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sel = []
for r in csv.DictReader(open(path)):
    if "gemm_sub_kernel<0>" in r["Kernel_Name"]:
        tiles = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        sel.append((tiles, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Stream_Id"]))
by_stream = {}
for tiles, us, st in sel:
    by_stream.setdefault(st, []).append((tiles, us))
for st, rows in sorted(by_stream.items()):
    rows = [x for x in rows if x[0] >= min_tiles]
    if rows:
        n = len(rows)
        flops = sum(2.0 * 256 * 128 * 128 * t for t, _ in rows)
        print(f"stream {st}: {n} launches of gemm_sub_kernel<0>, mean {sum(u for _, u in rows) / n:.1f} us, "
              f"{flops / sum(u for _, u in rows) / 1e6:.1f} TFLOP/s (tile-count flops)")
