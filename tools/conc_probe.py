"""Throughput with several solves in flight (one handle / stream / host thread each), fresh and with the
symbolic phases kept: python tools/conc_probe.py cfg3 2 4 6 8"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
for streams in [int(v) for v in sys.argv[2:]] or [4]:
    for reuse in (False, True):
        r = bench.concurrent_throughput(name, 0, streams, 12, reuse)
        print(f"{name} streams {streams} reuse {reuse}: {r['circuits_per_sec']:.1f} circuits/s, "
              f"{r['ms_per_circuit']:.2f} ms per circuit, latency {r['ms_latency_per_solve']:.2f} ms", flush=True)
