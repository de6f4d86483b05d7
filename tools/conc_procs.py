"""Throughput with solves in flight from SEVERAL PROCESSES (each with its own HIP runtime and hardware
queues) next to several streams of one process (tools/conc_probe.py):
python tools/conc_procs.py cfg3 PROCS STREAMS_PER_PROC [reuse]"""
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(name, streams, per_stream, reuse, barrier, out):
    import threading
    from nodal_amd import _ffi
    from nodal_amd import generators as gen
    table = {"cfg3": lambda: gen.grid_table(1000), "cfg5": lambda: gen.cfg5_table(1000)}[name]()
    handles = []
    for _ in range(streams):
        h = _ffi.Handle(0)
        h.upload(table)
        assert h.run(False) == 0
        handles.append(h)

    def work(h):
        for _ in range(per_stream):
            assert h.run(False, 0, reuse) == 0

    threads = [threading.Thread(target=work, args=(h,)) for h in handles]
    barrier.wait()
    t0 = time.time()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    out.put((t0, time.time()))
    for h in handles:
        h.close()


if __name__ == "__main__":
    name = sys.argv[1]
    procs, streams = int(sys.argv[2]), int(sys.argv[3])
    reuse = len(sys.argv) > 4 and sys.argv[4] == "reuse"
    per_stream = 16
    ctx = mp.get_context("spawn")
    barrier, out = ctx.Barrier(procs), ctx.Queue()
    ps = [ctx.Process(target=child, args=(name, streams, per_stream, reuse, barrier, out)) for _ in range(procs)]
    for p in ps:
        p.start()
    spans = [out.get() for _ in ps]
    for p in ps:
        p.join()
    elapsed = max(e for _, e in spans) - min(s for s, _ in spans)
    n = procs * streams * per_stream
    print(f"{name} procs {procs} x streams {streams} reuse {reuse} queues {os.environ.get('GPU_MAX_HW_QUEUES', 'default')}: "
          f"{n / elapsed:.1f} circuits/s", flush=True)
