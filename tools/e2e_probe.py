"""Times of the reference-style calls at 1e6 nodes on the GPU box (development aid): Netlist(path),
Circuit(netlist, sparse=True), .solve(), and a profile of the constructor."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import nodal_amd as n
from nodal_amd import generators as gen
import tempfile
path = os.path.join(tempfile.mkdtemp(), "g.csv")
gen.write_csv(gen.grid_rows(1000), path)
for rep in range(3):  # (the first one also pages the file and the tokenizer library in)
    t0 = time.perf_counter(); nl = n.Netlist(path); t1 = time.perf_counter()
    print(f"Netlist(path): {t1 - t0:.3f} s (host threads: {os.cpu_count()})")
for r in range(3):
    t1 = time.perf_counter(); c = n.Circuit(nl, sparse=True); t2 = time.perf_counter(); s = c.solve(); t3 = time.perf_counter()
    x = s.result; t4 = time.perf_counter()
    print(f"run {r}: Circuit() {1e3 * (t2 - t1):.1f} ms, solve() {1e3 * (t3 - t2):.1f} ms, .result {1e3 * (t4 - t3):.1f} ms, x0 {x[0]:.6f}")
t0 = time.perf_counter(); text = str(s); t1 = time.perf_counter()
print(f"str(solution): {t1 - t0:.2f} s ({text.count(chr(10))} lines; builds nodenum on first use)")
import cProfile, pstats
cProfile.run("n.Circuit(nl, sparse=True)", "/tmp/c.prof")
pstats.Stats("/tmp/c.prof").sort_stats("cumulative").print_stats(16)
