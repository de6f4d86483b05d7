"""Ordered kernel sequence of one solve from a rocprofv3 results.db: start offset, duration and the gap in
front of every kernel, between the last occurrence of a start marker and the next occurrence of an end marker.
    python tools/prof_sequence.py <dir-or-db> <start_substring> <end_substring> [max_lines]"""
import glob, sqlite3, sys

path, a, b = sys.argv[1], sys.argv[2], sys.argv[3]
limit = int(sys.argv[4]) if len(sys.argv) > 4 else 400
dbs = glob.glob(path + "/**/*.db", recursive=True) if not path.endswith(".db") else [path]
c = sqlite3.connect(dbs[0])
tables = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt = "kernels" if "kernels" in tables else [t for t in tables if "kernel" in t.lower()][0]
rows = list(c.execute(f"select name, start, end from {kt} order by start"))
starts = [i for i, r in enumerate(rows) if a in r[0]]
i0 = starts[-1]
# the last start marker that still has an end marker behind it
while not any(b in rows[i][0] for i in range(i0 + 1, len(rows))):
    starts.pop()
    i0 = starts[-1]
i1 = next(i for i in range(i0 + 1, len(rows)) if b in rows[i][0])
t0 = rows[i0][1]
prev_end = t0
tot_gap = 0
for r in rows[i0:i1 + 1][:limit]:
    nm = r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    gap = r[1] - prev_end
    tot_gap += max(gap, 0)
    print(f"{(r[1] - t0) / 1e3:9.1f} us  {nm:44s} {(r[2] - r[1]) / 1e3:7.1f} us  gap {gap / 1e3:6.1f}")
    prev_end = max(prev_end, r[2])
print(f"{i1 - i0 + 1} kernels, {(rows[i1][2] - t0) / 1e3:.1f} us, gaps {tot_gap / 1e3:.1f} us")
