"""Percentiles of one kernel's durations from a rocprofv3 results.db: python tools/prof_pct.py DIR substring"""
import glob, sqlite3, sys
path, sub = sys.argv[1], sys.argv[2]
db = glob.glob(path + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else cols[0]
d = sorted(r[0] / 1e3 for r in c.execute(f"select end-start from kernels where {name} like ?", (f"%{sub}%",)))
if d:
    pick = lambda q: d[min(len(d) - 1, int(q * len(d)))]
    print(f"{sub}: n {len(d)} min {d[0]:.1f} p10 {pick(.1):.1f} p25 {pick(.25):.1f} p50 {pick(.5):.1f} p75 {pick(.75):.1f} p90 {pick(.9):.1f} max {d[-1]:.1f} us")
