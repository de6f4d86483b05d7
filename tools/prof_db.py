"""Kernel statistics from a rocprofv3 results.db (rocpd sqlite): name, calls, total / avg / min / max ns."""
import glob, sqlite3, sys

def main():
    path = sys.argv[1]
    dbs = glob.glob(path + "/**/*.db", recursive=True) if not path.endswith(".db") else [path]
    c = sqlite3.connect(dbs[0])
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else cols[0]
    q = (f"select {name}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
         f"from kernels group by {name} order by 3 desc")
    rows = list(c.execute(q))
    total = sum(r[2] for r in rows)
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    print(f"{'kernel':60s} {'calls':>7s} {'total_us':>10s} {'avg_us':>8s} {'min_us':>8s} {'max_us':>8s} {'%':>6s}")
    for r in rows[:top]:
        nm = r[0].replace("(anonymous namespace)::", "")
        nm = nm[5:] if nm.startswith("void ") else nm
        nm = nm.split("(")[0][:60]
        print(f"{nm:60s} {r[1]:7d} {r[2]/1e3:10.1f} {r[3]/1e3:8.2f} {r[4]/1e3:8.2f} {r[5]/1e3:8.2f} {100*r[2]/total:6.2f}")
    print(f"total kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")

if __name__ == "__main__":
    main()
