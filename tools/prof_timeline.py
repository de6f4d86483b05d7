"""Timeline of one solve from a rocprofv3 results.db (rocpd sqlite): per-kernel durations, the
gaps between consecutive kernels, and the share of wall time the GPU sat idle between them.

    python tools/prof_timeline.py <dir-or-db> [first_kernel_substring] [occurrence] [end_kernel_substring]
"""
import glob, sqlite3, sys


def main():
    path = sys.argv[1]
    dbs = glob.glob(path + "/**/*.db", recursive=True) if not path.endswith(".db") else [path]
    c = sqlite3.connect(dbs[0])
    tables = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    kt = "kernels" if "kernels" in tables else [t for t in tables if "kernel" in t.lower()][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({kt})")]
    name = "name" if "name" in cols else cols[0]
    rows = list(c.execute(f"select {name}, start, end from {kt} order by start"))
    marker = sys.argv[2] if len(sys.argv) > 2 else "f_init"
    occ = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    starts = [i for i, r in enumerate(rows) if marker in r[0]]
    if not starts:
        print("marker kernel not found")
        return
    i0 = starts[occ]
    # the solve: from the marker to the next marker (or the end)
    if len(sys.argv) > 4:  # up to (not including) the first later kernel that matches the end marker
        nxt = [i for i in range(i0 + 1, len(rows)) if sys.argv[4] in rows[i][0]]
    else:
        nxt = [i for i in starts if i > i0]
    i1 = nxt[0] if nxt else len(rows)
    seg = rows[i0:i1]
    busy = sum(r[2] - r[1] for r in seg)
    wall = seg[-1][2] - seg[0][1]
    gaps = [seg[k + 1][1] - seg[k][2] for k in range(len(seg) - 1)]
    print(f"{len(seg)} kernels, wall {wall/1e3:.1f} us, busy {busy/1e3:.1f} us ({100*busy/wall:.1f} %), "
          f"gaps: mean {sum(gaps)/len(gaps)/1e3:.2f} us, >5us: {sum(1 for g in gaps if g > 5000)}, "
          f"sum of gaps > 5 us: {sum(g for g in gaps if g > 5000)/1e3:.1f} us")
    agg = {}
    for r in seg:
        nm = r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        a = agg.setdefault(nm, [0, 0])
        a[0] += 1
        a[1] += r[2] - r[1]
    for nm, (cnt, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"  {nm:50s} {cnt:5d} x {tot/cnt/1e3:7.2f} us = {tot/1e3:8.1f} us")


if __name__ == "__main__":
    main()
