#!/usr/bin/env python3
"""Where do the 4-6 us of a kilobyte-sized coarse-level kernel go?  Summarises rocprofv3 --pmc passes over
`python3 tools/sa_probe.py 1000 3 reuse` (config 3) into profiles/<round>_pmc_coarse_levels.json:
per (kernel, grid) the mean of every counter collected, plus derived figures --
  wait_share   = SQ_WAIT_ANY / SQ_WAVE_CYCLES       waves parked on s_waitcnt / barriers (memory latency)
  active_share = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  l2_hit       = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
  gui_active_us= GRBM_GUI_ACTIVE / 8 XCDs / clock    (guide: reads high on dispatches under 0.3 ms)

Usage: pmc_coarse.py OUT.json DIR [DIR ...]   (one DIR per pass)
"""
import collections, csv, glob, json, os, re, sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.split(r"\(", name)[0]


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for r in csv.DictReader(f):
                    a = acc[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]]
                    a[0] += 1
                    a[1] += float(r["Counter_Value"])
    want = re.compile(os.environ.get("PMC_KERNELS", r"^(k_|f_)"))  # PMC_KERNELS: other kernels (setup, stamping)
    res = {}
    for (name, grid), ctrs in sorted(acc.items(), key=lambda kv: (kv[0][0], kv[0][1])):
        if not want.search(name):
            continue
        m = {c: v[1] / v[0] for c, v in ctrs.items()}
        e = {"dispatches": max(v[0] for v in ctrs.values()), "grid_threads": grid, "counters_mean": m}
        wc = m.get("SQ_WAVE_CYCLES")
        if wc:
            for k, c in (("wait_share", "SQ_WAIT_ANY"), ("issue_stall_share", "SQ_WAIT_INST_ANY"),
                         ("active_share", "SQ_ACTIVE_INST_ANY")):
                if c in m:
                    e[k] = m[c] / wc
        if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_share"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
        if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m and m["TCC_HIT_sum"] + m["TCC_MISS_sum"] > 0:
            e["l2_hit"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
        res[f"{name}:{grid}"] = e
    with open(out, "w") as f:
        json.dump({"note": __doc__.strip().split("\n\n")[0].replace("\n", " "), "kernels": res}, f, indent=1)
    for k, e in res.items():
        print(k, {x: round(e[x], 3) for x in ("wait_share", "issue_stall_share", "active_share", "l2_hit", "lds_conflict_share") if x in e},
              "waves", e["counters_mean"].get("SQ_WAVES"))


if __name__ == "__main__":
    main()
