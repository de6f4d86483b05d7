"""Summary of gpurun_out/prof_gepp (rocprofv3 --kernel-trace --stats of tools/dense_small_probe.py)."""
import csv, glob
f = glob.glob("gpurun_out/prof_gepp/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "%9.1f us avg" % (float(r["AverageNs"]) / 1e3),
          "%8.2f ms" % (float(r["TotalDurationNs"]) / 1e6))
t = glob.glob("gpurun_out/prof_gepp/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(t)) if "gepp_panel" in r["Kernel_Name"]]
print("gepp_panel by workgroup size:", " ".join(
    f"{r['Workgroup_Size_X']}:{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f}" for r in rows[:31:2]))
