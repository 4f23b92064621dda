"""Dev: list the kernels of the LAST launch sequence in a rocprofv3 kernel_trace.csv, starting at the last kernel whose name
contains argv[2].  Usage: python scripts/trace_list.py <dir> <first-kernel-substring>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"])
t0, tot = int(rows[idx]["Start_Timestamp"]), 0.0
for r in rows[idx:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    n = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:64]
    print("%8.1f us  %7.1f us  wg %6s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d, r.get("Grid_Size_X", r.get("Grid_Size", "?")), n))
print("sum of kernel times %.1f us; span %.1f us" % (tot, (int(rows[-1]["End_Timestamp"]) - t0) / 1e3))
