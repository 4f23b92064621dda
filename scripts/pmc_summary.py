"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, mean of each counter
and mean dispatch duration (us)."""
import csv
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            name = re.split(r"\((?!anonymous)", name)[0]
            if name.startswith("at::") or name.startswith("__amd"):
                continue
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
            acc[name]["_dur_us_" + row["Counter_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for name, ctrs in sorted(acc.items()):
    durs = [sum(v) / len(v) for c, v in ctrs.items() if c.startswith("_dur_us_")]
    print(f"{name[:70]}  (avg dispatch {sum(durs)/len(durs):.1f} us)")
    for c, v in sorted(ctrs.items()):
        if not c.startswith("_dur"):
            print(f"    {c:32s} n={len(v):4d} mean={sum(v)/len(v):.5g}")
