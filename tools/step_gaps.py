"""Step anatomy from a rocprofv3 kernel trace (…_kernel_trace.csv): for the steady steps of the bench command, the
period between consecutive gradient launches, each kernel's duration and the idle gap in front of it.
usage: step_gaps.py trace.csv [gradient-kernel substring]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "true, 0>(cude::CpepArgs)"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
idx = idx[len(idx) // 2:]                       # the second half: steady clock, the timed steps
per, gaps, durs = [], defaultdict(list), defaultdict(list)
for a, b in zip(idx[:-1], idx[1:]):
    per.append(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]))
    for j in range(a + 1, b + 1):
        name = rows[j]["Kernel_Name"].split("(")[0][-40:]
        gaps[name].append(int(rows[j]["Start_Timestamp"]) - int(rows[j - 1]["End_Timestamp"]))
        durs[name].append(int(rows[j]["End_Timestamp"]) - int(rows[j]["Start_Timestamp"]))
med = lambda v: sorted(v)[len(v) // 2]
print(f"steps {len(per)}  period median {med(per) / 1e3:.1f} us")
for name in gaps:
    print(f"  {name:42s} gap before {med(gaps[name]) / 1e3:6.2f} us   duration {med(durs[name]) / 1e3:8.2f} us   (n = {len(gaps[name])})")
