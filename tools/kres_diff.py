"""Occupancy / register changes between two builds.  usage: kres_diff.py old.s new.s"""
import re, sys
def load(p):
    txt = open(p).read(); out = {}
    for m in re.finditer(r"^(_Z\w+):.*?; NumVgprs: (\d+)\n; NumAgprs: (\d+)\n; TotalNumVgprs: (\d+)\n; ScratchSize: (\d+)\n.*?; Occupancy: (\d+)", txt, re.S | re.M):
        out[m.group(1)] = (int(m.group(4)), int(m.group(5)), int(m.group(6)))
    return out
a, b = load(sys.argv[1]), load(sys.argv[2])
for k in sorted(b):
    if k in a and (a[k][2] != b[k][2] or a[k][1] != b[k][1]):
        print(f"{k[:95]:95s} regs {a[k][0]:3d}->{b[k][0]:3d} scratch {a[k][1]}->{b[k][1]} occ {a[k][2]}->{b[k][2]}")
print(len(b), "kernels,", sum(1 for k in b if k in a and a[k][2] > b[k][2]), "lost occupancy,", sum(1 for k in b if k in a and a[k][2] < b[k][2]), "gained")
