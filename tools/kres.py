"""Per-kernel register / scratch / occupancy summary of an AMDGPU .s file.  usage: kres.py file.s [substring]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):.*?; NumVgprs: (\d+)\n; NumAgprs: (\d+)\n; TotalNumVgprs: (\d+)\n; ScratchSize: (\d+)\n.*?; Occupancy: (\d+)",
                     txt, re.S | re.M):
    name = m.group(1)
    if flt in name:
        sp = re.search(r"%s.*?sgpr_spill_count: (\d+)" % re.escape(".name:           " + name), txt, re.S)
        print(f"{name[:90]:90s} vgpr {m.group(2):>3s} agpr {m.group(3):>3s} total {m.group(4):>3s} scratch {m.group(5):>4s} occ {m.group(6)}")
