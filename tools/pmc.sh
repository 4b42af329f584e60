#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters...>" [more counter groups ...]  -> gpurun_out/pmc_<tag>/summary.txt
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -o g -- python3 $ROOT/tools/quick_bench.py 125000 > $OUT/g$i.log 2>&1
  echo "group $i rc=$?"
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as fh:
    for k, d in acc.items():
        if "cpep_kernel" not in k: continue
        for c, v in sorted(d.items()):
            fh.write(f"{k:48s} {c:28s} n={len(v)} mean={sum(v)/len(v):.6g}\n")
print(open("$OUT/summary.txt").read())
PY
