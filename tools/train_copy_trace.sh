#!/bin/bash
# Host <-> device copies of cude_train_restarts as a function of its iteration counts (rocprofv3 --memory-copy-trace;
# the trace carries no sizes, so the evidence is the COUNT: copies that do not grow with the iterations are the one-time
# upload / download of the restarts).  usage: tools/train_copy_trace.sh [N=100000] [K=25]  -> gpurun_out/r05/train_copy_trace.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-100000}; K=${2:-25}
OUT=$ROOT/gpurun_out/r05/copytrace
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for cfg in "10 0" "20 0" "0 5" "0 10" "10 0 2" "20 0 2" "0 5 1" "0 10 1"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --memory-copy-trace --output-format csv -d $OUT/$tag -o t -- python3 $ROOT/tools/train_once.py $N $K $cfg > $OUT/$tag.log 2>&1
  echo "$cfg rc=$?"
done
cd $ROOT
python3 - $OUT $N $K > $ROOT/gpurun_out/r05/train_copy_trace.txt <<'PY'
import csv, glob, os, sys
out, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
print(f"cude_train_restarts, {K} restarts x {N} subjects (CPEP3 2x6x6x1): host<->device copies of the whole process")
print("(population upload, one forward solve for the synthetic observations, the call itself) by rocprofv3 --memory-copy-trace")
print(f"{'adam':>5} {'lbfgs':>6} {'optimiser state':>16} {'H2D copies':>11} {'H2D ms':>8} {'D2H copies':>11} {'D2H ms':>8}")
for tag in ["10_0", "20_0", "0_5", "0_10", "10_0_2", "20_0_2", "0_5_1", "0_10_1"]:
    f = glob.glob(os.path.join(out, tag, "**", "*memory_copy_trace.csv"), recursive=True)
    if not f:
        print(tag, "no trace"); continue
    h2d = d2h = 0; th = td = 0.0
    for r in csv.DictReader(open(f[0])):
        dt = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
        if "HOST_TO_DEVICE" in r["Direction"]: h2d += 1; th += dt
        elif "DEVICE_TO_HOST" in r["Direction"]: d2h += 1; td += dt
    p = tag.split("_")
    where = "device" if len(p) == 2 else ("host (train_host=%s)" % p[2])
    print(f"{p[0]:>5} {p[1]:>6} {where:>16} {h2d:>11} {th:>8.2f} {d2h:>11} {td:>8.2f}")
PY
cat $ROOT/gpurun_out/r05/train_copy_trace.txt
