#!/bin/bash
# Copies what tools/profile_rNN.sh left under gpurun_out/prof_rNN (scratch) into profiles/rNN (tracked) and installs the
# PMC record bench.py reads.  usage: tools/collect_profiles.sh [r05]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
R=${1:-r05}
SRC=$ROOT/gpurun_out/prof_$R
DST=$ROOT/profiles/$R
mkdir -p $DST
cp $SRC/stats/stats_kernel_stats.csv $DST/kernel_stats.csv
cp $SRC/supp_stage_inputs_stats/stats_kernel_stats.csv $DST/kernel_stats_supp_stage_inputs.csv
[ -f $SRC/supp_steps_stats/stats_kernel_stats.csv ] && cp $SRC/supp_steps_stats/stats_kernel_stats.csv $DST/kernel_stats_supp_steps_only.csv
for extra in supp_fill train estep smallgrad; do
  [ -f $SRC/${extra}_stats/stats_kernel_stats.csv ] && cp $SRC/${extra}_stats/stats_kernel_stats.csv $DST/kernel_stats_$extra.csv
done
cp $SRC/adaptive_stats/stats_kernel_stats.csv $DST/kernel_stats_adaptive.csv
cp $SRC/summary.txt $DST/rocprof_summary.txt
cp $SRC/bench_stats.json $DST/bench_under_rocprof.json
cp $SRC/occupancy.txt $DST/occupancy.txt
grep -v "^\[\|^W2\|^E2\|^I2" $SRC/adaptive.log > $DST/adaptive.txt || true
cp $SRC/adaptive_supp_stats/stats_kernel_stats.csv $DST/kernel_stats_adaptive_supp.csv
grep -v "^\[\|^W2\|^E2\|^I2" $SRC/adaptive_supp.log > $DST/adaptive_supp.txt || true
cp $SRC/pmc_traffic.json $ROOT/profiles/pmc_traffic.json
grep -o "SQ_[A-Z_0-9]*" $SRC/counters_available.txt | sort -u | tr '\n' ' ' > $DST/sq_counters_available.txt || true
echo "collected into $DST"
