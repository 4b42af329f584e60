#!/bin/bash
# Profiles bench.py on the GPU box: kernel-trace stats + separate PMC passes (FETCH_SIZE, WRITE_SIZE).
# usage: tools/profile.sh <tag>     (outputs under gpurun_out/prof_<tag>/)
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_stats.json 2> $OUT/stats.err
echo "stats rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/fetch.err
echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/write.err
echo "write rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -o sq -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/sq.err
echo "sq rc=$?"
cd $ROOT
find $OUT -name "*.csv" | head -20
python3 $ROOT/tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
