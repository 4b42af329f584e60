#!/bin/bash
# Round-5 profiles on the GPU box: rocprofv3 kernel-trace stats of the bench command, separate PMC passes (FETCH_SIZE /
# WRITE_SIZE / SQ counters: --pmc is never combined with trace domains other than --kernel-trace), the same for the
# suppression gradient kernel in its two checkpoint modes.  usage: tools/profile_r05.sh   -> gpurun_out/prof_r05/
set -u
# (rocprofv3 -L lists the counters this GPU exposes: kept beside the profiles)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r05
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $BENCH > $OUT/bench_stats.json 2> $OUT/stats.err
echo "stats rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $BENCH > /dev/null 2> $OUT/fetch.err
echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $BENCH > /dev/null 2> $OUT/write.err
echo "write rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -o sq -- $BENCH > /dev/null 2> $OUT/sq.err
echo "sq rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o sq2 -- $BENCH > /dev/null 2> $OUT/sq2.err
echo "sq2 rc=$?"
# (the wait breakdown of the headline kernel was measured in round 4: profiles/r04/experiments.md 4; not repeated)
SUPP="python3 $ROOT/tools/bench_supp.py 100000 --no-cpu"
for mode in stage_inputs; do
  if [ $mode = steps ]; then export CUDE_SUPP_CKPT=steps; else unset CUDE_SUPP_CKPT; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/supp_${mode}_stats -o stats -- $SUPP > $OUT/supp_${mode}.log 2> $OUT/supp_${mode}.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/supp_${mode}_fetch -o fetch -- $SUPP > /dev/null 2>> $OUT/supp_${mode}.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/supp_${mode}_write -o write -- $SUPP > /dev/null 2>> $OUT/supp_${mode}.err
  echo "supp $mode rc=$?"
done
unset CUDE_SUPP_CKPT
# the same kernel at one full fill of the chip (131 072 subjects = 2 048 waves)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/supp_fill_stats -o stats -- python3 $ROOT/tools/bench_supp.py 131072 --no-cpu > $OUT/supp_fill.log 2> $OUT/supp_fill.err
echo "supp fill rc=$?"
# SQ counters of the suppression gradient kernel (instructions per wave, busy share) and its granted occupancy
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/supp_sq -o sq -- $SUPP > /dev/null 2> $OUT/supp_sq.err
echo "supp sq rc=$?"
python3 $ROOT/tools/occupancy.py > $OUT/occupancy.txt 2>&1
cat $OUT/occupancy.txt
# adaptive mode: forward and gradient launches of the reference's c-peptide instance at 1e5 subjects
ADAPT="python3 $ROOT/tools/bench_adaptive.py 100000 4"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/adaptive_stats -o stats -- $ADAPT > $OUT/adaptive.log 2> $OUT/adaptive.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/adaptive_fetch -o fetch -- $ADAPT > /dev/null 2>> $OUT/adaptive.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/adaptive_write -o write -- $ADAPT > /dev/null 2>> $OUT/adaptive.err
echo "adaptive rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/adaptive_sq -o sq -- $ADAPT > /dev/null 2>> $OUT/adaptive.err
echo "adaptive sq rc=$?"
# ... and of the suppression model (the reference's EnsembleThreads solve, suppression_model.jl:113,123)
ADAPTS="python3 $ROOT/tools/bench_adaptive_supp.py 100000"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/adaptive_supp_stats -o stats -- $ADAPTS > $OUT/adaptive_supp.log 2> $OUT/adaptive_supp.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/adaptive_supp_fetch -o fetch -- $ADAPTS > /dev/null 2>> $OUT/adaptive_supp.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/adaptive_supp_write -o write -- $ADAPTS > /dev/null 2>> $OUT/adaptive_supp.err
echo "adaptive supp rc=$?"
# round 5: the restart trainer at 1e5 x 25, the speculative E-step at one GPU's share of configs[4], the small-population gradient
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_stats -o stats -- python3 $ROOT/tools/train_once.py 100000 25 10 5 > $OUT/train.log 2> $OUT/train.err
echo "train rc=$?"
CUDE_SPEC_DEPTHS=-1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/estep_stats -o stats -- python3 $ROOT/tools/bench_estep_spec.py 100 1250 10000 > $OUT/estep.log 2> $OUT/estep.err
echo "estep rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/smallgrad_stats -o stats -- python3 $ROOT/tools/bench_lossgrad.py 10000 > $OUT/smallgrad.log 2> $OUT/smallgrad.err
echo "small gradient rc=$?"
cd $ROOT
python3 $ROOT/tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
