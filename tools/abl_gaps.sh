# step anatomy (tools/step_gaps.py) of library variants: usage abl_gaps.sh variant...
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  cd $R && python tools/abl_step.py $v
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gaps_$v -o t -- python3 $R/tools/abl_step.py $v > /dev/null 2>&1
  cd $R && python tools/step_gaps.py gpurun_out/gaps_$v/t_kernel_trace.csv
done
