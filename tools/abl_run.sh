cd /tmp; export TMPDIR=/tmp
for L in 30 15 10 5 2; do export CUDE_CPEP_PATH=2:$L; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_fwdL$L -o fwd -- python3 $GRAFT_REPO_ROOT/tools/bench_fwd.py 640 > /dev/null 2>&1; echo "L=$L"; grep "fwd_kernel\|scan" $GRAFT_REPO_ROOT/gpurun_out/prof_fwdL$L/fwd_kernel_stats.csv | cut -d, -f1-4 | cut -c1-90; done
