for N in 1250 10000 40000; do for f in 1 0; do if [ $f = 1 ]; then export CUDE_NO_MH_FUSE=1; else unset CUDE_NO_MH_FUSE; fi; python tools/bench_estep.py $N 2>&1 | grep -v amdgpu | tail -1; done; done
