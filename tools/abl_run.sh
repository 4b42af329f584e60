for r in 1 2; do for v in pf2d pf2f; do python tools/abl_bench.py $v 125000; done; done
for v in pf2d pf2f; do python tools/abl_bench.py $v 1000000; done
export CUDE_CPEP_PATH=1
for v in pf2d pf2f; do python tools/abl_bench.py $v 64; done
