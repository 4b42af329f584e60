for n in 125000 10000 64; do for a in 2,6,2 2,4,2 2,8,2; do for v in pf2d pf2e; do python tools/abl_bench.py $v $n $a; done; done; done
