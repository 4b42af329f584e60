for r in 1 2; do for v in cur pfg1 pfg3 magic; do python tools/abl_bench.py $v 125000; done; done
