for v in suppacc suppa1 suppa2 suppa1; do python tools/abl_supp.py $v 100000 131072 37; done
