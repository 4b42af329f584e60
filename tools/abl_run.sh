for v in suppc1 suppc2 suppc1 suppc2; do python tools/abl_supp.py $v 100000 131072 200000 37; done
