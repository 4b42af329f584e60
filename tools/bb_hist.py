"""Opcode histogram of the largest basic blocks of one kernel in an AMDGPU .s file.
usage: bb_hist.py file.s mangled_kernel_name [nblocks]"""
import re, sys, collections
txt = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]; nb = int(sys.argv[3]) if len(sys.argv) > 3 else 2
start = next(i for i, l in enumerate(txt) if l.startswith(name + ':'))
end = next(i for i in range(start, len(txt)) if 's_endpgm' in txt[i])
blocks, cur, bn = [], [], 'entry'
for l in txt[start:end]:
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append((bn, cur)); bn = l.split(':')[0]; cur = []
    else:
        t = l.strip()
        if t and not t.startswith(';') and not t.startswith('.'): cur.append(t.split()[0])
blocks.append((bn, cur))
print("total instr", sum(len(b[1]) for b in blocks))
for n, b in sorted(blocks, key=lambda b: -len(b[1]))[:nb]:
    c = collections.Counter(b)
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    print(n, len(b), "VALU", valu, dict(c.most_common(18)))
