"""Per basic block of one kernel: instruction count, distinct VGPRs / AGPRs touched, spill traffic.
usage: bb_regs.py file.s mangled_kernel_name [min_instr]"""
import re, sys, collections
txt = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]; mn = int(sys.argv[3]) if len(sys.argv) > 3 else 60
start = next(i for i, l in enumerate(txt) if l.startswith(name + ':'))
end = next(i for i in range(start, len(txt)) if 's_endpgm' in txt[i])
blocks, cur, bn = [], [], 'entry'
for l in txt[start:end]:
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append((bn, cur)); bn = l.split(':')[0]; cur = []
    else:
        t = l.strip()
        if t and not t.startswith(';') and not t.startswith('.'): cur.append(t)
blocks.append((bn, cur))
def regs(b, pfx):
    r = set()
    for ins in b:
        for m in re.finditer(r'\b%s\[(\d+):(\d+)\]|\b%s(\d+)\b' % (pfx, pfx), ins):
            if m.group(1): r.update(range(int(m.group(1)), int(m.group(2)) + 1))
            else: r.add(int(m.group(3)))
    return r
for bn, b in blocks:
    if len(b) < mn: continue
    c = collections.Counter(i.split()[0] for i in b)
    sp = {k: v for k, v in c.items() if 'accvgpr' in k or 'scratch' in k or 'lane' in k or k.startswith('v_mov')}
    print(f"{bn:10s} n={len(b):4d} vgpr={len(regs(b,'v')):3d} agpr={len(regs(b,'a')):3d} {sp}")
