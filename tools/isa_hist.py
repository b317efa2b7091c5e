#!/usr/bin/env python3
"""tools/isa_hist.py file.s start end [top]: opcode histogram of a line range of a device assembly dump"""
import sys, re, collections
f, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]); top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
c = collections.Counter()
for l in open(f).read().split('\n')[a:b]:
    if l.startswith('\t') and not l.strip().startswith(('.', ';')):
        c[l.split()[0]] += 1
tot = sum(c.values())
grp = collections.Counter()
for k, v in c.items():
    g = 'f64' if re.match(r'v_\w+_f64', k) else 'v_other' if k.startswith('v_') else k.split('_')[0]
    grp[g] += v
print("total", tot, dict(grp))
for k, v in c.most_common(top): print("%6d %s" % (v, k))
