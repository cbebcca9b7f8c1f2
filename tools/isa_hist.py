#!/usr/bin/env python3
"""Instruction histogram of one kernel in a --save-temps .s file: tools/isa_hist.py file.s kernel_substring"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
m = None
for mm in re.finditer(r'^(_Z[^\s:]*):\s*; @', s, re.M):
    if pat in mm.group(1):
        m = mm
        break
if not m:
    sys.exit("kernel not found")
end = s.index('.Lfunc_end', m.end())
body = s[m.end():end]
ins = []
for l in body.split('\n'):
    l = l.strip()
    if not l or l.startswith(('.', ';', '/')) or l.endswith(':'):
        continue
    ins.append(l.split()[0])
c = collections.Counter(ins)
print(m.group(1), len(ins), 'instructions; valu', sum(v for k, v in c.items() if k.startswith('v_')), 'salu', sum(v for k, v in c.items() if k.startswith('s_')),
      'lds', sum(v for k, v in c.items() if k.startswith('ds_')), 'vmem', sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'flat_'))))
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print(f"  {k} {v}")
