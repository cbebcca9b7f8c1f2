#!/usr/bin/env python3
"""Where does a kernel use its highest vector registers?  tools/vgpr_peak.py file.s kernel_substring [threshold]
(file.s from `hipcc -S --cuda-device-only -gline-tables-only`): source lines (.loc) of the instructions that touch v[threshold..]."""
import collections
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 112
m = None
for mm in re.finditer(r'^(_Z[^\s:]*):\s*; @', s, re.M):
    if pat in mm.group(1):
        m = mm
        break
if not m:
    sys.exit("kernel not found")
end = s.index('.Lfunc_end', m.end())
files = {}
for fm in re.finditer(r'^\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s, re.M):
    files[int(fm.group(1))] = fm.group(3) or fm.group(2)
loc = None
hits = collections.Counter()
for l in s[m.end():end].split('\n'):
    l = l.strip()
    lm = re.match(r'\.loc\s+(\d+)\s+(\d+)', l)
    if lm:
        loc = (files.get(int(lm.group(1)), lm.group(1)).split('/')[-1], int(lm.group(2)))
        continue
    if not l or l.startswith(('.', ';')) or l.endswith(':'):
        continue
    regs = [int(x) for x in re.findall(r'\bv(\d+)\b', l)]
    for a, b in re.findall(r'v\[(\d+):(\d+)\]', l):
        regs += [int(a), int(b)]
    if regs and max(regs) >= thr:
        hits[loc] += 1
for (f, ln), c in sorted(((k, v) for k, v in hits.items() if k), key=lambda kv: (kv[0][0], kv[0][1])):
    print(f"{f}:{ln}  {c}")
