#!/usr/bin/env python3
"""Instruction histogram of one kernel in a .s file (hipcc -S --cuda-device-only [-gline-tables-only]):
tools/isa_hist.py file.s kernel_substring [top_n] [--phases header.h]
--phases: static instruction counts per phase of k_frame_lds - the source lines between consecutive FR_STAMP(n) markers of the
header; an instruction belongs to the phase of the last line of that header its .loc chain went through (inlined helpers count
for their call site).  Needs line tables (-gline-tables-only)."""
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


if "--phases" in sys.argv:
    hdr = sys.argv[sys.argv.index("--phases") + 1]
    hname = hdr.split("/")[-1]
    marks = []
    for ln, line in enumerate(open(hdr), 1):
        mm = re.search(r"FR_STAMP\((\d+)\)", line)
        if mm and "define" not in line:
            marks.append((ln, int(mm.group(1))))
    files = {}
    for fm in re.finditer(r'^\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s, re.M):
        files[int(fm.group(1))] = (fm.group(3) or fm.group(2)).split("/")[-1]
    cur = None
    per = collections.defaultdict(collections.Counter)
    for l in body.split("\n"):
        l = l.strip()
        lm = re.match(r"\.loc\s+(\d+)\s+(\d+)", l)
        if lm:
            if files.get(int(lm.group(1))) == hname:
                cur = int(lm.group(2))
            continue
        if not l or l.startswith((".", ";", "/")) or l.endswith(":"):
            continue
        ph = "prologue"
        if cur is not None:
            prev = [m for m in marks if m[0] <= cur]
            if prev:
                ph = f"after FR_STAMP({prev[-1][1]}) (line {prev[-1][0]})"
        op = l.split()[0]
        cls = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "other"
        per[ph][cls] += 1
    print("static instructions per phase (valu / salu / lds / vmem):")
    for ph, c in per.items():
        print(f"  {ph}: {c['valu']} / {c['salu']} / {c['lds']} / {c['vmem']}")
