#!/usr/bin/env python3
"""Static instruction mix per kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only).
usage: isa_count.py file.s [substring-of-kernel-name]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+|\w+):[^\n]*\n(.*?)s_endpgm', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if want not in name:
        continue
    ops = []
    for l in body.split('\n'):
        t = l.strip()
        if not l.startswith('\t') or not t or t[0] in ';.':
            continue
        ops.append(t.split()[0])
    c = collections.Counter(ops)
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    print(f"{name[:70]}: {len(ops)} instr, {valu} VALU, {c['s_nop']} s_nop")
    print("   ", ", ".join(f"{k} {v}" for k, v in c.most_common(10)))
