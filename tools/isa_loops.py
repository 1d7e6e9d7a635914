#!/usr/bin/env python3
"""Instruction mix of the innermost loops of selected kernels in a hipcc -S listing.
usage: isa_loops.py file.s name-prefix [...]"""
import re
import sys

txt = open(sys.argv[1]).read()
funcs = re.split(r'\n(?=_Z\w+:)', txt)
for f in funcs:
    name = f.split(':', 1)[0]
    if not any(name.startswith(p) for p in sys.argv[2:]):
        continue
    lines = f.split('\n')
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(lines):
        m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            t = m.group(1)
            body = [x.strip() for x in lines[labels[t]:i + 1]]
            body = [x for x in body if x and not x.startswith(('.', ';'))]
            cnt = lambda *p: sum(1 for x in body if x.startswith(p))
            loops.append((len(body), cnt('v_'), cnt('s_'), cnt('ds_'), cnt('global_', 'buffer_', 'flat_', 'scratch_'),
                          cnt('s_waitcnt'), cnt('s_nop'), t, labels[t], i))
    print(name)
    print("   (total, valu, salu, lds, vmem, waitcnt, nop, label, first, last)")
    for L in sorted(loops, key=lambda x: -x[0])[:40]:
        print("  ", L)
