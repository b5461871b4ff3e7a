#!/usr/bin/env python3
"""Instruction mix of the longest loop of a kernel in an AMDGPU assembly file (hipcc -S --cuda-device-only).
usage: python tools/loop_mix.py file.s <kernel-name-substring> [top]"""
import re, sys, collections
txt = open(sys.argv[1]).read()
sub = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 14
for m in re.finditer(r'^(\S*' + re.escape(sub) + r'\S*):[^\n]*\n', txt, flags=re.M):
    name = m.group(1)
    a = m.end(); b = txt.index('.Lfunc_end', a)
    body = txt[a:b].split('\n')
    labels = {mm.group(1): i for i, l in enumerate(body) for mm in [re.match(r'^(\.LBB\d+_\d+):', l)] if mm}
    best = None
    for i, l in enumerate(body):
        mm = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            span = i - labels[mm.group(1)]
            if best is None or span > best[0]:
                best = (span, labels[mm.group(1)], i)
    if not best:
        continue
    cnt = collections.Counter(l.split()[0] for l in body[best[1]:best[2] + 1] if l.strip() and not l.strip().startswith(('.', ';')))
    valu = sum(v for k, v in cnt.items() if k.startswith('v_'))
    short = re.sub(r'^_ZN5swmhd12_GLOBAL__N_1\d+', '', name); short = re.sub(r'EEvNS_.*$', '', short)
    print(f"{short}: loop VALU {valu}, all {sum(cnt.values())}; " + ", ".join(f"{k} {v}" for k, v in cnt.most_common(top)))
