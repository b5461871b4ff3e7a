#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stderr saved to a file): one line per kernel.
usage: python tools/resource_usage.py remarks.txt [substring]"""
import re, sys
txt = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
NUM = r": (\d+)"
KEYS = [("VGPR", r"VGPRs"), ("SGPR", r"SGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"), ("occ", r"Occupancy \[waves/SIMD\]"),
        ("LDS", r"LDS Size \[bytes/block\]")]
for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split('\n')[0].split(' ')[0]
    if sub not in name:
        continue
    short = re.sub(r'^_ZN5swmhd12_GLOBAL__N_1\d+', '', name)
    short = re.sub(r'EEvNS_.*$', '', short)
    num = lambda pat: re.search(pat + NUM, b).group(1)
    vals = " ".join("%s %5s" % (k, num(pat)) for k, pat in KEYS)
    print(f"{short:48s} {vals}")
