#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr) per fused-kernel instance."""
import re
import sys

txt = open(sys.argv[1]).read()
show_all = len(sys.argv) > 2
for b in re.split(r'remark: Function Name: ', txt)[1:]:
    name = b.split('\n')[0].split(' ')[0]

    def g(k):
        m = re.search(k + r': (\d+)', b)
        return int(m.group(1)) if m else -1
    m = re.search(r'fusedILi(\d)ELi(\d)ELi(\d+)E(\w+?)Li(\d)ELi(\d)E', name)
    if not m:
        continue
    K, NQ, NT, ty, bl = m.groups()
    sp, scr, occ = g('VGPRs Spill'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]')
    if show_all or ty == 'ff' or sp > 0 or scr > 0:
        print('K%s NQ%s T%s %-10s blend%s: vgpr %3d agpr %3d sgpr %3d spill %d scratch %d occ %d' % (
            K, NQ, NT, ty, bl, g('VGPRs'), g('AGPRs'), g('SGPRs'), sp, scr, occ))
