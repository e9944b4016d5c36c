#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stdin)."""
import re, sys
cur = None
d = {}
for l in sys.stdin:
    if 'error' in l or 'warning' in l:
        print(l.rstrip())
    m = re.search(r'Function Name: (\S+)', l)
    if m:
        cur = m.group(1); d = {}; continue
    m = re.search(r':\s+([A-Za-z][A-Za-z ]*?)(?: \[[^\]]*\])?: (\d+) \[-Rpass', l)
    if m and cur:
        d[m.group(1).strip()] = m.group(2)
        if 'LDS Size' in m.group(1):
            g = re.search(r'GeomILi(\d+)ELi(\d+)EEELi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELb(\d)', cur)
            tag = 'N=%s P=%s WG=%s OCC=%s win=%s dma=%s dbuf=%s' % g.groups() if g else cur[:48]
            print(tag, {k: d.get(k) for k in ('VGPRs', 'AGPRs', 'ScratchSize', 'Occupancy', 'VGPRs Spill')})
