#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stdin or file)."""
import re, subprocess, sys
txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
rows, cur = [], None
keys = {'V': r'VGPRs: (\d+)', 'A': r'AGPRs: (\d+)', 'S': r'SGPRs: (\d+)', 'scr': r'ScratchSize \[bytes/lane\]: (\d+)',
        'occ': r'Occupancy \[waves/SIMD\]: (\d+)', 'lds': r'LDS Size \[bytes/block\]: (\d+)'}
for line in txt.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = {'name': m.group(1)}; rows.append(cur)
    for k, pat in keys.items():
        m = re.search(pat, line)
        if m and cur is not None: cur[k] = m.group(1)
for r in rows:
    name = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip()
    name = name.replace('abg::', '').replace('(anonymous namespace)::', '')[:80]
    print("%-82s V=%s S=%s scratch=%s occ=%s lds=%s" % (name, r.get('V'), r.get('S'), r.get('scr'), r.get('occ'), r.get('lds')))
