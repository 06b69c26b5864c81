#!/usr/bin/env python3
"""VGPR footprint of the inner loops of an AMDGPU .s file: distinct VGPRs referenced between an inner-loop header and its back branch."""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
def regs(a, b):
    used = set()
    for l in lines[a:b]:
        t = l.split(';')[0]
        for m in re.finditer(r'\bv\[(\d+):(\d+)\]', t):
            used.update(range(int(m.group(1)), int(m.group(2)) + 1))
        for m in re.finditer(r'\bv(\d+)\b', t):
            used.add(int(m.group(1)))
    return used
for i, l in enumerate(lines):
    if 'This Inner Loop Header: Depth=2' in l:
        # the label is on the previous line
        k = i - 1
        while k > 0 and not re.match(r'^\.LBB\d+_\d+:', lines[k]):
            k -= 1
        name = lines[k].split(':')[0]
        for j in range(i + 1, len(lines)):
            if re.search(r's_cbranch_\w+ ' + re.escape(name) + r'\b', lines[j]):
                break
        seg = lines[k:j]
        print(name, 'lines', k, j, 'len', j - k, 'vgprs', len(regs(k, j)), 'mfma', sum('v_mfma' in x for x in seg), 'exp',
              sum('v_exp' in x for x in seg), 'vmem', sum(('buffer_load' in x or 'global_load' in x) for x in seg), 'scratch',
              sum('scratch_' in x for x in seg))
