#!/usr/bin/env python3
"""Compare the instruction streams of the kernels in two `-save-temps` assembly files (hipcc ... -save-temps): per kernel, the
number of instructions, a hash of the stream with labels / symbol names normalised, and the first differing lines.
usage: isa_diff.py old.s new.s [substring of the kernel names to look at]"""
import hashlib, re, sys

def kernels(path):
    out, cur, name = {}, None, None
    for line in open(path):
        t = line.strip()
        m = re.match(r'^(_Z\w+):', t)
        if m and cur is None:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if t.startswith('.Lfunc_end') or t.startswith('.end_amdhsa_kernel'):
                out[name], cur = cur, None
                continue
            if not t or t[0] in '.;' and not t.startswith('.LBB'):
                continue
            t = t.split(';')[0].strip()
            t = re.sub(r'\.LBB\d+_\d+', 'L', t)
            t = re.sub(r'_Z\w+', 'SYM', t)
            if t and not t.endswith(':'):
                cur.append(t)
    return out

def key(name):  # template arguments of k_decode_persistent<INT8, M, DH, UK, TP, GROUP, ...>
    m = re.search(r'k_decode_persistentI((?:L[bi]\d+E)+)', name)
    return tuple(re.findall(r'L[bi](\d+)E', m.group(1))[:6]) if m else name

a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
flt = sys.argv[3] if len(sys.argv) > 3 else 'k_decode_persistent'
ka = {key(n): v for n, v in a.items() if flt in n}
kb = {key(n): v for n, v in b.items() if flt in n}
for k in sorted(set(ka) | set(kb), key=str):
    va, vb = ka.get(k), kb.get(k)
    if va is None or vb is None:
        print(k, 'only in', 'old' if vb is None else 'new')
        continue
    ha, hb = (hashlib.md5('\n'.join(v).encode()).hexdigest()[:8] for v in (va, vb))
    nd = sum(1 for x, y in zip(va, vb) if x != y) + abs(len(va) - len(vb))
    print(k, len(va), len(vb), 'SAME' if ha == hb else f'DIFF ({nd} lines differ positionally)')
    if ha != hb and len(va) == len(vb):
        shown = 0
        for x, y in zip(va, vb):
            if x != y and shown < 4:
                print('   -', x, '\n   +', y)
                shown += 1
