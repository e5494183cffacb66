"""Instruction mix of the loops of one kernel in a hipcc -S listing:  python tools/isa_loops.py file.s <mangled-name regex>"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().splitlines()
pat = re.compile(sys.argv[2])
start = next(i for i, l in enumerate(lines) if l.endswith(':') is False and re.match(r'^_Z\S*:', l) and pat.search(l))
end = next(j for j in range(start, len(lines)) if lines[j].strip().startswith('s_endpgm'))
body = lines[start:end]
labels = {}
for k, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l.strip())
    if m:
        labels[m.group(1)] = k


def kind(op):
    return ('mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else
            'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global', 'buffer', 'flat')) else 'other')


for k, l in enumerate(body):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < k:
        a = labels[m.group(1)]
        c = Counter()
        for x in body[a:k + 1]:
            x = x.strip()
            if not x or x[0] in '.;' or x.endswith(':'):
                continue
            c[x.split()[0]] += 1
        cat = Counter()
        for op, n in c.items():
            cat[kind(op)] += n
        if cat['mfma']:
            print(f"loop {m.group(1)} lines {a}-{k}: {dict(cat)}")
            print("   ", sorted(((n, o) for o, n in c.items() if kind(o) in ('valu', 'salu')), reverse=True)[:18])
