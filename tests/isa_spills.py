#!/usr/bin/env python3
"""Where the SGPR spills of a kernel sit (v_writelane_b32 = a scalar parked in a vector lane, v_readlane_b32 = taken back) relative
to its row loop:  hipcc -O3 ... --offload-device-only -S r2l_api.hip -o r2l.s;  python tests/isa_spills.py r2l.s kernel ...
The row loop = the innermost large backward-branch region of the kernel (the 6-fold unrolled group of row steps).  For every kernel: counts
inside / outside the loop, per row step, and an excerpt of the loop's first spill accesses with their surroundings."""
import collections
import re
import sys

path, names = sys.argv[1], sys.argv[2:]
lines = open(path).read().split('\n')
for name in names:
    try:
        start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\d+' + name + r'\w*:', l))
    except StopIteration:
        print(f'{name}: not found')
        continue
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('\t.end_amdhsa_kernel') or lines[i].startswith('.Lfunc_end'))
    body = [l.split(';')[0].rstrip() for l in lines[start:end]
            if (l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')) or l.startswith('.LBB')]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.match(r'^\t(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)', l)
        if m and m.group(2) in labels and labels[m.group(2)] < i:
            loops.append((labels[m.group(2)], i))
    big = max(t[1] - t[0] for t in loops)
    a, b = min((t for t in loops if t[1] - t[0] >= 0.5 * big), key=lambda t: t[1] - t[0])   # the innermost large loop
    meta = {}
    for l in lines[start:end + 60]:
        m = re.match(r'^; (NumVgprs|ScratchSize|Occupancy|NumSgprs|sgpr_spill_count): (\d+)', l) or \
            re.match(r'^\s+\.(sgpr_spill_count|vgpr_spill_count):\s+(\d+)', l)
        if m:
            meta[m.group(1)] = int(m.group(2))
    spill = lambda l: ('v_readlane_b32' in l) or ('v_writelane_b32' in l)
    inside = [i for i in range(a, b + 1) if spill(body[i])]
    outside = [i for i in range(len(body)) if spill(body[i]) and not (a <= i <= b)]
    valu = sum(1 for l in body[a:b + 1] if l.startswith('\tv_'))
    c_in = collections.Counter(body[i].split()[0] for i in inside)
    c_out = collections.Counter(body[i].split()[0] for i in outside)
    print(f'== {name}  {meta}')
    print(f'   row loop: instructions {a}..{b} of {len(body)} ({b - a + 1} instructions, {valu} vector; 6 row steps)')
    print(f'   SGPR-spill accesses INSIDE the row loop: {dict(c_in)} = {len(inside) / 6:.1f} per row step '
          f'({100.0 * len(inside) / max(valu, 1):.1f} % of its vector instructions); OUTSIDE (item set-up, reductions): {dict(c_out)}')
    for i in inside[:3]:
        print('   ...')
        for j in range(max(a, i - 3), min(b, i + 4)):
            print(f'   {j:5d} {body[j].strip()}')
    print()
