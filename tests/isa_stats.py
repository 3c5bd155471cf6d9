#!/usr/bin/env python3
"""Instruction mix of the kernels in a gfx950 assembly listing (hipcc --offload-device-only -S)."""
import collections
import re
import sys

path = sys.argv[1]
want = sys.argv[2:] or None
cur = None
stats = collections.OrderedDict()
meta = {}
for line in open(path):
    m = re.match(r'^(\w+):\s*; @', line)
    if m:
        cur = m.group(1)
        stats[cur] = collections.Counter()
        continue
    if cur is None:
        continue
    if line.startswith('\t.end_amdhsa_kernel') or line.startswith('.Lfunc_end'):
        pass
    m = re.match(r'^\t([a-z_0-9]+)', line)
    if m and not line.startswith('\t.'):
        stats[cur][re.sub(r'_(e32|e64|sdwa|dpp)$', '', m.group(1))] += 1
    m = re.match(r'^; (NumVgprs|NumAgprs|ScratchSize|Occupancy|SGPRBlocks|NumSgprs|LDSByteSize): (\d+)', line)
    if m:
        meta.setdefault(cur, {})[m.group(1)] = int(m.group(2))
for k, c in stats.items():
    if want and not any(w in k for w in want):
        continue
    tot = sum(c.values())
    if tot < 50:
        continue
    grp = collections.Counter()
    for op, n in c.items():
        if op.startswith('v_pk_'):
            grp[op] += n
        elif op.startswith('v_'):
            grp['valu_other'] += n
            if op in ('v_fma_f32', 'v_fmac_f32', 'v_mul_f32', 'v_add_f32', 'v_mov_b32', 'v_readlane_b32',
                      'v_writelane_b32', 'v_log_f32', 'v_exp_f32', 'v_accvgpr_write_b32', 'v_accvgpr_read_b32'):
                grp[op] += n
        elif op.startswith('s_'):
            grp['salu/smem'] += n
        elif op.startswith('ds_'):
            grp[op] += n
        elif op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')):
            grp[op] += n
    print(k, meta.get(k, {}))
    print('   total', tot, dict(sorted(grp.items(), key=lambda kv: -kv[1])))
