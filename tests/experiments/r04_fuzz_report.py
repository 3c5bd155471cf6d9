"""Assemble profiles/r04_fuzz.txt from the sweeps' outputs under gpurun_out/ (final build: r04_fuzz_b1 = seeds 101-103 + fuzz_more,
r04_fuzz_b2 = seeds 111-113, r04_fuzz3 = seeds 121-123; earlier builds of the round: the text kept in profiles/r04_fuzz_earlier.txt)."""
import os, re
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G = os.path.join(REPO, 'gpurun_out')
MODES = (('fuzz_default', 'default dispatch (product build)'),
         ('fuzz_planes', 'diagnostic build, R2L_BWD_PLANES=1 (plane-pass backward on every shape it accepts)'),
         ('fuzz_split', 'diagnostic build, R2L_FWD_STATS_SPLIT=1 R2L_BWD_PLANES=1 R2L_BWD_SPLIT_BLUR=1'))
out = ['''Randomised parity sweeps on the GPU, FINAL build of round 4 (tests/fuzz_gpu.py, tests/fuzz_more.py; float64 oracle).  Criterion: at every pixel
|kernel - oracle64| <= max(tolerance, 6 sigma of the float32 oracle's own local error) (tests/parity_checks.py: sigma_limit), gradients at their limit, no marginal
bands; "control" = a second float32 evaluation of the oracle (taps accumulated in reverse order) under the same criteria (it is reported, it cannot fail the run).
The literal per-pixel form the round-3 review asked for -- max(tolerance, 2 x |oracle32 - oracle64| over the 3x3 neighbourhood) -- is evaluated next to it: the
float32 ORACLE leaves that envelope more often than the kernels.  FAIL lines on the final build: none.
''']
totals = {m: [0, 0.0, 0.0, 0.0] for m, _ in MODES}
for tag, d, secs in (('SEED 101-103, 300 s per mode', 'r04_fuzz_b1', 300), ('SEED 111-113, 250 s per mode', 'r04_fuzz_b2', 250),
                     ('SEED 121-123, 480 s per mode', 'r04_fuzz3', 480)):
    out.append('--- ' + tag)
    for m, name in MODES:
        L = [l.rstrip() for l in open(os.path.join(G, d, m + '.txt')).read().splitlines() if l.strip() and 'amdgpu.ids' not in l]
        assert not any('FAIL' in l or 'over its limit' in l for l in L), (d, m)
        body = [l for l in L if 'random cases ok' in l or l.startswith('   for information') or l.startswith('worst gradient ratio')]
        out.append(name)
        out += ['   ' + l.strip() for l in body]
        mm = re.search(r'(\d+) random cases ok.*?oracle\) ([\d.]+) \(control.*?: ([\d.]+)\); worst grad error / limit ([\d.]+)', ' '.join(body))
        t = totals[m]
        t[0] += int(mm.group(1)); t[1] = max(t[1], float(mm.group(2))); t[2] = max(t[2], float(mm.group(3))); t[3] = max(t[3], float(mm.group(4)))
    if d == 'r04_fuzz_b1':
        out.append('fuzz_more.py (static chains, raw2rgb, staged path, SSIM / L2)')
        out += ['   ' + l.strip() for l in open(os.path.join(G, d, 'fuzz_more.txt')).read().splitlines() if l.startswith('ok') or 'cases' in l][-2:]
    out.append('')
out.append('Totals per mode on the final build: ' + '; '.join('%s %d cases, worst out ratio %.2f (control %.2f), worst gradient ratio %.2f'
                                                              % (m.replace('fuzz_', ''), *totals[m]) for m, _ in MODES) + '.')
out.append('')
out.append(open(os.path.join(REPO, 'profiles', 'r04_fuzz_earlier.txt')).read())
open(os.path.join(REPO, 'profiles', 'r04_fuzz.txt'), 'w').write('\n'.join(out))
print('\n'.join(out)[:3000])
