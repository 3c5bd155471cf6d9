#!/usr/bin/env python3
"""Register / scratch / LDS table of every kernel in a built libr2l_isp.so (code-object metadata, `llvm-readelf --notes`
of the gfx950 image inside the fat binary):  python tests/kernel_resources.py [lib.so] [name filter ...]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_table(lib):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, 'fat.bin'), os.path.join(d, 'dev.co')
        subprocess.run([f'{LLVM}/llvm-objcopy', f'--dump-section=.hip_fatbin={fat}', lib, os.path.join(d, 'x.so')], check=True)
        subprocess.run([f'{LLVM}/clang-offload-bundler', '--unbundle', '--type=o', f'--input={fat}',
                        '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--output={co}'], check=True)
        notes = subprocess.run([f'{LLVM}/llvm-readelf', '--notes', co], check=True, capture_output=True, text=True).stdout
    rows, cur = [], {}
    for line in notes.splitlines():
        m = re.match(r'\s+\.(\w+):\s+(\S+)', line)
        if not m:
            continue
        k, v = m.groups()
        if k == 'group_segment_fixed_size' and cur.get('name'):
            rows.append(cur)
            cur = {}
        if k in ('name', 'vgpr_count', 'agpr_count', 'sgpr_count', 'vgpr_spill_count', 'sgpr_spill_count',
                 'private_segment_fixed_size', 'group_segment_fixed_size', 'max_flat_workgroup_size'):
            cur[k] = v
    if cur.get('name'):
        rows.append(cur)
    return rows


def main():
    lib = os.path.join(REPO, 'raw2logit_amd', 'libr2l_isp.so')
    args = sys.argv[1:]
    if args and args[0].endswith('.so'):
        lib = args.pop(0)
    print(f'# {os.path.relpath(lib, REPO)}')
    print(f'{"kernel":58s} {"vgpr":>5s} {"sgpr":>5s} {"vspill":>6s} {"sspill":>6s} {"scratch":>7s} {"lds":>7s} {"wg":>5s}')
    for r in kernel_table(lib):
        name = re.sub(r'^_Z\d+', '', r['name'])
        name = re.sub(r'_kernel.*$', '', name)
        if args and not any(a in name for a in args):
            continue
        print(f'{name:58s} {r.get("vgpr_count", "?"):>5s} {r.get("sgpr_count", "?"):>5s} {r.get("vgpr_spill_count", "?"):>6s} '
              f'{r.get("sgpr_spill_count", "?"):>6s} {r.get("private_segment_fixed_size", "?"):>7s} '
              f'{r.get("group_segment_fixed_size", "?"):>7s} {r.get("max_flat_workgroup_size", "?"):>5s}')


if __name__ == '__main__':
    main()
