#!/usr/bin/env python3
"""Merge per-kernel PMC traffic files (tests/pmc_static.sh / tests/profile_round.sh output) into ONE profiles/*.json with a
`_meta` block that says which build they were collected on -- bench.py reads that tag instead of asserting one (VERDICT r5 #1a).

    python tests/tools/merge_pmc.py <out.json> <round> <in.json> [<in.json> ...] [--digest <hex>]

Run on the GPU box right after the counter passes: the digest is taken from the tree that just ran."""
import datetime
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def main(argv):
    digest = None
    if '--digest' in argv:
        i = argv.index('--digest')
        digest = argv[i + 1]
        argv = argv[:i] + argv[i + 2:]
    out, rnd, ins = argv[0], argv[1], argv[2:]
    if digest is None:
        from raw2logit_amd import _lib
        digest = _lib.source_digest()
    merged = {}
    for f in ins:
        with open(f) as fh:
            merged.update({k: v for k, v in json.load(fh).items() if not k.startswith('_')})
    merged['_meta'] = {'round': rnd, 'library_digest': digest,
                       'collected': datetime.datetime.utcnow().strftime('%Y-%m-%d %H:%MZ'),
                       'how': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950), '
                              'KiB -> bytes (MI355X_MICROARCH.md, HBM / rocprofv3 section)'}
    with open(out, 'w') as fh:
        json.dump(merged, fh, indent=1)
    print(out, sorted(k for k in merged if not k.startswith('_')), digest)


if __name__ == '__main__':
    main(sys.argv[1:])
