#!/usr/bin/env python3
"""tests/golden/grad_achieved_gpu.json from a GPU run's parity log (R2L_PARITY_LOG=... pytest -m gpu):

    python tests/tools/make_grad_baseline.py profiles/r06_parity_gpu.tsv

For every golden parametrized case and parameter: the error the shipped kernels ACHIEVED against the float64 oracle.
tests/parity_checks.py: check_param_case holds later builds to ACHIEVED_K x that (never looser than the old 1.5e-3-of-scale
limit, never tighter than what a correct float32 kernel reaches on well-conditioned frames) -- VERDICT r5 weak #1: the golden
check must notice a regression long before 1.5e-3 of the gradient's scale.  The file is DATA about this library's kernels, not
about the reference; the oracle and the reference's golden vectors stay the judges of correctness."""
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(tsv):
    out = {}
    for line in open(tsv):
        what, err, _tol = line.rstrip('\n').split('\t')
        m = re.match(r'param/(.+)/grad (\S+) vs float64 oracle$', what)
        if m:
            out[f'{m[1]}/{m[2]}'] = float(err)
    sys.path.insert(0, REPO)
    from raw2logit_amd import _lib
    doc = {'_meta': {'source': os.path.relpath(tsv, REPO), 'library_digest': _lib.source_digest(),
                     'what': 'max |grad - float64 oracle| achieved on the GPU by the fused kernels, per golden case and parameter'},
           'achieved': dict(sorted(out.items()))}
    path = os.path.join(REPO, 'tests', 'golden', 'grad_achieved_gpu.json')
    with open(path, 'w') as f:
        json.dump(doc, f, indent=1)
    print(f'{len(out)} entries -> {path}')


if __name__ == '__main__':
    main(sys.argv[1])
