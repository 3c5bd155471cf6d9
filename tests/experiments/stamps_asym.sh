#!/bin/bash
# per-workgroup cycle totals of kernel B2 with even / uneven tile shares (stamps build)
cd "$(dirname "$0")/../.."
for a in ${ASYMS:-0 4}; do
  echo "== R2L_B2_ASYM=$a"
  R2L_B2_ASYM=$a R2L_STAMPS_DETAIL=1 R2L_STAMPS_KEEP_LUMA=1 python tests/stamps.py 2>/dev/null | grep -v "^fwd\|fill-phase" | cut -c1-500
done
