#!/bin/bash
# round 6, review item 5: what would sharing the halo rows of neighbouring bands be worth for the Malvar2004 / bilinear short chains?
# Upper bound: -DR2L_EXP_STATIC_NO_HALO fetches every halo row from inside the band (wrong results, timing only).  Same buffers, one
# process, interleaved (tests/static_ab_inproc.py).  Libraries built beforehand: tests/build_ab.sh nohalo "-DR2L_EXP_STATIC_NO_HALO -DR2L_STREAM_BF=1"
# bf1 "-DR2L_STREAM_BF=1" (bilinear takes the branch-free item only with R2L_STREAM_BF=1; Malvar2004 always does)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_malvar_nohalo.txt
{
echo "== Malvar2004 short chain, 256x1024x1024"; DEB=1 python3 tests/static_ab_inproc.py nohalo=ab/nohalo.so bf1=ab/bf1.so
echo "== bilinear short chain (branch-free item in both A/B builds), 256x1024x1024"; DEB=0 python3 tests/static_ab_inproc.py nohalo=ab/nohalo.so bf1=ab/bf1.so
echo "== Malvar2004, 1024x512x512"; DEB=1 B=1024 S=512 python3 tests/static_ab_inproc.py nohalo=ab/nohalo.so bf1=ab/bf1.so
} > $OUT 2>&1
cat $OUT
