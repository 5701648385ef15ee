#!/bin/bash
# Same-box A/B of two builds of libhig.so (boxes differ by a few per cent from call to call: only numbers from ONE call compare).
# usage (on the GPU box): tools/ab_lib.sh <other .so, relative to the package dir> <rounds> <command...>
# Runs the command alternately with the other library swapped in ("A") and with the tree's own build ("B").
P=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}/human-interaction-generation_amd
other=$1; rounds=$2; shift 2
cp $P/libhig.so /tmp/libhig_B.so
for r in $(seq $rounds); do
  cp $P/$other $P/libhig.so; echo "A: $("$@" 2>&1 | grep -v amdgpu.ids | tail -1)"
  cp /tmp/libhig_B.so $P/libhig.so; echo "B: $("$@" 2>&1 | grep -v amdgpu.ids | tail -1)"
done
