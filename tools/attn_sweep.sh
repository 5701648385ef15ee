#!/bin/bash
# Attention tile / occupancy sweep at the config-5 shape (B = 32, T = 300, H = 8, head dim 128) and at config 2
# (B = 64, T = 196, head dim 64), both attention kinds.  Runs on the GPU box; output -> gpurun_out/attn_sweep.log
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
echo "## linear attention (tools/attn_time.py): workgroups of the chunk-walking kernels"
for w in 256 512 1024 2048; do
  echo "-- HIG_APPLY_WGS=$w HIG_APPLY_BWD_WGS=$w HIG_CTX_BWD_WGS=$w"
  HIG_APPLY_WGS=$w HIG_APPLY_BWD_WGS=$w HIG_CTX_BWD_WGS=$w python3 tools/attn_time.py 2>&1 | grep "B="
done
for c in 1 0; do
  echo "-- HIG_CTX_WALK=$c (1: one workgroup per (sample, head) walks the rows; 0: chunk-parallel + combine)"
  HIG_CTX_WALK=$c python3 tools/attn_time.py 2>&1 | grep "B="
done
echo "## full softmax attention (tools/fullattn_bench.py): waves per workgroup of the matrix-core kernels"
for nw in 2 4 8; do
  for b in 32 64; do
    echo "-- HIG_FULLATTN_WAVES=$nw B=$b"
    HIG_FULLATTN_WAVES=$nw python3 tools/fullattn_bench.py $b 300 2>&1 | grep "us"
  done
done
