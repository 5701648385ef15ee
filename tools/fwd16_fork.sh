#!/bin/bash
# same-call A/B of the bf16-storage forward's B-row fork (HIG_FWD16_FORK: bit 0 embedding chain, bit 1 text side)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for f in 0 1 2 3 0 3; do
  echo "HIG_FWD16_FORK=$f"
  HIG_FWD16_FORK=$f python3 $R/tools/fwd16_time.py 64 2>&1 | grep -v amdgpu.ids | grep "ms/forward"
  HIG_FWD16_FORK=$f python3 $R/tools/fwd16_time.py 32 2>&1 | grep -v amdgpu.ids | grep "ms/forward"
  HIG_FWD16_FORK=$f python3 $R/tools/fwd_cfg5_time.py 2>&1 | grep "config-5"
done
