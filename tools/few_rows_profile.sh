#!/bin/bash
# per-kernel times of the few-row GEMMs (time / text embedding MLP, stylization emb_layers) with and without the split
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  export HIG_FEW_ROWS_SPLIT=$v
  rm -rf /tmp/fr$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fr$v -- python3 $R/tools/fwd_time.py > /dev/null 2>&1
  echo "split=$v"
  python3 - <<PY
import csv, glob
f = glob.glob("/tmp/fr$v/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if ("gemm_f32_kernel" in n and (", 3, false" in n or "false, false, 0, false, 0," in n)) or "reduce_slabs" in n or "add_bias_res" in n:
        print("  %-100s calls %5s avg %8.2f us" % (n[:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
