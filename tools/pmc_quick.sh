#!/bin/bash
# FETCH / WRITE PMC passes for the FFN linear1 GEMM only (tuning aid): tools/pmc_quick.sh <tag>
set -u
TAG=${1:-pmcq}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/tools/ffn_gemm_pmc.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/tools/ffn_gemm_pmc.py > $O/pmc_write.log 2>&1
python3 - <<PY
import csv, glob, collections
for name in ("pmc_fetch", "pmc_write"):
    f = sorted(glob.glob("$O/%s/**/*counter_collection.csv" % name, recursive=True))[-1]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gemm_f32_kernel" in r["Kernel_Name"] or "gemm_wsp32_kernel" in r["Kernel_Name"]]
    print(name, "launches", len(v), "mean", sum(v) / len(v), "min", min(v), "max", max(v))
PY
