#!/bin/bash
# kernel-time breakdown of forward + backward at config 2 (rocprofv3 --kernel-trace --stats)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fb -- python3 $R/tools/fwdbwd_time.py > /dev/null 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/fb/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print("%-86s %5s x %8.2f us = %5.2f %%" % (r["Name"].replace("(anonymous namespace)::", "")[:86], r["Calls"], float(r["AverageNs"]) / 1e3,
                                            100 * float(r["TotalDurationNs"]) / tot))
print("sum of kernel time per iteration: %.2f ms (13 iterations)" % (tot / 13 / 1e6))
PY
