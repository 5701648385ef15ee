#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (separate: TCC slots) over the HBM-bound kernels.  usage: tools/pmc_hbm.sh <tag>
set -u
TAG=${1:-pmc_hbm}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/tools/hbm_pmc.py > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/tools/hbm_pmc.py > $O/write.log 2>&1
python3 - <<PY
import csv, glob, json, collections
def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    return n[:70]
out = collections.OrderedDict()
for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = sorted(glob.glob("$O/%s/**/*counter_collection.csv" % name, recursive=True))[-1]
    t = sorted(glob.glob("$O/%s/**/*kernel_trace.csv" % name, recursive=True))[-1]
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(t)):
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr: acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if "elementwise" in k or "reduce_kernel" in k or "copyBuffer" in k: continue
        e = out.setdefault(short(k), {})
        e[ctr + "_KB"] = sum(v) / len(v)
        e["us_" + name] = sum(dur[k]) / len(dur[k]) / 1e3
for k, e in out.items():
    if "FETCH_SIZE_KB" in e and "WRITE_SIZE_KB" in e:
        e["hbm_MB"] = (2 * e["FETCH_SIZE_KB"] + e["WRITE_SIZE_KB"]) * 1024 / 1e6    # guide: FETCH_SIZE reads half on gfx950
        e["GB_per_s"] = e["hbm_MB"] / e["us_fetch"] * 1e3
        e["frac_of_8TB_s"] = e["GB_per_s"] / 8000.0
json.dump(out, open("$O/summary.json", "w"), indent=1)
for k, e in out.items(): print("%-72s %s" % (k, {a: round(b, 1) if a != "frac_of_8TB_s" else round(b, 3) for a, b in e.items()}))
PY
