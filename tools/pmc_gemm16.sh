#!/bin/bash
# rocprofv3 counter passes for one bf16-storage GEMM shape (separate passes: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2).
# usage: tools/pmc_gemm16.sh <tag> [shape] [B]     results: gpurun_out/<tag>/summary.json
set -u
TAG=${1:-pmc16}; SHAPE=${2:-ffn1}; BB=${3:-64}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/$1 -- python3 $R/tools/gemm16_pmc.py $SHAPE $BB > $O/$1.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
run lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
python3 - <<PY
import csv, glob, json, collections
out = {"shape": "$SHAPE", "B": $BB}
for name in ("fetch", "write", "sq", "lds", "tcc"):
    fs = sorted(glob.glob("$O/%s/**/*counter_collection.csv" % name, recursive=True))
    if not fs: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        if any(k in r["Kernel_Name"] for k in ("gemm_bf16_kernel", "gemm_ws16_kernel", "gemm_wsp16_kernel")):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): out[k] = sum(v) / len(v)
    ts = sorted(glob.glob("$O/%s/**/*kernel_trace.csv" % name, recursive=True))
    if ts:
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(ts[-1])) if any(k in r["Kernel_Name"] for k in ("gemm_bf16_kernel", "gemm_ws16_kernel", "gemm_wsp16_kernel"))]
        if d: out["dur_us_" + name] = sum(d) / len(d) / 1e3
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    out["traffic_bytes"] = (2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024   # guide: FETCH_SIZE reads half on gfx950
if "SQ_VALU_MFMA_BUSY_CYCLES" in out and "GRBM_GUI_ACTIVE" in out:
    out["mfma_util"] = out["SQ_VALU_MFMA_BUSY_CYCLES"] / (out["GRBM_GUI_ACTIVE"] / 8 * 256 * 4)
    out["clock_ghz"] = out["GRBM_GUI_ACTIVE"] / 8 / (out["dur_us_sq"] * 1e3)
json.dump(out, open("$O/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
