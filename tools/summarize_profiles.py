#!/usr/bin/env python3
"""Condenses gpurun_out/<tag>/ (tools/collect_profiles.sh) into profiles/<tag>_*.{csv,json,md}."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    f = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    return f[-1] if f else None  # newest run (gpurun merges successive runs into one directory)


shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, tag + "_bench.json"))
shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, tag + "_bench_under_rocprof.json"))
ks = one("trace/**/*kernel_stats.csv")
shutil.copy(ks, os.path.join(dst, tag + "_kernel_stats.csv"))
# round 2: kernel stats of the bf16-storage sampling step and of the fp32 training step, counter summaries of the bf16 FFN
# GEMM and of the HBM-bound kernels (written by tools/collect_profiles.sh next to the headline's trace)
for pattern, name in (("ws16_stamps.txt", "_ws16_stamps.txt"), ("apply16_stamps.txt", "_apply16_stamps.txt"),
                      ("gemm16_shapes.txt", "_gemm16_shapes.txt"), ("train_step_eager_vs_captured.txt", "_train_step_eager_vs_captured.txt"),
                      ("trace_bf16/**/*kernel_stats.csv", "_kernel_stats_bf16_sampling_step.csv"),
                      ("trace_train/**/*kernel_stats.csv", "_kernel_stats_train_step.csv"),
                      ("pmc_hbm/summary.json", "_hbm_kernels_pmc.json"), ("pmc16_ffn1/summary.json", "_pmc16_ffn1.json")):
    f = one(pattern)
    if f:
        shutil.copy(f, os.path.join(dst, tag + name))
lines = ["# %s: rocprofv3 summaries (MI355X, one GPU)" % tag, "",
         "Command profiled: `python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra`", "",
         "## Kernel time (rocprofv3 --kernel-trace --stats), top 12", "",
         "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
for r in list(csv.DictReader(open(ks)))[:12]:
    n = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    lines.append("| `%s` | %s | %.2f | %.1f | %.2f |" % (n[:100], r["Calls"], int(r["TotalDurationNs"]) / 1e6,
                                                        float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
lines += ["", "## PMC passes for the FFN linear1 GEMM (M=12544,K=512,N=1024, cold caches), per launch", ""]
vals = {}
for name in ("pmc_fetch", "pmc_write", "pmc_sq"):
    f = one(name + "/**/*counter_collection.csv")
    if not f:
        continue
    by, dur = collections.defaultdict(list), {}
    for r in csv.DictReader(open(f)):
        if "gemm_f32_kernel" in r["Kernel_Name"]:
            by[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, v in by.items():
        vals[k] = sum(v) / len(v)
    if dur:
        vals["dur_us_" + name] = sum(dur.values()) / len(dur) / 1e3
for k, v in sorted(vals.items()):
    lines.append("* %s = %.6g" % (k, v))
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    traffic = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
    lines += ["", "HBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B = %.1f MB "
              "(gfx950: FETCH_SIZE counts wide coalesced reads at half, MI355X_MICROARCH.md HBM section); "
              "algorithmic bytes = 79.2 MB (X 25.7 + W 2.1 + out 51.4)." % (traffic / 1e6)]
    vals["traffic_bytes"] = traffic
if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
    util = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (vals["GRBM_GUI_ACTIVE"] / 8 * 1024)
    clk = vals["GRBM_GUI_ACTIVE"] / 8 / (vals["dur_us_pmc_sq"] * 1e-6) / 1e9
    lines += ["", "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs) = %.1f %%; "
              "clock under load = %.2f GHz" % (100 * util, clk)]
    vals["mfma_util"], vals["clock_ghz"] = util, clk
json.dump(vals, open(os.path.join(dst, tag + "_ffn_gemm_pmc.json"), "w"), indent=1)
open(os.path.join(dst, tag + "_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
