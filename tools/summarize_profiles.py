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
                      ("pmc_hbm/summary.json", "_hbm_kernels_pmc.json"), ("pmc16_ffn1/summary.json", "_pmc16_ffn1.json"),
                      ("trace_train16/**/*kernel_stats.csv", "_kernel_stats_train_step_bf16s.csv"), ("train16_time.txt", "_train16_time.txt"),
                      ("train16_kernels_solo.txt", "_train16_kernels_solo.txt"), ("rccl_world1_probe.json", "_rccl_world1_probe.json"),
                      ("ws16_store_policy.txt", "_ws16_store_policy.txt"), ("fwd32_fold_textfork.txt", "_fwd32_fold_textfork.txt"),
                      ("cfg5_time.txt", "_cfg5_time.txt"),
                      # round 5 / 6: probes, stamps and per-shape timings behind the specialised-wave kernels
                      ("l2_fetch_probe.txt", "_l2_fetch_probe.txt"), ("coissue_probe.txt", "_coissue_probe.txt"),
                      ("lds_read_probe.txt", "_lds_read_probe.txt"), ("wsp16_stamps.txt", "_wsp16_stamps.txt"),
                      ("wgrad_stamps.txt", "_wgrad_stamps.txt"), ("wgrad_time.txt", "_wgrad_time.txt"),
                      ("two_person16_time.txt", "_two_person16_time.txt"), ("gemm32_shapes.txt", "_gemm32_shapes.txt"),
                      ("wsp32_stamps.txt", "_wsp32_stamps.txt"), ("coissue32_probe.txt", "_coissue32_probe.txt"),
                      ("mfma32_stream_probe.txt", "_mfma32_stream_probe.txt"), ("rowkernels_time.txt", "_rowkernels_time.txt"),
                      ("capture_soak.json", "_capture_soak.json"), ("overlap_forms.txt", "_train_step_fork_forms.txt")):
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
# ---- the roofline microbenchmark's own launches, cut out of the kernel TRACE by the two hig_marker_kernel marks bench.py puts
# around it (the same template also runs inside the forward, on two streams and at half the rows: the per-name statistics
# above mix the two) ----
kt = one("trace/**/*kernel_trace.csv")
if kt:
    rows = list(csv.DictReader(open(kt)))
    marks = sorted((int(r["Start_Timestamp"]), int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0)) for r in rows
                   if "hig_marker_kernel" in r["Kernel_Name"])
    m1 = [t for t, g in marks if g == 1]
    m2 = [t for t, g in marks if g == 2]
    if m1 and m2:
        lo, hi = m1[-1], m2[-1]
        durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows
                if ("gemm_f32_kernel" in r["Kernel_Name"] or "gemm_wsp32_kernel" in r["Kernel_Name"]) and lo < int(r["Start_Timestamp"]) < hi]
        if durs:
            avg = sum(durs) / len(durs) / 1e3
            flops = 2.0 * 12544 * 512 * 1024
            info = {"avg_us": round(avg, 2), "launches": len(durs), "tflops": round(flops / (avg * 1e-6) / 1e12, 1),
                    "frac_of_157.3": round(flops / (avg * 1e-6) / 1e12 / 157.3, 4),
                    "what": "FFN linear1 launches (M=12544, K=512, N=1024, bias+GELU) between hig_marker_kernel 1 and 2 of the "
                            "profiled bench.py run: the roofline microbenchmark alone"}
            json.dump(info, open(os.path.join(dst, tag + "_ffn_gemm_trace.json"), "w"), indent=1)
            lines += ["", "## FFN linear1 launches of the roofline microbenchmark (kernel trace between the two markers)", "",
                      "%d launches, average %.2f us -> %.1f TFLOP/s = %.3f of the fp32 MFMA peak (`roofline.frac` of the "
                      "bench line is computed from HIP events over the same launches)" %
                      (len(durs), avg, info["tflops"], info["frac_of_157.3"])]
    # per (kernel, grid) statistics: the forward's half-batch launches and the full-batch ones differ in their grids
    by_grid = collections.defaultdict(list)
    for r in rows:
        g = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
        by_grid[(r["Kernel_Name"], g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(os.path.join(dst, tag + "_kernel_by_grid.csv"), "w") as f:
        f.write("kernel,grid_x,calls,total_ms,avg_us\n")
        for (name, g), v in sorted(by_grid.items(), key=lambda kv: -sum(kv[1]))[:60]:
            n = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
            f.write('"%s",%s,%d,%.3f,%.2f\n' % (n[:160], g, len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3))
lines += ["", "## PMC passes for the FFN linear1 GEMM (M=12544,K=512,N=1024, cold caches), per launch", ""]
vals = {}
for name in ("pmc_fetch", "pmc_write", "pmc_sq"):
    f = one(name + "/**/*counter_collection.csv")
    if not f:
        continue
    by, dur = collections.defaultdict(list), {}
    for r in csv.DictReader(open(f)):
        if "gemm_f32_kernel" in r["Kernel_Name"] or "gemm_wsp32_kernel" in r["Kernel_Name"]:
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

# ---- round 6: the same kernel in its steady regime (tools/ffn_gemm_pmc.py warm: 400 launches back to back, the last 100) -------
def last100(path, want_counters):
    by, dur = collections.defaultdict(list), collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if "gemm_wsp32_kernel" not in r["Kernel_Name"] and "gemm_f32_kernel" not in r["Kernel_Name"]:
            continue
        key = r.get("Dispatch_Id", str(len(dur)))
        dur[key] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if want_counters:
            by[r["Counter_Name"]].append(float(r["Counter_Value"]))
    d = list(dur.values())[-100:]
    return {k: sum(v[-100:]) / len(v[-100:]) for k, v in by.items()}, (sum(d) / len(d) / 1e3 if d else None)


fw, kw = one("pmc_sq_warm/**/*counter_collection.csv"), one("kt_warm/**/*kernel_trace.csv")
if fw and kw:
    sq, dur_pmc = last100(fw, True)
    _, dur_kt = last100(kw, False)
    busy = sq.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024          # per SIMD
    steady = {"what": "400 launches back to back over 8 operand sets (HBM-resident activations), the LAST 100 summarised (a fresh "
                      "process finds the chip idle: its first ~100 launches run 15 % slower)",
              "sq": sq, "dur_us_under_pmc": round(dur_pmc, 2), "dur_us_kernel_trace_only": round(dur_kt, 2),
              "mfma_busy_cycles_per_simd": busy, "mfma_busy_check": "49 tiles x 128 MFMAs x 32 cycles = 200704"}
    if "GRBM_GUI_ACTIVE" in sq:
        steady["mfma_util_grbm"] = sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (sq["GRBM_GUI_ACTIVE"] / 8 * 1024)
        steady["grbm_clock_ghz_over_trace_duration"] = sq["GRBM_GUI_ACTIVE"] / 8 / (dur_kt * 1e-6) / 1e9
        steady["grbm_note"] = ("a quotient above the 2.4 GHz maximum means the counter's window is wider than the ~0.1 ms dispatch "
                               "(MI355X_MICROARCH.md: the quotient reads high below ~0.3 ms): MfmaUtil by GRBM under-reads here")
    steady["mfma_util_lower_bound_at_2p4GHz"] = busy / (dur_kt * 1e-6 * 2.4e9)
    steady["mfma_util_at_2p2GHz"] = busy / (dur_kt * 1e-6 * 2.2e9)
    vals["steady_state"] = steady
    lines += ["", "Steady state (last 100 of 400 back-to-back launches): %.1f us by kernel trace, MFMA busy %.0f cycles per SIMD -> "
              "MfmaUtil >= %.3f (at the 2.4 GHz maximum clock), %.3f at 2.2 GHz; by GRBM_GUI_ACTIVE %.3f (window wider than the dispatch)" %
              (dur_kt, busy, steady["mfma_util_lower_bound_at_2p4GHz"], steady["mfma_util_at_2p2GHz"], steady.get("mfma_util_grbm", float("nan")))]
vals["kernel"] = "FFN linear1 GEMM (M=12544, K=512, N=1024, bias+GELU), tools/ffn_gemm_pmc.py, separate rocprofv3 --pmc passes"
vals["algorithmic_bytes"] = 79.2e6
json.dump(vals, open(os.path.join(dst, tag + "_ffn_gemm_pmc.json"), "w"), indent=1)
open(os.path.join(dst, tag + "_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
