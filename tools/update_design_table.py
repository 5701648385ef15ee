#!/usr/bin/env python3
"""Rewrites the measurement table of DESIGN.md section 5 from profiles/<tag>_bench.json + <tag>_ffn_gemm_pmc.json."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
d = json.load(open(os.path.join(ROOT, "profiles", tag + "_bench.json")))
e = d["extra"]
a, b = s.index("| quantity | value |"), s.index("Vendor reference point")
old = s[a:b]


def sub(pattern, repl):
    global old
    new, n = re.subn(pattern, lambda m: repl, old, count=1, flags=re.S)
    assert n == 1, pattern
    old = new


sub(r"\| denoiser forward, fp32 \(headline\) \|[^\n]*",
    "| denoiser forward, fp32 (headline) | **%.2f ms → %.2f M frames/s** (%.0f TFLOP/s algorithmic = %.0f %% of the fp32 MFMA peak) |"
    % (d["ms_per_step"], d["value"] / 1e6, d["fwd_tflops"], 100 * d["fwd_tflops"] / 157.3))
sub(r"\| FFN linear1 GEMM \(dominant kernel class\), fp32 \| [0-9.]+ ms, \*\*[0-9.]+ TFLOP/s = [0-9]+ % of 157.3\*\*",
    "| FFN linear1 GEMM (dominant kernel class), fp32 | %.3f ms, **%.1f TFLOP/s = %.0f %% of 157.3**"
    % (d["roofline"]["avg_launch_ms"], d["roofline"]["achieved"], 100 * d["roofline"]["frac"]))
if d["roofline"].get("traffic"):
    tr = d["roofline"]["traffic"] / 1e6
    sub(r"HBM traffic \*\*[0-9.]+ MB/launch vs 79.2 MB algorithmic\*\*", "HBM traffic **%.1f MB/launch vs 79.2 MB algorithmic**" % tr)
    sub(r"which is the remaining [0-9.]+ MB", "which is the remaining %.1f MB" % (tr - 79.2))
sub(r"\| same forward, split-bf16 products[^\n]*",
    '| same forward, split-bf16 products (`precision="bf16x3"`, opt-in) | %.2f ms → %.2f M frames/s, rel-L2 %.1e vs the fp32 path |'
    % (e["fwd_bf16x3"]["ms_per_step"], e["fwd_bf16x3"]["frames_per_s"] / 1e6, e["fwd_bf16x3"]["rel_l2_vs_f32_path"]))
sub(r"\| same forward, bf16 products[^\n]*",
    '| same forward, bf16 products (`precision="bf16"`, configs 3/5 arithmetic) | %.2f ms → %.2f M frames/s, rel-L2 %.1e |'
    % (e["fwd_bf16"]["ms_per_step"], e["fwd_bf16"]["frames_per_s"] / 1e6, e["fwd_bf16"]["rel_l2_vs_f32_path"]))
sub(r"\| train step, denoiser core[^\n]*",
    "| train step, denoiser core (q_sample+fwd+loss+bwd+clip+Adam), fp32 / bf16x3 | %.1f ms / %.1f ms → %.0f k / %.0f k frames/s |"
    % (e["train_step_f32"]["ms_per_step"], e["train_step_bf16x3"]["ms_per_step"], e["train_step_f32"]["frames_per_s"] / 1e3,
       e["train_step_bf16x3"]["frames_per_s"] / 1e3))
sub(r"\| train step, every trainable parameter[^\n]*",
    "| train step, every trainable parameter (text head inside; better of eager launches and hipGraph replay), fp32 | %.1f ms → %.0f k frames/s |"
    % (e["train_step_full_f32"]["ms_per_step"], e["train_step_full_f32"]["frames_per_s"] / 1e3))
sub(r"\| DDPM sampling B=32[^\n]*",
    "| DDPM sampling B=32, hipGraph replay, fp32 / bf16 | %.2f / %.2f ms per step → %.1f / %.1f samples/s per 1000 steps |"
    % (e["ddpm_sampling_f32"]["ms_per_denoise_step"], e["ddpm_sampling_bf16"]["ms_per_denoise_step"],
       e["ddpm_sampling_f32"]["samples_per_s_1000_steps"], e["ddpm_sampling_bf16"]["samples_per_s_1000_steps"]))
sub(r"\| two-person denoiser[^\n]*",
    "| two-person denoiser (32 pairs × 91 tokens × 263 features), fp32 | forward %.2f ms, forward+backward %.1f ms; captured PIT training step (16 pairs run twice) %.1f ms |"
    % (e["two_person"]["fwd_ms"], e["two_person"]["fwd_bwd_ms"], e["two_person"]["pit_train_step_ms"]))
sub(r"\| config-5 shape[^\n]*",
    "| config-5 shape (B=32, T=300, d=1024, L=12, hd=128) forward, fp32 / bf16 products | %.1f ms (%.0f TFLOP/s) / %.1f ms |"
    % (e["config5_long_sequence"]["fwd_ms_f32"], e["config5_long_sequence"]["fwd_tflops_f32"], e["config5_long_sequence"]["fwd_ms_bf16"]))
sub(r"\| text head fwd\+bwd[^\n]*",
    "| text head fwd+bwd at B=64 (77 tokens, d=256, ff=2048), HIP vs stock PyTorch-ROCm ops | %.2f ms vs %.2f ms |"
    % (e["text_head"]["fwd_bwd_ms_hip"], e["text_head"]["fwd_bwd_ms_stock_torch"]))
sub(r"\| CPU oracle on the GPU box[^\n]*",
    "| CPU oracle on the GPU box's host (16 threads = best of a doubling probe) | %.1f k frames/s → GPU/CPU = %.0f× (fp32 headline) |"
    % (d["cpu_baseline"]["value"] / 1e3, e["speedup_vs_cpu"]))
if "config2_variants" in e:
    v = e["config2_variants"]
    rowv = ("| config-2 variants (SURVEY 8d), fp32 forward | ragged lengths U{40..196}: %.2f ms; F = 263: %.2f ms (%.0f TFLOP/s); "
            "ff = 4·d = 2048: %.2f ms (%.0f TFLOP/s) |\n"
            % (v["lengths_U40_T"]["fwd_ms"], v["F263"]["fwd_ms"], v["F263"]["fwd_tflops"], v["ff2048"]["fwd_ms"],
               v["ff2048"]["fwd_tflops"]))
    if "| config-2 variants" in old:
        old = re.sub(r"\| config-2 variants[^\n]*\n", lambda m: rowv, old)
    else:
        old = old.replace("| text head fwd+bwd", rowv + "| text head fwd+bwd", 1)
if "evaluator" in e:
    v = e["evaluator"]
    rowe = ("| evaluator feature extraction, 256 pairs through both classifiers (91 tokens, 259 features, d=512, L=8), HIP vs stock "
            "PyTorch-ROCm ops | %.1f ms (%.0f pairs/s) vs %.1f ms |\n" % (v["ms_hip"], v["pairs_per_s_hip"], v["ms_stock_torch"]))
    if "| evaluator feature extraction" in old:
        old = re.sub(r"\| evaluator feature extraction[^\n]*\n", lambda m: rowe, old)
    else:
        old = old.replace("| CPU oracle on the GPU box", rowe + "| CPU oracle on the GPU box", 1)
hb = e["hbm_bound_kernels"]
row = ("| HBM-bound kernels at config 2 (`extra.hbm_bound_kernels`: algorithmic bytes / launch time, of 8 TB/s) | "
       + "; ".join("%s %.0f µs = %.0f %%" % (k.split(" (")[0], v["us"], 100 * v["frac_of_8TB_s"]) for k, v in hb.items()) + " |\n")
old = re.sub(r"\| HBM-bound kernels at config 2[^\n]*\n", lambda m: row, old)
open(p, "w").write(s[:a] + old + s[b:])
print("DESIGN.md table updated from", tag)
