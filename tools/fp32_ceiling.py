"""Ceiling statement for the fp32 headline: GEMM FLOPs of one forward at 0.70 of the fp32 MFMA peak + the
measured floor of the non-GEMM kernels (each timed ALONE by bench.py: `extra.hbm_bound_kernels`), against the achieved forward.
usage: python tools/fp32_ceiling.py [profiles/r05_bench.json]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_bench.json")
d = json.loads(open(path).read().strip().splitlines()[-1])
B, T, F, dm, L, ff, N, Lt, E = 64, 196, 150, 512, 8, 1024, 77, 256, 2048
M, Mt = B * T, B * N
per_layer = 2 * M * dm * 3 * dm + 4 * 2 * M * dm * dm + 2 * 2 * M * dm * ff + 2 * Mt * Lt * 2 * dm      # q/k/v, 3 stylization out + cross-attention query, FFN, text K/V
gemm = L * per_layer + 2 * 2 * M * F * dm + 2 * B * (dm * E + E * E) + 2 * B * E * (3 * L * 2 * dm)      # + joint_embed / out, time embedding, stylization (scale, shift) GEMM
peak = d["roofline"]["peak"] * 1e12
k = d["extra"]["hbm_bound_kernels"]
us = lambda name: next(v["us"] for kk, v in k.items() if kk.startswith(name))
non = 16 * us("linattn_apply (") + 8 * us("linattn_ctx (") + 8 * us("linattn_ctx (") * N / T + 24 * us("ln_mod_silu") + 1 * us("layernorm")
t_gemm = gemm / (0.70 * peak) * 1e3
print("GEMM work of one forward: %.1f GFLOP -> %.3f ms at 0.70 of %.1f TFLOP/s" % (gemm / 1e9, t_gemm, peak / 1e12))
print("non-GEMM kernels, each timed alone: 16 apply + 8 context builds over T + 8 over the %d text tokens (scaled) + 24 stylization fronts + 1 LayerNorm = %.3f ms" % (N, non / 1e3))
print("ceiling %.3f ms; achieved (headline of this file) %.3f ms: %.1f %% apart" % (t_gemm + non / 1e3, d["ms_per_step"], 100 * (d["ms_per_step"] / (t_gemm + non / 1e3) - 1)))
