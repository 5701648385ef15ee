"""Times hig_gemm_bf16 on the denoiser's shapes (bf16 storage), rotating over operand sets larger than the Infinity
Cache.  Tuning knobs are read by the library from the environment (HIG_BF16_TILE=64|128, HIG_BF16_PERCU, HIG_BF16_THR):
run one process per setting.  usage: gemm16_bench.py [B] [cfg5]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hig_amd import _lib
if os.environ.get("HIG_LIB_ALT"): _lib.LIB_PATH = os.environ["HIG_LIB_ALT"]   # (A/B of a variant build)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M = B * 196
SHAPES = [("qkv", M, 1536, 512, _lib.EPI_BIAS), ("sty_out", M, 512, 512, _lib.EPI_BIAS_RES),
          ("ffn1", M, 1024, 512, _lib.EPI_BIAS_GELU), ("ffn2", M, 512, 1024, _lib.EPI_BIAS), ("ca_q", M, 512, 512, _lib.EPI_BIAS),
          ("text_kv", B * 77, 1024, 256, _lib.EPI_BIAS), ("emb_ss", B, 24576, 2048, _lib.EPI_BIAS),
          ("te2", B, 2048, 2048, _lib.EPI_BIAS_SILU)]
if len(sys.argv) > 2 and sys.argv[2] == "cfg5":   # BASELINE config 5: B = 32, T = 300, d = 1024, ff = 1024
    M = B * 300
    SHAPES = [("qkv", M, 3072, 1024, _lib.EPI_BIAS), ("sty_out", M, 1024, 1024, _lib.EPI_BIAS_RES),
              ("ffn1", M, 1024, 1024, _lib.EPI_BIAS_GELU), ("ffn2", M, 1024, 1024, _lib.EPI_BIAS),
              ("text_kv", B * 77, 2048, 256, _lib.EPI_BIAS)]
L = _lib.lib()
dev = "cuda"
tag = " ".join("%s=%s" % (k, os.environ[k]) for k in ("HIG_BF16_TILE", "HIG_BF16_PERCU", "HIG_BF16_THR", "HIG_BF16_WS", "HIG_BF16_WS_NWJ", "HIG_BF16_WS_SLOTS") if k in os.environ)
for name, I, J, R, epi in SHAPES:
    per = (I * R + I * J * 2) * 2
    nb = max(2, min(16, int(600e6 // per) + 1))
    Xs = [torch.randn(I, R, device=dev).to(torch.bfloat16) for _ in range(nb)]
    Cs = [torch.empty(I, J, device=dev, dtype=torch.bfloat16) for _ in range(nb)]
    Rs = [torch.randn(I, J, device=dev).to(torch.bfloat16) for _ in range(nb)] if epi == _lib.EPI_BIAS_RES else None
    W = (torch.randn(J, R, device=dev) * 0.05).to(torch.bfloat16)
    b = torch.randn(J, device=dev)
    descs = []
    for k in range(nb):
        d = _lib.Gemm16Desc()
        d.X, d.ldx, d.Y, d.ldy, d.C, d.ldc, d.c_f32 = Xs[k].data_ptr(), R, W.data_ptr(), R, Cs[k].data_ptr(), J, 0
        d.I, d.J, d.R, d.epi, d.bias = I, J, R, epi, b.data_ptr()
        if Rs:
            d.res, d.ldr, d.res_f32 = Rs[k].data_ptr(), J, 0
        descs.append(d)
    for k in range(nb):
        _lib.check(L.hig_gemm_bf16(C.byref(descs[k]), _lib.stream_ptr()))
    reps = 4 * nb
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for k in range(reps):
        _lib.check(L.hig_gemm_bf16(C.byref(descs[k % nb]), _lib.stream_ptr()))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = 2.0 * I * J * R
    by = (I * R + J * R + I * J * (2 if Rs else 1)) * 2
    print("%-22s %-8s I=%-6d J=%-6d R=%-5d %8.1f us %8.1f TFLOP/s %7.0f GB/s" % (tag, name, I, J, R, us, fl / us / 1e6, by / us / 1e3))
