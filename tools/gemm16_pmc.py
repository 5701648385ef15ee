#!/usr/bin/env python3
"""Launches ONE bf16-storage GEMM shape (default: FFN linear1, M=12544 K=512 N=1024, bias+GELU) a few times, every
launch from cold caches (a 1 GiB read sweep in between), for rocprofv3 --pmc passes.  usage: gemm16_pmc.py [name] [B]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from hig_amd import _lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ffn1"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
M = B * 196
I, J, R, epi = {"ffn1": (M, 1024, 512, _lib.EPI_BIAS_GELU), "qkv": (M, 1536, 512, _lib.EPI_BIAS),
                "sty_out": (M, 512, 512, _lib.EPI_BIAS_RES), "ffn2": (M, 512, 1024, _lib.EPI_BIAS)}[name]
dev = "cuda"
X = torch.randn(I, R, device=dev).to(torch.bfloat16)
W = (torch.randn(J, R, device=dev) * 0.05).to(torch.bfloat16)
b = torch.randn(J, device=dev)
out = torch.empty(I, J, device=dev, dtype=torch.bfloat16)
res = torch.randn(I, J, device=dev).to(torch.bfloat16)
d = _lib.Gemm16Desc()
d.X, d.ldx, d.Y, d.ldy, d.C, d.ldc, d.c_f32 = X.data_ptr(), R, W.data_ptr(), R, out.data_ptr(), J, 0
d.I, d.J, d.R, d.epi, d.bias = I, J, R, epi, b.data_ptr()
if epi == _lib.EPI_BIAS_RES:
    d.res, d.ldr, d.res_f32 = res.data_ptr(), J, 0
junk = torch.ones(256 << 20, device=dev)
torch.cuda.synchronize()
for _ in range(6):
    junk.sum().item()
    _lib.check(_lib.lib().hig_gemm_bf16(C.byref(d), _lib.stream_ptr()))
torch.cuda.synchronize()
print("done", name, I, J, R)
