import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from hig_amd import _lib
B, T, H, hd = 64, 196, 8, 64
d = H * hd; dev = "cuda"
L = _lib.lib(); s = _lib.stream_ptr()
qkv = torch.randn(B * T, 3 * d, device=dev).to(torch.bfloat16)
dy = torch.randn(B * T, d, device=dev).to(torch.bfloat16)
A = torch.randn(B, H, hd, hd, device=dev)
dq = torch.empty(B * T, d, device=dev, dtype=torch.bfloat16)
dA = torch.empty(B, H, hd, hd, device=dev)
bscr = torch.zeros(L.hig_linattn_bwd_scratch_floats(B, T, H, hd), device=dev)
for _ in range(5):
    _lib.check(L.hig_linattn_apply_bwd_bf16(_lib.ptr(dy), d, _lib.ptr(qkv), 3 * d, _lib.ptr(A), _lib.ptr(dq), d, _lib.ptr(dA), B, T, H, hd, _lib.ptr(bscr), s))
torch.cuda.synchronize()
