"""Times hig_wgrad_bf16 (csrc/wgrad16.hip) alone on the training step's weight shapes, rotating over operand sets.
usage: python tools/wgrad_time.py [B]   (HIG_WG16_FORM=1: the round-4 kernel)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hig_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M = B * 196
L = _lib.lib()
dev = "cuda"
for J, K in ((512, 512), (1536, 512), (1024, 512), (512, 1024)):
    nb = 6
    dC = [torch.randn(M, J, device=dev).to(torch.bfloat16) for _ in range(nb)]
    act = [torch.randn(M, K, device=dev).to(torch.bfloat16) for _ in range(nb)]
    dW, dbias = torch.empty(J, K, device=dev), torch.empty(J, device=dev)
    n = L.hig_wgrad_bf16_scratch_floats(J, K, 0); slabs = torch.empty(n, device=dev)
    def run(i):
        _lib.check(L.hig_wgrad_bf16(_lib.ptr(dC[i % nb]), J, _lib.ptr(act[i % nb]), K, M, J, K, _lib.ptr(dW), _lib.ptr(dbias), 0, _lib.ptr(slabs), n, _lib.stream_ptr()))
    for i in range(nb): run(i)
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for i in range(4 * nb): run(i)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / (4 * nb) * 1e3)
    ts.sort()
    print("wgrad16 J=%-5d K=%-5d  %.1f us (median of 5; kernel + slab reduction)  %.0f TFLOP/s" % (J, K, ts[2], 2.0 * M * J * K / ts[2] / 1e6))
