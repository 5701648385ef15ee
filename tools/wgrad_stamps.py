"""Where a workgroup of the weight-gradient kernel (csrc/wgrad16.hip, wgrad16x_kernel) spends its time: s_memtime stamps.
usage: python tools/wgrad_stamps.py [J] [K] [B] [hot]"""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hig_amd import _lib
J = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
hot = len(sys.argv) > 4 and sys.argv[4] == "hot"
M = B * 196
L = _lib.lib(); dev = "cuda"
dC = torch.randn(M, J, device=dev).to(torch.bfloat16); act = torch.randn(M, K, device=dev).to(torch.bfloat16)
dW, dbias = torch.empty(J, K, device=dev), torch.empty(J, device=dev)
n = L.hig_wgrad_bf16_scratch_floats(J, K, 0); slabs = torch.empty(n, device=dev)
stamps = torch.zeros(8192, dtype=torch.int64, device=dev)
L.hig_wgrad16_debug_stamps(C.c_void_p(stamps.data_ptr()))
junk = torch.ones(256 << 20, device=dev)
for it in range(3):
    if not hot: junk.sum().item()
    stamps.zero_()
    _lib.check(L.hig_wgrad_bf16(_lib.ptr(dC), J, _lib.ptr(act), K, M, J, K, _lib.ptr(dW), _lib.ptr(dbias), 0, _lib.ptr(slabs), n, _lib.stream_ptr()))
    torch.cuda.synchronize()
    m = stamps[:4096].view(256, 16).cpu(); l = stamps[4096:].view(256, 16).cpu()
    ok = m[:, 0] > 0
    m, l = m[ok], l[ok]
    print("J=%d K=%d B=%d %s run %d: %d workgroups, chunks %d..%d" % (J, K, B, "hot" if hot else "cold", it, len(m), m[:, 14].min(), m[:, 14].max()))
    def show(label, t, a, b):
        okk = (t[:, a] > 0) & (t[:, b] > 0)
        d = (t[okk, b] - t[okk, a]).double()
        if len(d): print("   %-52s median %7.0f cycles   p10 %7.0f   p90 %7.0f" % (label, d.median(), d.quantile(0.1), d.quantile(0.9)))
    show("matrix: start -> first chunks landed (B_0)", m, 0, 1)
    show("matrix: chunk loop", m, 1, 2)
    show("matrix: group sum + stores", m, 2, 3)
    show("matrix: whole", m, 0, 3)
    show("loader: start -> chunks 0, 1 landed", l, 0, 1)
    show("loader iteration 4: DMA issue (8 instructions)", l, 2, 3)
    show("loader iteration 4: wait for chunk 6", l, 3, 4)
    show("loader: chunk loop", l, 1, 5)
L.hig_wgrad16_debug_stamps(None)
