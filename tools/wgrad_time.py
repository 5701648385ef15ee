"""Weight-gradient GEMM (dW = dC^T . act over the M rows, reduce-slow operands, split-R slabs + reduction) against
forward GEMMs of the same FLOPs.  usage: python tools/wgrad_time.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hig_amd import _lib
L, s, dev = _lib.lib(), _lib.stream_ptr(), "cuda"
M = 12544


def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (n_out, k_in) in ((512, 512), (1536, 512), (1024, 512), (512, 1024)):
    NB = 4
    dCs = [torch.randn(M, n_out, device=dev) for _ in range(NB)]
    acts = [torch.randn(M, k_in, device=dev) for _ in range(NB)]
    dW = torch.empty(n_out, k_in, device=dev)
    bsum = torch.empty(n_out, device=dev)
    descs = []
    for dC, act in zip(dCs, acts):
        d = _lib.GemmDesc()
        d.X, d.ldx, d.x_rs, d.Y, d.ldy, d.y_rs = dC.data_ptr(), n_out, 1, act.data_ptr(), k_in, 1
        d.C, d.ldc, d.I, d.J, d.R = dW.data_ptr(), k_in, n_out, k_in, M
        d.xcolsum = bsum.data_ptr()
        descs.append(d)
    nfl = L.hig_gemm_split_scratch_floats(C.byref(descs[0]), 0)
    slabs = torch.empty(nfl, device=dev)
    it = [0]

    def wg():
        _lib.check(L.hig_gemm_split(C.byref(descs[it[0] % NB]), 0, slabs.data_ptr(), nfl, s)); it[0] += 1

    us = timeit(wg)
    fl = 2.0 * M * n_out * k_in
    # forward GEMM of the same FLOPs: (M, k_in) x (n_out, k_in)^T
    W = torch.randn(n_out, k_in, device=dev)
    outs = [torch.empty(M, n_out, device=dev) for _ in range(NB)]
    tail = torch.zeros(L.hig_gemm_tail_ws_bytes(), dtype=torch.uint8, device=dev)
    fd = []
    for act, o in zip(acts, outs):
        d = _lib.GemmDesc()
        d.X, d.ldx, d.Y, d.ldy, d.C, d.ldc, d.I, d.J, d.R = act.data_ptr(), k_in, W.data_ptr(), k_in, o.data_ptr(), n_out, M, n_out, k_in
        fd.append(d)

    def fw():
        _lib.check(L.hig_gemm_ws(C.byref(fd[it[0] % NB]), tail.data_ptr(), tail.numel(), s)); it[0] += 1

    usf = timeit(fw)
    print("dW %4d x %4d over M=%d: wgrad %6.1f us (%5.1f TFLOP/s)   forward GEMM of the same FLOPs %6.1f us (%5.1f TFLOP/s)" %
          (n_out, k_in, M, us, fl / us * 1e-6, usf, fl / usf * 1e-6), flush=True)
