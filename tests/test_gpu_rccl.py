"""The RCCL path on ONE GPU: a single-rank "nccl" process group with HIG_FORCE_EXCHANGE=1 makes
DDPMTrainer issue every gradient all-reduce it would issue on 8 GPUs -- per decoder layer from the backward's host hook
on the communication stream (train_step_fused, parallel.OverlappedGradAllReduce), and between graph A and graph B of
the captured step (parallel.FlatGradAllReduce).  gloo stages through the host and hides stream-ordering hazards; RCCL
enqueues device kernels on its own stream.  An all-reduce over one rank is the identity: parameters, loss, gradient
norm and step counter must equal the same steps run without any process group, bit for bit."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def test_rccl_single_rank_exchange_is_the_identity(rccl_worker):
    import dp_worker
    import rccl_worker as rw
    outdir, codes, logs = rccl_worker
    assert codes == [0], "\n".join(logs)[-6000:]
    got = torch.load(os.path.join(outdir, "rccl.pt"))
    assert got["backend"] == "nccl"
    assert "HIG_FORCE_EXCHANGE" not in os.environ
    c, m, tr, gi, x0, noise = dp_worker.build_and_inputs()      # this process: no process group, no exchange
    # collectives that reached RCCL: 3 overlapped steps x (2 ranges per decoder layer + 2 tail ranges) + 3 captured steps x 1
    assert got["n_allreduce"] == 3 * (2 * c["L"] + 2) + 3, got["n_allreduce"]
    ref = rw.steps(tr, m, gi, x0, noise)
    assert got["step"] == ref["step"] == 6
    assert got["losses"] == ref["losses"]
    assert got["gnorm"] == ref["gnorm"]
    assert torch.equal(got["flat"], ref["flat"])
