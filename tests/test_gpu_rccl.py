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
    for storage in ("f32", "bf16"):
        c, m, tr, gi, x0, noise = dp_worker.build_and_inputs(storage)      # this process: no process group, no exchange
        ref = rw.steps(tr, m, gi, x0, noise)
        per_step = 2 * c["L"] + 2             # overlapped exchange: 2 ranges per decoder layer + 2 tail ranges
        for form in ("in_graph", "split"):
            g = got["%s/%s" % (storage, form)]
            # collectives that reached RCCL from Python.  3 eager overlapped steps; the captured step with the exchange
            # inside the graph issues them twice (warm-up + capture) and its replays none; the split form one flat
            # all-reduce per step
            if form == "in_graph":
                assert g["capture_error"] is None, g["capture_error"]
                assert g["captured_form"] == "one graph, exchange inside"
                assert g["n_allreduce"] == 3 * per_step + 2 * per_step, g["n_allreduce"]
            else:
                assert g["captured_form"] == "graph A | all-reduce | graph B"
                assert g["n_allreduce"] == 3 * per_step + 3, g["n_allreduce"]
            assert g["step"] == ref["step"] == 6
            assert g["losses"] == ref["losses"], (storage, form)
            assert g["gnorm"] == ref["gnorm"], (storage, form)
            assert torch.equal(g["flat"], ref["flat"]), (storage, form)
        # two-person model, PIT mode: 2 eager overlapped steps + warm-up and capture of the in-graph form
        pg, pref = got["pair/" + storage], rw.pair_steps(storage)
        assert pg["captured_form"] == "one graph, exchange inside" and pref["captured_form"] == "one graph"
        assert pg["n_allreduce"] == 4 * (2 * pg["L"] + 2), pg["n_allreduce"]
        assert pg["losses"] == pref["losses"] and pg["gnorm"] == pref["gnorm"], (storage, pg["losses"], pref["losses"])
        assert torch.equal(pg["flat"], pref["flat"]), storage
        if storage == "bf16":
            # bf16 on the wire: every gradient element rounded once more (relative 2^-9) before the sum over ranks.  The clip
            # norm is a sum of 81 M squares: the roundings average out
            w = got["bf16/wire_bf16"]
            assert w["step"] == 6
            assert w["wire"] == "bf16" and not torch.equal(w["flat"], ref["flat"])      # (the option really took effect)
            assert abs(w["gnorm"] - ref["gnorm"]) <= 2e-3 * abs(ref["gnorm"]), (w["gnorm"], ref["gnorm"])   # (after 6 steps)
            assert all(abs(a - b) <= 1e-3 * abs(b) for a, b in zip(w["losses"], ref["losses"])), (w["losses"], ref["losses"])
            d = (w["flat"] - ref["flat"]).double()
            # Adam normalises each element's update to ~lr, so a flipped rounding moves a parameter by a fraction of lr = 2e-4
            stats = (float(d.abs().max()), float(d.norm() / ref["flat"].double().norm()))
            assert stats[0] <= 6 * 2 * 2e-4 * 1.01 and stats[1] < 1e-3, stats
