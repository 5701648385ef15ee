"""ONE rank on the real RCCL backend (tests/test_gpu_rccl.py): `init_process_group("nccl", world_size=1)` on the single
GPU of the box with HIG_FORCE_EXCHANGE=1, so that DDPMTrainer's fused steps really issue their gradient all-reduces --
the per-layer ones from the host hook on the side stream while the backward is still running on its two streams
(train_step_fused), and the one between graph A and graph B (train_step_captured).  An all-reduce over one rank is the
identity, so every number must equal the step without a process group, bit for bit.
usage: rccl_worker.py <port> <outdir>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def steps(tr, m, gi, x0, noise):
    """The two fused steps of dp_worker.py on the FULL batch; returns what the test compares."""
    losses = []
    for s, step in enumerate((tr.train_step_fused, tr.train_step_captured)):
        loss = step(x0[s], gi["t"], gi["length"], gi["xf_proj"], gi["xf_out"], noise=noise[s])
        losses.append(loss.item())
    for s in range(2):   # and two more of each, replaying the captured graphs / re-arming the overlapped exchange
        losses.append(tr.train_step_captured(x0[s], gi["t"], gi["length"], gi["xf_proj"], gi["xf_out"], noise=noise[1 - s]).item())
        losses.append(tr.train_step_fused(x0[1 - s], gi["t"], gi["length"], gi["xf_proj"], gi["xf_out"], noise=noise[s]).item())
    torch.cuda.synchronize()
    fp = m.flat_params()
    return {"flat": fp.flat[:fp.core_numel].cpu(), "losses": losses, "gnorm": tr.fused_state()["gnorm"].item(),
            "step": tr.fused_state()["step"].item()}


def main():
    port, outdir = sys.argv[1], sys.argv[2]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1", HIG_FORCE_EXCHANGE="1")
    import dp_worker
    from hig_amd import parallel

    def body(rank, world):
        assert dist.get_backend() == "nccl" and world == 1
        assert parallel.exchange_active()
        c, m, tr, gi, x0, noise = dp_worker.build_and_inputs()
        out = steps(tr, m, gi, x0, noise)
        out["backend"] = dist.get_backend()
        out["n_allreduce"] = body.count
        torch.save(out, os.path.join(outdir, "rccl.pt"))

    # count the collectives that really reach the backend
    body.count = 0
    real = dist.all_reduce

    def counting(*a, **k):
        body.count += 1
        return real(*a, **k)
    dist.all_reduce = counting
    parallel.run_distributed(body, 0, 1, "nccl")


if __name__ == "__main__":
    main()
