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


def pair_steps(storage):
    """Two-person model, PIT mode (DDPMMulTrainer): two eager fused steps (overlapped exchange from the hooked backward with 4
    stylization blocks per layer) and two captured ones."""
    import types
    import hig_amd
    from oracle import fill
    c = dict(B=2, T=33, F=20, d=256, H=4, L=2, ff=256, N=77, Lt=64, num_frames=40, lengths=(33, 12), t=(4, 700))
    m = hig_amd.MotionInteractionTransformer(input_feats=c["F"], num_frames=c["num_frames"], latent_dim=c["d"], ff_size=c["ff"],
                                             num_layers=c["L"], num_heads=c["H"], text_latent_dim=c["Lt"], storage=storage)
    m.load_state_dict(fill.fill_state_dict(m.state_dict()), strict=True)
    m = m.to("cuda").train()
    args = types.SimpleNamespace(device=torch.device("cuda"), diffusion_steps=1000, is_train=True, lr=2e-4, batch_size=c["B"],
                                 num_epochs=1, log_every=50, save_latest=500, save_every_e=5, is_continue=False, model_dir="/tmp",
                                 multi=True, label_path=None, cap_id=False)
    tr = hig_amd.DDPMMulTrainer(args, m)
    B, rows = c["B"], 4 * c["B"]
    x0 = (fill.tensor_for("rccl2.x0", (2 * B, c["T"], c["F"])) * 10).to("cuda")
    noise = [(fill.tensor_for("rccl2.nz.%d" % k, x0.shape) * 10).to("cuda") for k in range(2)]
    tt, length = torch.tensor(c["t"], device="cuda"), torch.tensor(c["lengths"], device="cuda")
    xp = (fill.tensor_for("rccl2.xp", (rows, 4 * c["d"])) * 10).to("cuda")
    xo = (fill.tensor_for("rccl2.xo", (rows, c["N"], c["Lt"])) * 10).to("cuda")
    losses = [step(x0, tt, length, xp, xo, noise=noise[k]).item()
              for step in (tr.train_step_fused, tr.train_step_captured) for k in range(2)]
    torch.cuda.synchronize()
    fp = m.flat_params()
    return {"flat": fp.flat[:fp.core_numel].cpu(), "losses": losses, "gnorm": tr.fused_state()["gnorm"].item(),
            "captured_form": tr.fused_state().get("captured_form"), "L": c["L"]}


def main():
    port, outdir = sys.argv[1], sys.argv[2]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1", HIG_FORCE_EXCHANGE="1")
    import dp_worker
    from hig_amd import parallel

    def body(rank, world):
        assert dist.get_backend() == "nccl" and world == 1
        assert parallel.exchange_active()
        out = {"backend": dist.get_backend()}
        # every form of the step: fp32 / bf16 storage x the captured step with the exchange inside the graph / split around it
        for storage in ("f32", "bf16"):
            for in_graph in (True, False):
                c, m, tr, gi, x0, noise = dp_worker.build_and_inputs(storage)
                tr.opt.capture_exchange = in_graph
                body.count = 0
                r = steps(tr, m, gi, x0, noise)
                r["n_allreduce"] = body.count
                r["captured_form"] = tr.fused_state().get("captured_form")
                r["capture_error"] = tr.fused_state().get("capture_exchange_error")
                out["%s/%s" % (storage, "in_graph" if in_graph else "split")] = r
        for storage in ("f32", "bf16"):
            body.count = 0
            r = pair_steps(storage)
            r["n_allreduce"] = body.count
            out["pair/" + storage] = r
        # bf16 on the wire (opt-in): what one extra rounding of every gradient element does to the clip norm and the update
        c, m, tr, gi, x0, noise = dp_worker.build_and_inputs("bf16")
        tr.opt.grad_wire = "bf16"
        r = steps(tr, m, gi, x0, noise)
        r["wire"] = tr.fused_state()["overlap"].wire
        out["bf16/wire_bf16"] = r
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
