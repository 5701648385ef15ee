"""One rank of the data-parallel trainer test (tests/test_gpu_data_parallel.py starts two of these on ONE GPU over
the gloo backend).  Each rank builds the same model, takes ITS half of the batch through DDPMTrainer's fused step
(step 1: eager launches; step 2: the hipGraph-captured variant) and writes its post-step parameters to `outdir`.
usage: dp_worker.py <rank> <world> <port> <outdir>"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def build_and_inputs(storage="f32"):
    import hig_amd
    from oracle import fill
    c = dict(fill.CASES["config1"])
    if storage == "bf16":              # bf16 storage serves head dims 64 / 128: config 1 with d = 256 over 4 heads
        c.update(d=256, H=4)
    m = hig_amd.MotionTransformer(input_feats=c["F"], num_frames=c["num_frames"], latent_dim=c["d"], ff_size=c["ff"],
                                  num_layers=c["L"], num_heads=c["H"], text_latent_dim=c["Lt"])
    m.load_state_dict(fill.fill_state_dict(m.state_dict()), strict=True)
    m = m.to("cuda").train()
    m.storage = storage
    B = 4      # global batch; every rank owns B / world samples
    inp = fill.inputs(B, c["T"], c["F"], c["d"], c["N"], c["Lt"], (60, 41, 17, 60), (0, 500, 999, 250))
    gi = {k: v.to("cuda") for k, v in inp.items()}
    x0 = [(fill.tensor_for("dp.x0.%d" % s, (B, c["T"], c["F"])) * 10).to("cuda") for s in range(2)]
    noise = [(fill.tensor_for("dp.noise.%d" % s, (B, c["T"], c["F"])) * 10).to("cuda") for s in range(2)]
    args = types.SimpleNamespace(device=torch.device("cuda"), diffusion_steps=1000, is_train=True, lr=2e-4,
                                 batch_size=B, num_epochs=1, log_every=50, save_latest=500, save_every_e=5,
                                 is_continue=False, model_dir="/tmp")
    return c, m, hig_amd.DDPMTrainer(args, m), gi, x0, noise


def shard(t, rank, world):
    n = t.shape[0] // world
    return t[rank * n:(rank + 1) * n].contiguous()


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c, m, tr, gi, x0, noise = build_and_inputs()
    if rank != 0:                      # replicas that start DIFFERENT: sync_replicas() must repair this
        with torch.no_grad():
            for p in m.core_parameters():
                p.add_(0.01 * (rank + 1))
    tr.sync_replicas()
    losses = []
    for s, step in enumerate((tr.train_step_fused, tr.train_step_captured)):
        loss = step(shard(x0[s], rank, world), shard(gi["t"], rank, world), shard(gi["length"], rank, world),
                    shard(gi["xf_proj"], rank, world), shard(gi["xf_out"], rank, world), noise=shard(noise[s], rank, world))
        losses.append(loss.item())
    torch.cuda.synchronize()
    fp = m.flat_params()
    torch.save({"flat": fp.flat[:fp.core_numel].cpu(), "losses": losses,
                "gnorm": tr.fused_state()["gnorm"].item(), "step": tr.fused_state()["step"].item()},
               os.path.join(outdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
