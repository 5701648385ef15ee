"""One rank of the data-parallel EPOCH-LOOP test (tests/test_gpu_data_parallel.py): two of these on one GPU over gloo run
`DDPMMulTrainer.train(dataset, rank, world)` -- sharded sampler x fused step with the gradient exchange x rank-0
checkpoints x resume -- on the synthetic on-disk split of oracle/synth_dataset.py, started through
`hig_amd.parallel.run_distributed` (the packaged counterpart of tools/train.py:53-90).
usage: dp_train_worker.py <rank> <world> <port> <outdir>"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def build_model(seed):
    import hig_amd
    torch.manual_seed(seed)
    m = hig_amd.MotionInteractionTransformer(input_feats=263, num_frames=96, latent_dim=64, ff_size=128, num_layers=2,
                                             num_heads=8, text_latent_dim=64, num_text_layers=1, text_ff_size=128,
                                             text_num_heads=4)
    with torch.no_grad():       # the reference's zero-initialised tensors would keep every gradient upstream at zero
        for p in m.parameters():
            if p.requires_grad and float(p.abs().max()) == 0.0:
                p.normal_(0.0, 0.02)
    return m


def make_args(outdir, **kw):
    a = dict(device=torch.device("cuda"), diffusion_steps=1000, is_train=True, lr=2e-4, batch_size=2, num_epochs=2,
             log_every=2, save_latest=3, save_every_e=1, is_continue=False, model_dir=os.path.join(outdir, "model"),
             multi=True, label_path=None, cap_id=False, fused_step=True, num_workers=0)
    a.update(kw)
    return types.SimpleNamespace(**a)


def body(rank, world, outdir):
    import hig_amd
    from hig_amd.datasets import Text2MotionMulDataset
    from oracle import synth_dataset as SD
    opt = SD.write(os.path.join(outdir, "data_r%d" % rank))          # deterministic bytes: every rank its own copy
    mean, std = SD.stats(np.float32)

    class Logged(Text2MotionMulDataset):
        seen = []

        def __getitem__(self, i):
            Logged.seen.append(int(i))
            return super().__getitem__(i)

    ds = Logged(opt, mean.copy(), std.copy(), opt.split_file, times=1)
    if rank == 0:
        os.makedirs(os.path.join(outdir, "model"), exist_ok=True)
    dist.barrier()
    out = {"n": len(ds)}

    def run(tag, trainer):
        saves = []
        real_save = trainer.save
        trainer.save = lambda f, ep, total_it: (saves.append((os.path.basename(f), ep, total_it)), real_save(f, ep, total_it))[1]
        Logged.seen = []
        trainer.train(ds, rank, world)
        torch.cuda.synchronize()
        dist.barrier()
        out[tag + ".params"] = torch.cat([p.detach().reshape(-1).cpu() for p in trainer.encoder.parameters() if p.requires_grad])
        out[tag + ".seen"], out[tag + ".saves"] = list(Logged.seen), saves

    # phase 1: two epochs from scratch (different initial weights per rank: sync_replicas must repair that)
    run("p1", hig_amd.DDPMMulTrainer(make_args(outdir), build_model(seed=rank).to("cuda")))
    # phase 2: resume from rank 0's checkpoint into fresh models, one more epoch
    run("p2", hig_amd.DDPMMulTrainer(make_args(outdir, is_continue=True, num_epochs=3), build_model(seed=10 + rank).to("cuda")))
    # phase 3: the reference's own step (forward + update) on a BARE model in a 2-rank group: train() must average
    run("p3", hig_amd.DDPMMulTrainer(make_args(outdir, fused_step=False, num_epochs=1, model_dir=os.path.join(outdir, "model3")),
                                     build_model(seed=20 + rank).to("cuda")))
    torch.save(out, os.path.join(outdir, "train_rank%d.pt" % rank))


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    torch.cuda.set_device(0)
    from hig_amd import parallel
    if rank == 0:
        os.makedirs(os.path.join(outdir, "model3"), exist_ok=True)
    parallel.run_distributed(body, rank, world, "gloo", outdir, set_device=False)


if __name__ == "__main__":
    main()
