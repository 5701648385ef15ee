"""Input pipeline: our Text2MotionMulDataset / DeviceMotionBank against what the
reference's own dataset class produced on the same synthetic files (golden g11), bit for bit."""
import random

import numpy as np
import pytest
import torch

import hig_amd  # noqa: F401
from hig_amd.datasets import DeviceMotionBank, Text2MotionMulDataset, build_dataloader
from hig_amd.datasets.mul_dataset import frame_indices
from oracle import synth_dataset as SD


def make(tmp_path, tag):
    opt = SD.write(str(tmp_path))
    mean, std = SD.stats(np.float32 if tag == "f32" else np.float64)
    ds = Text2MotionMulDataset(opt, mean.copy(), std.copy(), opt.split_file, times=2,
                               label_path=opt.label_path if tag == "f64" else None)
    return opt, ds


def check_item(g, tag, i, c1, c2, m1, m2, ml, fid):
    assert [c1, c2, str(int(ml)), fid] == g["%s.%d.meta" % (tag, i)].tolist()
    for nm, m in (("m1", m1), ("m2", m2)):
        m = np.ascontiguousarray(m)
        assert m.dtype == np.float32 and tuple(m.shape) == tuple(g["%s.%d.%s.shape" % (tag, i, nm)])
        assert np.array_equal(m[[0, 1, 2, 45, 89, 90]].view(np.uint32), g["%s.%d.%s.rows" % (tag, i, nm)].view(np.uint32))
        assert np.array_equal(m[:, ::37].view(np.uint32), g["%s.%d.%s.cols" % (tag, i, nm)].view(np.uint32))
        assert np.uint64(m.view(np.uint32).astype(np.uint64).sum()) == g["%s.%d.%s.bits" % (tag, i, nm)]


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_host_dataset_matches_reference_golden(gold, tmp_path, tag):
    g = gold("g11_dataset.npz")
    opt, ds = make(tmp_path, tag)
    assert list(ds.name_list) == g[tag + ".names"].tolist() and len(ds) == int(g[tag + ".len"])
    import os
    assert np.array_equal(np.load(os.path.join(opt.meta_dir, "mean.npy")), g[tag + ".mean_saved"])
    assert np.array_equal(np.load(os.path.join(opt.meta_dir, "std.npy")), g[tag + ".std_saved"])   # feat_bias rule
    random.seed(1234)
    for i in range(len(ds)):
        check_item(g, tag, i, *ds[i])
    x = np.ones(263, dtype=np.float32)
    assert np.allclose(ds.inv_transform(x), ds.std + ds.mean)


def test_frame_index_rule_edges():
    class R:                      # records the randint range the window rule asks for
        def randint(self, a, b):
            self.ab = (a, b)
            return b
    r = R()
    ix = frame_indices(20, r)     # short clip: init row, 20 frames, last frame repeated 70 times
    assert ix[0] == 20 and ix[1:21].tolist() == list(range(20)) and set(ix[21:].tolist()) == {19} and len(ix) == 91
    ix = frame_indices(90, r)     # exactly 90 frames: no room to shift
    assert r.ab == (0, 0) and ix.tolist() == [90] + list(range(90))
    ix = frame_indices(150, r)    # shift in [0, 150 - 89 - 1 - 1]
    assert r.ab == (0, 59) and ix.tolist() == [150] + list(range(59, 149))


def test_build_dataloader_shards_like_the_reference_sampler(tmp_path):
    _, ds = make(tmp_path, "f32")
    seen = []
    for rank in range(2):
        dl = build_dataloader(ds, rank, 2, samples_per_gpu=3, workers_per_gpu=0, shuffle=True, drop_last=True)
        idx = list(dl.sampler)
        g = torch.Generator()
        g.manual_seed(0)
        perm = torch.randperm(len(ds), generator=g).tolist()
        assert idx == perm[rank::2]
        seen += idx
        cap1, cap2, m1, m2, lens, ids = next(iter(dl))
        assert m1.shape == (3, 91, 263) and len(cap1) == 3 and lens.shape == (3,)
    assert sorted(seen) == list(range(len(ds)))


def test_cap_id_needs_a_caption_table(tmp_path):
    opt = SD.write(str(tmp_path))
    opt.cap_id = True
    mean, std = SD.stats()
    with pytest.raises(ValueError, match="caption_table"):
        Text2MotionMulDataset(opt, mean, std, opt.split_file)
    table = {50: SD.CAPTIONS[0], 51: SD.CAPTIONS[1], 52: SD.CAPTIONS[2], 53: SD.CAPTIONS[3]}
    ds = Text2MotionMulDataset(opt, mean, std, opt.split_file, caption_table=table)
    c1, c2 = ds[0][:2]
    assert isinstance(c1, list) and isinstance(c1[0], int) and 0 <= c1[0] < 7


def test_device_bank_refuses_cpu(tmp_path):
    _, ds = make(tmp_path, "f32")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        DeviceMotionBank(ds, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_device_bank_batches_match_reference_golden(gold, tmp_path, tag):
    g = gold("g11_dataset.npz")
    _, ds = make(tmp_path, tag)
    bank = DeviceMotionBank(ds, "cuda")
    random.seed(1234)                       # same `random` stream as the golden's item-by-item loop
    n, bs = len(ds), 5
    for start in range(0, n, bs):
        items = list(range(start, min(n, start + bs)))
        cap1, cap2, m1, m2, lens, ids = bank.make_batch(items)
        assert m1.is_cuda and m1.shape == (len(items), 91, 263)
        assert m1.data_ptr() + 4 * m1.numel() == m2.data_ptr()          # one (2B, 91, F) buffer
        for k, i in enumerate(items):
            check_item(g, tag, i, cap1[k], cap2[k], m1[k].cpu().numpy(), m2[k].cpu().numpy(), lens[k], ids[k])


@pytest.mark.gpu
def test_device_bank_feeds_the_two_person_trainer(tmp_path):
    import types
    _, ds = make(tmp_path, "f32")
    bank = DeviceMotionBank(ds, "cuda")
    m = hig_amd.MotionInteractionTransformer(263, num_frames=90, latent_dim=64, ff_size=128, num_layers=2, num_heads=8,
                                             text_latent_dim=32).to("cuda")
    args = types.SimpleNamespace(device=torch.device("cuda"), diffusion_steps=1000, is_train=True, lr=2e-4, batch_size=4,
                                 num_epochs=1, log_every=50, save_latest=500, save_every_e=5, is_continue=False,
                                 model_dir=str(tmp_path), multi=True, label_path=None, cap_id=False)
    tr = hig_amd.DDPMMulTrainer(args, m)
    tr.opt_encoder = torch.optim.Adam(m.parameters(), lr=2e-4)
    random.seed(7)
    tr.train_mode()
    for start in (0, 4):
        tr.forward(bank.make_batch(list(range(start, start + 4))))
        logs = tr.update()
    assert tr.fake_noise.shape == (16, 91, 263) and np.isfinite(logs["loss_mot_rec"])


@pytest.mark.gpu
def test_end_to_end_two_person_pipeline(tmp_path):
    """Disk -> HBM-resident bank -> captured PIT training steps (text head inside) -> hipGraph sampling ->
    world-space joints: the whole two-person path of tools/train.py + tools/visualization.py on the device."""
    import types
    from hig_amd.models import gaussian_diffusion as gdm
    from hig_amd.utils.motion_process import generated_to_joints
    _, ds = make(tmp_path, "f32")
    bank = DeviceMotionBank(ds, "cuda")
    torch.manual_seed(0)
    m = hig_amd.MotionInteractionTransformer(263, num_frames=90, latent_dim=64, ff_size=128, num_layers=2, num_heads=8,
                                             text_latent_dim=32).to("cuda")
    args = types.SimpleNamespace(device=torch.device("cuda"), diffusion_steps=1000, is_train=True, lr=2e-3, batch_size=4,
                                 num_epochs=1, log_every=50, save_latest=500, save_every_e=5, is_continue=False,
                                 model_dir=str(tmp_path), multi=True, label_path=None, cap_id=False)
    tr = hig_amd.DDPMMulTrainer(args, m)
    random.seed(11)
    np.random.seed(11)             # the timestep sampler draws with numpy, the noise with torch
    torch.manual_seed(11)
    tr.train_mode()
    losses = []
    for it in range(60):
        items = [(4 * it + k) % len(ds) for k in range(4)]
        losses.append(tr.train_fused_batch(bank.make_batch(items), captured=True).item())
    assert all(np.isfinite(losses))
    assert abs(losses[0] - 1.0) < 0.1            # zero-initialised output head: the first loss is E[noise^2]
    assert np.mean(losses[-10:]) < 0.95 * np.mean(losses[:3])  # and it trains (per-step losses are noisy: t is random)
    assert tr.fused_state()["step"].item() == 60
    # sampling (50-step schedule, hipGraph loop) and joint recovery, all on the device
    tr.diffusion = hig_amd.GaussianDiffusion(betas=gdm.get_named_beta_schedule("linear", 50),
                                             model_mean_type=gdm.ModelMeanType.EPSILON,
                                             model_var_type=gdm.ModelVarType.FIXED_SMALL, loss_type=gdm.LossType.MSE)
    out = tr.generate_batch(["one pushes the other"], ["one is pushed by the other"], torch.tensor([91]), 263)
    assert out.shape == (2, 90, 263) or out.shape == (2, 91, 263)
    joints = generated_to_joints(out, ds.mean, ds.std, ds.init_mean, ds.init_std, joints_num=22)
    assert joints.shape == (2, out.shape[1] - 1, 22, 3) and torch.isfinite(joints).all()
