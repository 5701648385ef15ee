"""CPU-only checks (`-m "not gpu"`): the C-ABI library loads and exports every symbol
include/hig.h declares, the host-side mirror of the reference API behaves like the reference
(schedule tables == golden G1, sampler sharding, flat parameter layout, state-dict contract),
the product path refuses to run without a ROCm device, and the data-parallel exchange has DDP
semantics (world_size-2 gloo)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

import hig_amd  # noqa: E402
from hig_amd import _lib  # noqa: E402
from hig_amd.models import gaussian_diffusion as gdm  # noqa: E402
from hig_amd.parallel import ShardedSampler  # noqa: E402
from oracle import fill  # noqa: E402


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.lib()
    header = open(os.path.join(ROOT, "include", "hig.h")).read()
    declared = set(re.findall(r"\b(hig_[a-z0-9_]+)\s*\(", header))
    declared -= {"hig_dims", "hig_gemm_desc", "hig_stream_t"}
    assert declared, "no prototypes parsed"
    for sym in sorted(declared):
        assert hasattr(lib, sym), "libhig.so does not export %s" % sym
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    assert lib.hig_version() >= 100


def test_abi_struct_sizes_and_dim_validation():
    lib = _lib.lib()
    d = _lib.Dims(B=2, T=16, F=12, d=64, H=8, ff=128, L=2, N=77, Lt=32, num_frames=20, attn_kind=0, prec=0)
    assert lib.hig_workspace_bytes(C.byref(d), 0) > 0
    assert lib.hig_workspace_bytes(C.byref(d), 1) > lib.hig_workspace_bytes(C.byref(d), 0)
    assert lib.hig_textctx_bytes(C.byref(d), 1) > 0 and lib.hig_bwd_workspace_bytes(C.byref(d)) > 0
    bad = _lib.Dims(B=2, T=30, F=12, d=64, H=8, ff=128, L=2, N=77, Lt=32, num_frames=20, attn_kind=0, prec=0)
    assert lib.hig_workspace_bytes(C.byref(bad), 0) < 0          # T > num_frames
    assert "num_frames" in _lib.last_error()
    bad = _lib.Dims(B=2, T=16, F=12, d=60, H=8, ff=128, L=2, N=77, Lt=32, num_frames=20, attn_kind=0, prec=0)
    assert lib.hig_workspace_bytes(C.byref(bad), 0) < 0          # d % H != 0
    # NULL arguments are rejected with an error code, never a crash
    assert lib.hig_rowstats(None, 4, 1, 4, None, None) == -1
    # evaluator encoders: size query and validation, no compute
    e = _lib.EvalDims(B=4, T=91, F=259, d=512, H=8, ff=1024, L=8, C=26, cls=0, prec=0)
    n0 = lib.hig_eval_encoder_workspace_bytes(C.byref(e))
    e.cls = 1
    assert 0 < n0 < lib.hig_eval_encoder_workspace_bytes(C.byref(e))        # the [cls] token adds a row per pair
    ok128 = _lib.EvalDims(B=4, T=91, F=259, d=512, H=4, ff=1024, L=8, C=26, cls=0, prec=0)
    assert lib.hig_eval_encoder_workspace_bytes(C.byref(ok128)) > 0               # head dim 128: matrix-core attention
    bad = _lib.EvalDims(B=4, T=91, F=259, d=512, H=2, ff=1024, L=8, C=26, cls=0, prec=0)
    assert lib.hig_eval_encoder_workspace_bytes(C.byref(bad)) < 0 and "head dim" in _lib.last_error()
    bad = _lib.EvalDims(B=4, T=1, F=259, d=512, H=8, ff=1024, L=8, C=26, cls=0, prec=0)
    assert lib.hig_eval_encoder_workspace_bytes(C.byref(bad)) < 0           # a pair needs the init-pose row + a frame
    assert lib.hig_eval_encoder_fwd(C.byref(e), None, None, None, None, None, None, None, None) < 0
    t = _lib.TextDims(B=2, N=77, W=512, Lt=256, H=4, ff=2048, L=4, E=2048, prec=0)
    assert lib.hig_text_head_workspace_bytes(C.byref(t), 1) > lib.hig_text_head_workspace_bytes(C.byref(t), 0) > 0


def test_no_cpu_fallback_in_product_path():
    c = fill.CASES["tiny"]
    m = hig_amd.MotionTransformer(c["F"], num_frames=c["num_frames"], latent_dim=c["d"], ff_size=c["ff"],
                                  num_layers=c["L"], num_heads=c["H"], text_latent_dim=c["Lt"])
    inp = fill.inputs(c["B"], c["T"], c["F"], c["d"], c["N"], c["Lt"], c["lengths"], c["t"])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(inp["x"], inp["t"], length=inp["length"], xf_proj=inp["xf_proj"], xf_out=inp["xf_out"])
    with pytest.raises(RuntimeError, match="no CPU"):
        hig_amd.models.transformer.timestep_embedding(inp["t"], 64)
    # the product package never imports the oracle
    for root, _, files in os.walk(os.path.join(ROOT, "human-interaction-generation_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("the oracle", "").replace("oracle/", ""), f


def test_state_dict_contract_and_zero_init(gold):
    g = gold("g7_state_dict_keys.npz")
    c = fill.CASES["tiny"]
    m = hig_amd.MotionTransformer(c["F"], num_frames=c["num_frames"], latent_dim=c["d"], ff_size=c["ff"],
                                  num_layers=c["L"], num_heads=c["H"], text_latent_dim=c["Lt"])
    sd = m.state_dict()
    assert set(sd) == set(g.files)
    assert all(tuple(sd[k].shape) == tuple(g[k]) for k in g.files)
    zero = [k for k in sd if k.startswith("out.") or (k.startswith("temporal_decoder_blocks.") and
                                                      (".ffn.linear2." in k or ".out_layers.2." in k))]
    assert len(zero) == 2 + c["L"] * (2 + 6) and all(sd[k].abs().sum() == 0 for k in zero)
    assert m.time_embed_dim == 4 * c["d"] and m.input_feats == c["F"] and m.num_frames == c["num_frames"]
    assert not any(p.requires_grad for p in m.clip.parameters())          # frozen CLIP
    msk = m.generate_src_mask(5, torch.tensor([5, 2, 0]))
    assert msk.tolist() == [[1, 1, 1, 1, 1], [1, 1, 0, 0, 0], [0, 0, 0, 0, 0]]


def test_flat_parameter_layout_fuses_qkv_kv_and_stylization():
    c = fill.CASES["tiny"]
    m = hig_amd.MotionTransformer(c["F"], num_frames=c["num_frames"], latent_dim=c["d"], ff_size=c["ff"],
                                  num_layers=c["L"], num_heads=c["H"], text_latent_dim=c["Lt"])
    before = {k: v.clone() for k, v in m.state_dict().items()}
    fp = hig_amd.models.transformer._FlatParams(m)
    assert all(torch.equal(before[k], v) for k, v in m.state_dict().items())   # values preserved
    assert len(fp.group_offsets) == _lib.NGLOBAL + c["L"] * _lib.NLAYER
    sa = m.temporal_decoder_blocks[1].sa_block
    d = c["d"]
    assert sa.key.weight.data_ptr() == sa.query.weight.data_ptr() + 4 * d * d
    assert sa.value.weight.data_ptr() == sa.query.weight.data_ptr() + 8 * d * d
    assert sa.value.bias.data_ptr() == sa.query.bias.data_ptr() + 8 * d
    ca = m.temporal_decoder_blocks[0].ca_block
    assert ca.value.weight.data_ptr() == ca.key.weight.data_ptr() + 4 * d * c["Lt"]
    stys = [s for b in m.temporal_decoder_blocks for s in (b.sa_block.proj_out, b.ca_block.proj_out, b.ffn.proj_out)]
    base = stys[0].emb_layers[1].weight.data_ptr()
    for i, s in enumerate(stys):
        assert s.emb_layers[1].weight.data_ptr() == base + 4 * i * 2 * d * 4 * d
    assert all(off % 64 == 0 for off in fp.group_offsets if off is not None)    # 256-byte aligned groups
    assert sum(off is None for off in fp.group_offsets) == 4 + 8 * c["L"]          # two-person entries: NULL
    # in-place updates through either view are visible through the other
    with torch.no_grad():
        fp.flat.zero_()
    assert sum(p.abs().sum().item() for p in fp.params) == 0


def test_schedule_tables_match_reference_golden(gold):
    g = gold("g1_schedule.npz")
    for n in (1000, 50):
        gd = hig_amd.GaussianDiffusion(betas=gdm.get_named_beta_schedule("linear", n),
                                       model_mean_type=gdm.ModelMeanType.EPSILON,
                                       model_var_type=gdm.ModelVarType.FIXED_SMALL, loss_type=gdm.LossType.MSE)
        assert gd.num_timesteps == n
        for k in g.files:
            if k.startswith("n%d." % n):
                assert np.array_equal(getattr(gd, k.split(".", 1)[1]), g[k]), k
    s = gdm.create_named_schedule_sampler("uniform", gd)
    np.random.seed(0)
    t, w = s.sample(8, "cpu")
    assert t.dtype == torch.int64 and t.min() >= 0 and t.max() < 50 and torch.all(w == 1)
    with pytest.raises(NotImplementedError):
        gd.ddim_sample()


def test_diffusion_parametrisations_beyond_the_trainers_choice(gold):
    """Cosine schedule, FIXED_LARGE variance, START_X / PREVIOUS_X mean types, q_mean_variance and the conversions between
    eps / x0 / x_{t-1} (gaussian_diffusion.py:247-274,382-397,488-560,1041-1047): plain tensor arithmetic, held to the
    reference's outputs on the same inputs (golden G15, oracle/make_golden.py)."""
    g = gold("g15_diffusion_variants.npz")
    for n in (50, 1000):
        assert np.array_equal(gdm.get_named_beta_schedule("cosine", n), g["cosine_betas_%d" % n])
    x, mo, noise, t = (torch.from_numpy(g[k]) for k in ("x", "model_out", "noise", "t"))
    for mean in ("EPSILON", "START_X", "PREVIOUS_X"):
        for var in ("FIXED_SMALL", "FIXED_LARGE"):
            gd = hig_amd.GaussianDiffusion(betas=gdm.get_named_beta_schedule("cosine", 50), model_mean_type=gdm.ModelMeanType[mean],
                                           model_var_type=gdm.ModelVarType[var], loss_type=gdm.LossType.MSE)
            for clip in (False, True):
                pmv = gd.p_mean_variance(lambda *_a, **_k: mo, x, t, clip_denoised=clip)
                for k, v in pmv.items():
                    assert np.array_equal(v.numpy(), g["%s.%s.clip%d.%s" % (mean, var, int(clip), k)]), (mean, var, clip, k)
            tl = gd.training_losses(lambda *_a, **_k: mo, x, t, noise=noise)
            assert np.array_equal(tl["mse"].numpy(), g["%s.%s.tl_mse" % (mean, var)])
            assert np.array_equal(tl["target"].numpy(), g["%s.%s.tl_target" % (mean, var)])
    for v, k in zip(gd.q_mean_variance(x, t), ("q_mean", "q_variance", "q_log_variance")):
        assert np.array_equal(v.numpy(), g[k])
    assert np.array_equal(gd._predict_eps_from_xstart(x, t, mo).numpy(), g["eps_from_xstart"])
    assert np.array_equal(gd._predict_xstart_from_xprev(x, t, mo).numpy(), g["xstart_from_xprev"])
    # what stays out of scope says so
    with pytest.raises(NotImplementedError):
        hig_amd.GaussianDiffusion(betas=gdm.get_named_beta_schedule("linear", 10), model_mean_type=gdm.ModelMeanType.EPSILON,
                                  model_var_type=gdm.ModelVarType.LEARNED_RANGE, loss_type=gdm.LossType.MSE)
    with pytest.raises(NotImplementedError):
        hig_amd.GaussianDiffusion(betas=gdm.get_named_beta_schedule("linear", 10), model_mean_type=gdm.ModelMeanType.EPSILON,
                                  model_var_type=gdm.ModelVarType.FIXED_SMALL, loss_type=gdm.LossType.KL)


def test_generic_diffusion_plumbing_on_host_tensors(gold):
    """q_sample / p_sample / training_losses over a stub model: the tensor-op branch used for
    anything that is not (fp32, ROCm) follows the reference arithmetic (golden G4)."""
    g = gold("g4_diffusion.npz")
    gd = hig_amd.GaussianDiffusion(betas=gdm.get_named_beta_schedule("linear", 1000),
                                   model_mean_type=gdm.ModelMeanType.EPSILON,
                                   model_var_type=gdm.ModelVarType.FIXED_SMALL, loss_type=gdm.LossType.MSE)
    x, eps, t = torch.tensor(g["x"]), torch.tensor(g["eps"]), torch.tensor(g["t"])
    z0 = fill.tensor_for("g4.z.0", x.shape) * 10.0
    assert torch.equal(gd.q_sample(x, t, noise=z0), torch.tensor(g["q_sample"]))
    pmv = gd.p_mean_variance(lambda *_a, **_k: eps, x, t, clip_denoised=False)
    assert torch.equal(pmv["mean"], torch.tensor(g["mean"]))
    assert torch.equal(pmv["pred_xstart"], torch.tensor(g["pred_xstart"]))
    assert torch.equal(pmv["log_variance"], torch.tensor(g["log_variance"]))
    tl = gd.training_losses(lambda *_a, **_k: eps, x, t, noise=fill.tensor_for("g4.noise", x.shape) * 10)
    assert np.allclose(tl["mse"].numpy(), g["tl_mse"], rtol=1e-6)
    assert set(tl) == {"mse", "target", "pred"}


def test_sharded_sampler_matches_reference_rule():
    """datasets/dataloader.py:16-53: epoch-seeded permutation, wrap-around pad, rank stride."""
    n, world = 10, 4
    g = torch.Generator()
    g.manual_seed(0)
    perm = torch.randperm(n, generator=g).tolist()
    total = -(-n // world) * world
    padded = (perm * (total // n + 1))[:total]
    seen = []
    for r in range(world):
        s = ShardedSampler(n, r, world, shuffle=True)
        assert list(s) == padded[r:total:world] and len(s) == total // world
        seen += list(s)
    assert set(seen) == set(range(n))
    assert list(ShardedSampler(5, 0, 1, shuffle=False)) == [0, 1, 2, 3, 4]


_DP_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from hig_amd.parallel import FlatGradAllReduce, broadcast_parameters
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[3])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.manual_seed(100 + rank)
lin = torch.nn.Linear(6, 3)
broadcast_parameters(lin)                                   # rank 0's init everywhere
flat_g = torch.arange(8, dtype=torch.float32) * (rank + 1)   # per-rank gradient
w = FlatGradAllReduce()(flat_g)
ok = w == world and torch.equal(flat_g, torch.arange(8, dtype=torch.float32) * sum(range(1, world + 1)))
mean = flat_g / w                                            # DDP semantics: mean over ranks
ps = [p.detach().clone() for p in lin.parameters()]
gathered = [torch.zeros_like(ps[0]) for _ in range(world)]
dist.all_gather(gathered, ps[0])
ok = ok and all(torch.equal(gathered[0], t) for t in gathered)
ok = ok and torch.equal(mean, torch.arange(8, dtype=torch.float32) * (sum(range(1, world + 1)) / world))
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 1)
"""


def test_data_parallel_exchange_world2_gloo(tmp_path):
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER % ROOT)
    port = str(29500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", port]) for r in range(2)]
    assert [p.wait(timeout=120) for p in procs] == [0, 0]


def test_flat_allreduce_is_noop_without_process_group():
    g = torch.ones(4)
    assert hig_amd.parallel.FlatGradAllReduce()(g) == 1 and torch.equal(g, torch.ones(4))


def test_launcher_spawns_ranks_runs_them_in_a_group_and_tears_it_down(tmp_path):
    """hig_amd.parallel.launch == tools/train.py:53-102 (setup -> body -> destroy_process_group, one process per
    rank): gloo here (no GPU), 2 ranks, 127.0.0.1 rendezvous on a free port."""
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import launch_helper
    from hig_amd import parallel
    parallel.launch(launch_helper.allreduce_rank, world_size=2, backend="gloo", args=(str(tmp_path),), kwargs={"scale": 2.0})
    assert not dist.is_initialized()                    # the parent never joins
    for r in range(2):
        assert open(tmp_path / ("rank%d.txt" % r)).read() == "gloo 2 6"
    # under torchrun's variables the caller IS a rank: joins, runs, destroys the group again
    env = {"RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(parallel._free_port())}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        parallel.launch(launch_helper.allreduce_rank, backend="gloo", args=(str(tmp_path),))
        assert open(tmp_path / "rank0.txt").read() == "gloo 1 1" and not dist.is_initialized()
        with pytest.raises(ValueError, match="boom"):
            parallel.launch(launch_helper.failing_rank, backend="gloo")
        assert not dist.is_initialized()                # destroyed also when the body raises
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)


def test_train_refuses_a_world_size_without_a_process_group():
    import types
    import hig_amd
    from hig_amd.trainers import ddpm_trainer as tr
    t = object.__new__(hig_amd.DDPMTrainer)
    t.opt = types.SimpleNamespace(lr=1e-4, model_dir="/tmp", is_continue=False, fused_step=False)
    t.device, t._fused = "cpu", None
    t.encoder = torch.nn.Linear(2, 2)
    t.to = lambda dev: None
    with pytest.raises(RuntimeError, match="initialised process group"):
        t.train([], 0, 2)
    assert tr._dist_world() == 1


def test_running_mean_prints_on_iteration_multiples_like_the_reference():
    from hig_amd.trainers.ddpm_trainer import _RunningMean
    m = _RunningMean(4)
    outs = [m.add({"l": 1.0}, it=it) for it in range(7, 13)]        # resumed at it = 6: the reference prints at 8 and 12
    assert [o is not None for o in outs] == [False, True, False, False, False, True]
    assert outs[1]["l"] == 2.0 / 4 and outs[5]["l"] == 1.0           # (sums divided by log_every, ddpm_trainer.py:250)
    assert m.add({"l": 1.0}, it=16, read=False) is None and m.n == 0 # non-printing ranks never read the values back


def test_host_entry_points_under_asan_ubsan():
    """SURVEY section 5 (sanitizers on the CPU build): the host side of every translation unit compiled with
    -fsanitize=address,undefined (no device code), driven through the entry points that return before any HIP call --
    workspace / scratch size queries over a sweep of shapes, and descriptor validation with null / misaligned / negative
    arguments (tests/host_sanitize/driver.cpp).  A sanitizer report aborts the driver."""
    csrc = os.path.join(ROOT, "human-interaction-generation_amd", "csrc")
    b = subprocess.run(["make", "-C", csrc, "-j8", "sanitize"], capture_output=True, text=True, timeout=900)
    assert b.returncode == 0, b.stdout[-3000:] + b.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([os.path.join(ROOT, "build", "host_sanitize", "driver")], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "host entry points clean" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


def test_no_kernel_of_the_built_library_spills_registers(tmp_path):
    """Every gfx950 kernel in libhig.so, by its code-object metadata: no VGPR spill (a spilled fragment in a matrix loop
    costs a vmcnt(0) there: the fused stylization block ran 2 % slower with one), and scratch memory only where a kernel keeps a
    dynamically indexed private array (the VALU `no_eff` backward for head dim 64, reachable through HIG_FULLATTN_VALU=1 only).
    (SGPR "spills" are scalar values parked in lanes of a vector register -- no memory traffic -- and are not counted.)"""
    import re
    tools = "/opt/rocm/lib/llvm/bin"
    if not all(os.path.exists(os.path.join(tools, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")):
        pytest.skip("LLVM object tools not found")
    fat = str(tmp_path / "fat.bin")
    subprocess.run([os.path.join(tools, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", _lib.LIB_PATH, fat], check=True)
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), data)]
    assert starts, "no offload bundle in libhig.so"
    kernels, offenders = 0, []
    for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(data)])):
        bundle, obj = str(tmp_path / ("b%d.bin" % n)), str(tmp_path / ("co%d.o" % n))
        open(bundle, "wb").write(data[a:b])
        subprocess.run([os.path.join(tools, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + bundle,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + obj], check=True, capture_output=True)
        notes = subprocess.run([os.path.join(tools, "llvm-readelf"), "--notes", obj], check=True, capture_output=True, text=True).stdout
        names = re.findall(r"\.name:\s+(\S+)", notes)
        vs, ss = re.findall(r"\.vgpr_spill_count:\s+(\d+)", notes), re.findall(r"\.sgpr_spill_count:\s+(\d+)", notes)
        scratch = re.findall(r"\.private_segment_fixed_size:\s+(\d+)", notes)
        assert len(names) == len(vs) == len(ss) == len(scratch)
        kernels += len(names)
        for nm, v, sg, sc in zip(names, vs, ss, scratch):
            if int(v) or (int(sc) and "full_bwd_q_kernelILi64E" not in nm):
                offenders.append((nm, int(v), int(sg), int(sc)))
    assert kernels > 300
    assert not offenders, offenders


def test_every_environment_switch_of_the_library_is_documented_and_flipped_by_a_test():
    """The switch surface stays closed: every getenv("HIG_...") in csrc/ is a row of DESIGN.md's "Switches" table and -- unless it is one
    of the diagnostic switches whose builds are wrong by construction -- an entry of tests/test_gpu_knobs.py."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    csrc = os.path.join(root, "human-interaction-generation_amd", "csrc")
    for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")):
        names |= set(re.findall(r'getenv\("(HIG_[A-Z0-9_]+)"\)', open(f).read()))
    assert len(names) >= 20
    design = open(os.path.join(root, "DESIGN.md")).read()
    table = design[design.index("**Switches.**"):]
    knobs = open(os.path.join(root, "tests", "test_gpu_knobs.py")).read()
    diagnostic = {"HIG_BF16_DBG", "HIG_BF16_WSP_DBG", "HIG_F32_WSP_DBG"}
    for n in sorted(names):
        assert "`%s`" % n in table, "%s is read by the library but missing from DESIGN.md's switch table" % n
        if n not in diagnostic:
            assert '("%s"' % n in knobs, "%s is not flipped by tests/test_gpu_knobs.py" % n
