"""Evaluator feature extraction: oracle vs the reference's own MotionEncoder /
MotionConsistencyEvalModel outputs (golden g13), the host metrics vs the reference's utils/metrics.py
(golden g14), the state-dict contract, and -- on the MI355X -- hig_eval_encoder_fwd behind the mirrored
classes and EvaluatorModelWrapper."""
import types

import numpy as np
import pytest
import torch

import hig_amd
from hig_amd.datasets import EvaluatorModelWrapper, evaluate_fid, evaluate_matching_score
from hig_amd.utils import metrics as M
from oracle import eval_models_ref as R
from oracle import fill
from oracle.eval_models_ref import EVAL_CASES, eval_inputs


def oracle_params(kind, c):
    return {k: fill.tensor_for(k, v) for k, v in
            R.param_shapes(kind, c["F"], c["d"], c["ff"], c["L"], c["num_frames"]).items()}


def build(kind, c, **kw):
    cls = hig_amd.MotionEncoder if kind == "enc" else hig_amd.MotionConsistencyEvalModel
    m = cls(input_feats=c["F"], num_frames=c["num_frames"], latent_dim=c["d"], ff_size=c["ff"],
            num_layers=c["L"], num_heads=c["H"], **kw)
    m.load_state_dict(fill.fill_state_dict(m.state_dict()), strict=True)
    return m.eval()


# ---- CPU: oracle, metrics, host logic ----------------------------------------------------------

@pytest.mark.parametrize("case", list(EVAL_CASES))
def test_oracle_matches_reference_golden(gold, case):
    g = gold("g13_eval_models.npz")
    c = EVAL_CASES[case]
    x1, x2, length = eval_inputs(case, c)
    logits, feat = R.motion_encoder_forward(oracle_params("enc", c), x1, x2, length, c["H"])
    clogits = R.consistency_forward(oracle_params("con", c), x1, x2, length, c["H"])
    for got, name in ((logits, "enc.logits"), (feat, "enc.feature"), (clogits, "con.logits")):
        ref = g[case + "." + name]
        assert got.shape == ref.shape
        assert np.abs(got.numpy() - ref).max() < 1e-6, name


def test_state_dict_contract(gold):
    g = gold("g13_eval_models.npz")
    c = EVAL_CASES["tiny"]
    for kind in ("enc", "con"):
        ref = {k[len("keys.%s." % kind):]: tuple(v) for k, v in g.items() if k.startswith("keys.%s." % kind)}
        sd = build(kind, c).state_dict()
        assert list(sd.keys()) == list(ref.keys())                  # same names, same order as the reference module
        assert all(tuple(sd[k].shape) == ref[k] for k in ref)
        assert {k: tuple(v) for k, v in R.param_shapes(kind, c["F"], c["d"], c["ff"], c["L"], c["num_frames"]).items()} == ref


def test_src_masks_and_no_cpu_fallback():
    c = EVAL_CASES["tiny"]
    enc, con = build("enc", c), build("con", c)
    m = enc.generate_src_mask(5, [5, 2, 0])
    assert m.shape == (3, 10) and m.tolist()[1] == [1, 1, 0, 0, 0] * 2 and m.tolist()[2] == [0] * 10
    m = con.generate_src_mask(5, torch.tensor([5, 2]))
    assert m.shape == (2, 11) and m.tolist()[1] == [1] + [1, 1, 0, 0, 0] * 2
    x1, x2, length = eval_inputs("tiny", c)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        enc(x1, x2, length=length)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        con(x1, x2, length=length)


def test_metrics_match_reference_golden(gold):
    g = gold("g14_metrics.npz")
    a, b, mm = g["a"], g["b"], g["mm"]
    mu_a, cov_a = M.calculate_activation_statistics(a)
    mu_b, cov_b = M.calculate_activation_statistics(b)
    assert np.array_equal(mu_a, g["mu_a"]) and np.array_equal(cov_a, g["cov_a"])
    assert abs(M.calculate_frechet_distance(mu_a, cov_a, mu_b, cov_b) - float(g["fid_ab"])) < 1e-9
    assert abs(M.calculate_frechet_distance(mu_a, cov_a, mu_a, cov_a) - float(g["fid_aa"])) < 1e-9
    np.random.seed(7)
    assert M.calculate_diversity(a, 100) == float(g["diversity"])
    np.random.seed(8)
    assert M.calculate_multimodality(mm, 5) == float(g["multimodality"])
    assert np.array_equal(M.euclidean_distance_matrix(a[:40], b[:40]), g["dist"], equal_nan=True)
    noisy = a[:40] + 2.0 * b[:40]
    assert np.array_equal(M.calculate_R_precision(a[:40], noisy, 3, sum_all=True), g["rprec"])
    assert np.array_equal(M.calculate_R_precision(a[:40], noisy, 3), g["rprec_mat"])
    assert M.calculate_matching_score(a[:40], b[:40], sum_all=True) == float(g["match"])
    assert M.calculate_matching_score(a[:40], b[:40]).shape == (40,)
    # size-independent properties: FID is symmetric, zero against itself, grows with a mean shift
    assert abs(M.calculate_frechet_distance(mu_b, cov_b, mu_a, cov_a) - float(g["fid_ab"])) < 1e-6
    assert M.calculate_frechet_distance(mu_a + 1.0, cov_a, mu_b, cov_b) > float(g["fid_ab"])
    with pytest.raises(AssertionError):
        M.calculate_diversity(a[:50], 50)


def test_metric_properties():
    """Size-independent properties: FID ignores sample order and any rotation applied to both sets, R-precision of a
    set against a slightly perturbed copy is perfect and cumulative in k, the distance matrix equals the direct norm."""
    rs = np.random.RandomState(3)
    a, b = rs.standard_normal((400, 12)), rs.standard_normal((350, 12)) * 0.7 + 0.3
    fid = M.calculate_frechet_distance(*M.calculate_activation_statistics(a), *M.calculate_activation_statistics(b))
    q, _ = np.linalg.qr(rs.standard_normal((12, 12)))
    fid_rot = M.calculate_frechet_distance(*M.calculate_activation_statistics(a @ q),
                                           *M.calculate_activation_statistics(b @ q))
    fid_perm = M.calculate_frechet_distance(*M.calculate_activation_statistics(a[rs.permutation(400)]),
                                            *M.calculate_activation_statistics(b[rs.permutation(350)]))
    assert abs(fid_rot - fid) < 1e-8 * max(1.0, fid) and abs(fid_perm - fid) < 1e-9 * max(1.0, fid)
    top = M.calculate_R_precision(a[:50], a[:50] + 1e-3, 3)         # (an exact copy gives sqrt(-0) = NaN self-distances,
    assert top.all() and top.shape == (50, 3)                       #  which the reference's expansion formula has too)
    far = M.calculate_R_precision(a[:50], b[:50], 3)
    assert (far[:, 1:] >= far[:, :-1]).all()                        # top-k hits are cumulative in k
    d = M.euclidean_distance_matrix(a[:30], b[:30])
    assert np.allclose(d, M.euclidean_distance_matrix(b[:30], a[:30]).T, atol=1e-9)
    assert np.allclose(d, np.linalg.norm(a[:30, None] - b[None, :30], axis=2), atol=1e-9)
    np.random.seed(0)
    same = np.repeat(a[:1], 20, axis=0)
    assert M.calculate_diversity(same, 5) == 0.0 and M.calculate_multimodality(same[None], 5) == 0.0


def test_evaluate_fid_host_logic():
    rs = np.random.RandomState(0)
    acts = {"ground truth": rs.standard_normal((200, 8)), "model": rs.standard_normal((200, 8)) + 0.5}
    fid = evaluate_fid(acts)
    assert list(fid) == ["ground truth", "model"] and abs(fid["ground truth"]) < 1e-6 and fid["model"] > 1.0


# ---- MI355X: the HIP path ----------------------------------------------------------------------

DEV = "cuda"


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(EVAL_CASES))
def test_hip_classifiers_match_reference_golden(gold, case):
    g = gold("g13_eval_models.npz")
    c = EVAL_CASES[case]
    x1, x2, length = eval_inputs(case, c)
    enc, con = build("enc", c).to(DEV), build("con", c).to(DEV)
    with torch.no_grad():
        logits, feat = enc(x1.to(DEV), x2.to(DEV), length=length)
        clogits = con(x1.to(DEV), x2.to(DEV), length=length.tolist())
    assert logits.shape == (c["B"], 26) and feat.shape == (c["B"], c["d"]) and clogits.shape == (c["B"], 2)
    assert rel(logits, g[case + ".enc.logits"]) < 1e-5
    assert rel(feat, g[case + ".enc.feature"]) < 1e-5
    assert rel(clogits, g[case + ".con.logits"]) < 1e-5


_WIDE_ORACLE = {}


@pytest.mark.gpu
@pytest.mark.parametrize("prec,tol", [("f32", 1e-5), ("bf16x3", 1e-4)])
def test_hip_classifiers_full_width_vs_oracle(prec, tol):
    """The evaluator's real shape: 91 tokens per person, 259 features, d=512, 8 heads, 8 layers; ragged lengths incl.
    a single valid token and a full sequence; padding invariance (features after `length` must not matter)."""
    c = dict(B=6, T=91, F=259, d=512, H=8, ff=1024, L=8, num_frames=196, length=[91, 90, 64, 65, 1, 17])
    x1, x2, length = eval_inputs("wide", c)
    enc, con = build("enc", c, precision=prec).to(DEV), build("con", c, precision=prec).to(DEV)
    if "wide" not in _WIDE_ORACLE:      # (the CPU oracle of the full-width classifiers: once for both product modes)
        _WIDE_ORACLE["wide"] = (R.motion_encoder_forward(oracle_params("enc", c), x1, x2, length, c["H"]),
                                R.consistency_forward(oracle_params("con", c), x1, x2, length, c["H"]))
    (ref_logits, ref_feat), ref_c = _WIDE_ORACLE["wide"]
    with torch.no_grad():
        logits, feat = enc(x1.to(DEV), x2.to(DEV), length=length.to(DEV))
        clogits = con(x1.to(DEV), x2.to(DEV), length=length)
        assert rel(logits, ref_logits) < tol and rel(feat, ref_feat) < tol and rel(clogits, ref_c) < tol
        if prec == "f32":
            y1, y2 = x1.clone(), x2.clone()
            for b, n in enumerate(c["length"]):
                y1[b, n:] = 7.0
                y2[b, n:] = -3.0
            l2, f2 = enc(y1.to(DEV), y2.to(DEV), length=length)
            assert torch.equal(l2, logits) and torch.equal(f2, feat)
            assert torch.equal(con(y1.to(DEV), y2.to(DEV), length=length), clogits)
            # swapping the persons swaps nothing in the pooled feature's definition except token order inside the
            # attention: the classifier is NOT symmetric (positions differ), but batch order must not matter
            perm = torch.tensor([3, 0, 5, 1, 4, 2])
            l3, f3 = enc(x1[perm].to(DEV), x2[perm].to(DEV), length=length[perm])
            assert rel(f3, feat[perm]) < 1e-6 and rel(l3, logits[perm]) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,length", [(1, 2, [5]), (1, 2, [1]), (70, 3, None), (2, 65, [64, 65])])
def test_hip_classifiers_edge_shapes(B, T, length):
    """One pair, the shortest sequence (init-pose row + one frame), lengths beyond T (the reference's mask loop is then
    empty: everything valid), more pairs than one attention chunk holds rows, a 64-key chunk boundary."""
    c = dict(B=B, T=T, F=15, d=64, H=8, ff=128, L=2, num_frames=80, length=length or [1 + (i % T) for i in range(B)])
    x1, x2, ln = eval_inputs("edge%d_%d" % (B, T), c)
    enc, con = build("enc", c).to(DEV), build("con", c).to(DEV)
    ref_logits, ref_feat = R.motion_encoder_forward(oracle_params("enc", c), x1, x2, ln, c["H"])
    ref_c = R.consistency_forward(oracle_params("con", c), x1, x2, ln, c["H"])
    with torch.no_grad():
        logits, feat = enc(x1.to(DEV), x2.to(DEV), length=ln)
        clogits = con(x1.to(DEV), x2.to(DEV), length=ln)
    assert rel(logits, ref_logits) < 1e-5 and rel(feat, ref_feat) < 1e-5 and rel(clogits, ref_c) < 1e-5
    with pytest.raises(AssertionError):
        enc(x1.to(DEV), x2[:, :, :-1].to(DEV), length=ln)               # shape mismatch is refused before any launch
    with pytest.raises(NotImplementedError):
        with torch.enable_grad():
            enc.train()(x1.to(DEV), x2.to(DEV), length=ln)              # inference only


@pytest.mark.gpu
def test_key_padding_attention_kernel():
    import ctypes as C
    from hig_amd import _lib
    torch.manual_seed(3)
    B, T, H, hd = 3, 150, 4, 32
    d = H * hd
    qkv = torch.randn(B, T, 3 * d, device=DEV)
    kpad = torch.zeros(B, T, dtype=torch.uint8, device=DEV)
    kpad[0, 100:] = 1
    kpad[1, 10:140] = 1                        # whole 64-key chunks of padded keys in the middle
    kpad[2, :70] = 1                           # the first chunk is all padding
    y = torch.empty(B, T, d, device=DEV)
    lse = torch.empty(B, H, T, device=DEV)
    _lib.check(_lib.lib().hig_fullattn_fwd_kpad(_lib.ptr(qkv), 3 * d, C.c_void_p(qkv.data_ptr() + 4 * d),
                                                C.c_void_p(qkv.data_ptr() + 8 * d), 3 * d, B, T, T, H, hd, None,
                                                _lib.ptr(kpad), _lib.ptr(y), d, _lib.ptr(lse), _lib.stream_ptr()))
    q, k, v = (t.view(B, T, H, hd).transpose(1, 2).double() for t in qkv.split(d, dim=-1))
    s = (q @ k.transpose(-1, -2)) / hd ** 0.5
    s = s.masked_fill(kpad.bool()[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, T, d)
    assert rel(y, ref) < 1e-6
    assert rel(lse, torch.logsumexp(s, -1)) < 1e-6


@pytest.mark.gpu
def test_evaluator_wrapper_and_scoring_passes():
    opt = types.SimpleNamespace(dataset_name="ntu_mul", device=torch.device(DEV), num_layers=2, latent_dim=64)
    wrapper_models = None
    # what build_models(opt, load=False) would make, filled deterministically (no checkpoints travel)
    from hig_amd.datasets.evaluator import build_models
    probe = types.SimpleNamespace(dim_pose=263, max_motion_length=196, num_layers=2, latent_dim=64)
    enc, con = build_models(probe, load=False)
    for m in (enc, con):
        m.load_state_dict(fill.fill_state_dict(m.state_dict()), strict=True)
    wrapper_models = (enc, con)
    w = EvaluatorModelWrapper(opt, models=wrapper_models)
    assert opt.dim_pose == 263 and opt.max_motion_length == 196 and not enc.training
    with pytest.raises(KeyError):
        EvaluatorModelWrapper(types.SimpleNamespace(dataset_name="other", device=DEV), models=wrapper_models)
    torch.manual_seed(5)
    B, T = 5, 91
    m1, m2 = torch.randn(B, T, 263), torch.randn(B, T, 263)
    lens = torch.tensor([91, 50, 91, 3, 77])
    fin, feat, cons = w.get_motion_embeddings(m1, m2, lens)
    pe = {k: v.detach().cpu() for k, v in enc.state_dict().items()}
    pc = {k: v.detach().cpu() for k, v in con.state_dict().items()}
    rl, rf = R.motion_encoder_forward(pe, m1[:, :, :-4], m2[:, :, :-4], lens, enc.num_heads)
    rc = R.consistency_forward(pc, m1[:, :, :-4], m2[:, :, :-4], lens, con.num_heads)
    assert rel(fin, rl) < 1e-5 and rel(feat, rf) < 1e-5 and rel(cons, rc) < 1e-5
    labels = rl.max(dim=1).indices.numpy()
    loaders = {"ground truth": [(labels, None, None, m1, m2, lens)],
               "model": [((labels + 1) % 26, None, None, m1[:3] * 1.5, m2[:3], lens[:3]),
                         (labels[3:], None, None, m1[3:], m2[3:] * 0.5, lens[3:])]}
    loaders["model"][0] = (((labels + 1) % 26)[:3],) + loaders["model"][0][1:]
    acc, act, act2, consistency = evaluate_matching_score(w, loaders)
    assert acc["ground truth"] == 1.0 and 0.0 <= acc["model"] <= 1.0
    assert act["ground truth"].shape == (B, 64) and act["model"].shape == (B, 64) and act2["model"].shape == (B, 26)
    assert rel(act["ground truth"], rf) < 1e-5
    want = float((rc.max(dim=1).indices == 0).float().mean())
    assert abs(consistency["ground truth"] - want) < 1e-9
