"""The captured bf16-storage sampling loop at B = 32, 200 steps: which launches does ONE step consist of?  (run under rocprofv3
--kernel-trace --stats; calls / 200 = launches per step)"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, hig_amd
from hig_amd.models import gaussian_diffusion as gdm
dev = torch.device("cuda", 0)
c = dict(bench.CFG, B=32)
m = bench.build_model(c, dev).eval(); bench.set_mode(m, "bf16s"); m.cache_text_context = True
i = bench.make_inputs(c, dev, 0)
gd = hig_amd.GaussianDiffusion(betas=gdm.get_named_beta_schedule("linear", 200), model_mean_type=gdm.ModelMeanType.EPSILON,
                               model_var_type=gdm.ModelVarType.FIXED_SMALL, loss_type=gdm.LossType.MSE)
kw = {"xf_proj": i["xf_proj"], "xf_out": i["xf_out"], "length": i["length"]}
out = gd.p_sample_loop(m, (c["B"], c["T"], c["F"]), clip_denoised=False, model_kwargs=kw)
torch.cuda.synchronize()
print("finite", bool(torch.isfinite(out).all()))
