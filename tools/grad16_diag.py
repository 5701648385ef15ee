"""Diagnostic: per-tensor gradient norms / errors of the bf16-storage backward vs the fp32 HIP path and the CPU oracle."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import importlib.util
spec = importlib.util.spec_from_file_location("t16", os.path.join(sys.path[1], "tests", "test_gpu_bf16_training.py"))
t16 = importlib.util.module_from_spec(spec); spec.loader.exec_module(t16)
case = sys.argv[1] if len(sys.argv) > 1 else "width"
c = t16.CASES[case]
res = {}
for storage in ("f32", "bf16"):
    m = t16.build(c, storage=storage).train()
    res[storage] = t16.grads_vs_oracle(c, m)
r32, r16 = res["f32"], res["bf16"]
names = r16["names"]
rows = []
for k in names:
    g16, g32, gr = r16["named"][k].grad.double().cpu(), r32["named"][k].grad.double().cpu(), r16["p"][k].grad.double()
    rows.append((((g16 - gr).norm() / gr.norm().clamp_min(1e-30)).item(), k, g16.norm().item(), g32.norm().item(), gr.norm().item(),
                 ((g32 - gr).norm() / gr.norm().clamp_min(1e-30)).item()))
rows.sort(reverse=True)
print("case", case)
for e, k, n16, n32, nr, e32 in rows[:14]:
    print("%-62s err16 %.2e  |g16| %.3e |g32| %.3e |ref| %.3e  err32 %.1e" % (k, e, n16, n32, nr, e32))
