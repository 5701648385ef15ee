"""What the gradient exchange costs the training step on ONE MI355X, measured with the real RCCL backend at world size 1
(DESIGN section 6 marks these two numbers "unmeasured"): an all-reduce over one rank moves no bytes over
xGMI, but RCCL still enqueues its kernel on the communication stream per bucket, and that kernel co-runs with the backward's
GEMMs -- so (step with the forced exchange) - (step without) is the LAUNCH + CO-RUNNING cost of the 2 L + 2 bucket
all-reduces of the overlapped step (captured inside the step's hipGraph: `overlapped_exchange_cost_ms`; issued eagerly from
the backward's host hook: `eager_overlapped_exchange_cost_ms`), and of the single flat all-reduce between graph A and graph B
(`flat_exchange_cost_ms`), with the wire time excluded.
Also: one all-reduce of each bucket size alone (RCCL's fixed cost per collective at world 1).
Runs in ONE process: the process group is created before anything touches the GPU."""
import json, os, sys, time, types
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29631"), RANK="0", WORLD_SIZE="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("TORCH_NCCL_CUDA_EVENT_CACHE", "0")
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import bench, hig_amd
from hig_amd import parallel

c = dict(bench.CFG)
i = bench.make_inputs(c, dev, 0)
args = types.SimpleNamespace(device=dev, diffusion_steps=1000, is_train=True, lr=2e-4, batch_size=c["B"], num_epochs=1, log_every=50,
                             save_latest=500, save_every_e=5, is_continue=False, model_dir="/tmp")
noise = torch.randn_like(i["x0"])

def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

out = {"what": "one MI355X, RCCL ('nccl') process group of ONE rank; HIG_FORCE_EXCHANGE=1 makes the step issue every all-reduce it "
               "would issue on 8 GPUs (identity at world 1: launch + co-running cost only, no wire time)", "config": "B=64 T=196 d=512 L=8"}
def step_ms(storage, forced, form, **opt):
    """One trainer per measurement: `form` = "fused" (eager launches) or "captured" (hipGraph replay)."""
    os.environ["HIG_FORCE_EXCHANGE"] = forced
    m = bench.build_model(c, dev).train(); m.storage = storage
    a = types.SimpleNamespace(**vars(args)); vars(a).update(opt)
    tr = hig_amd.DDPMTrainer(a, m)
    assert parallel.exchange_active() == (forced == "1")
    step = tr.train_step_fused if form == "fused" else tr.train_step_captured
    ms = timed(lambda: step(i["x0"], i["t"], i["length"], i["xf_proj"], i["xf_out"], noise=noise))
    st = tr.fused_state()
    if form == "captured" and forced == "1":
        want = "one graph, exchange inside" if opt.get("capture_exchange", True) else "graph A | all-reduce | graph B"
        assert st.get("captured_form") == want, (st.get("captured_form"), st.get("capture_exchange_error"))
    del tr, m
    return round(ms, 3)

for storage in ("f32", "bf16"):
    r = {"fused_exchange_off_ms": step_ms(storage, "0", "fused"),
         "captured_exchange_off_ms": step_ms(storage, "0", "captured"),
         "fused_exchange_on_ms": step_ms(storage, "1", "fused"),                                   # 2 L + 2 all-reduces from the host hook
         "captured_exchange_in_graph_ms": step_ms(storage, "1", "captured"),                      # the same, captured with the backward
         "captured_exchange_split_ms": step_ms(storage, "1", "captured", capture_exchange=False),  # graph A | flat all-reduce | graph B
         "captured_exchange_in_graph_bf16_wire_ms": step_ms(storage, "1", "captured", grad_wire="bf16")}
    r["overlapped_exchange_cost_ms"] = round(r["captured_exchange_in_graph_ms"] - r["captured_exchange_off_ms"], 3)
    r["overlapped_exchange_cost_bf16_wire_ms"] = round(r["captured_exchange_in_graph_bf16_wire_ms"] - r["captured_exchange_off_ms"], 3)
    r["eager_overlapped_exchange_cost_ms"] = round(r["fused_exchange_on_ms"] - r["fused_exchange_off_ms"], 3)
    r["flat_exchange_cost_ms"] = round(r["captured_exchange_split_ms"] - r["captured_exchange_off_ms"], 3)
    out["train_step_" + storage] = r
os.environ["HIG_FORCE_EXCHANGE"] = "0"
# one collective of each bucket size, alone on the stream
sizes = {"layer_own_params_15.4MB": 15.4e6, "layer_sty_emb_rows_25.2MB": 25.2e6, "tail_28MB": 28e6, "flat_324MB": 324e6}
per = {}
for name, nbytes in sizes.items():
    buf = torch.zeros(int(nbytes // 4), device=dev)
    per[name] = round(timed(lambda: dist.all_reduce(buf), n=20) * 1e3, 1)
out["allreduce_alone_us"] = per
out["allreduce_alone_note"] = "dist.all_reduce of one fp32 buffer of that size, world 1, back to back (host launch + RCCL kernel)"
print(json.dumps(out))
dist.destroy_process_group()
