#!/usr/bin/env python3
"""Soak of the captured gradient exchange (DDPMTrainer.train_step_captured with the per-layer RCCL all-reduces inside the step's
hipGraph): CYCLES times { drop every captured graph, warm up + capture again, replay twice } on ONE rank over the real RCCL
backend (HIG_FORCE_EXCHANGE=1: the all-reduces are issued in a group of one rank), fp32 and bf16 storage alternating, under
HIG_POISON=1, next to a second process that keeps the GPU and the host busy.  Round 5 saw the process group's watchdog thread
die once, on a loaded box, with "operation not permitted on an event last recorded in a capturing stream"; since round 6 no
Work handle of the exchange survives its call and the capture waits for the process group's own waitForPendingWorks
(hig_amd/parallel.py).  Every cycle must capture the exchange ("one graph, exchange inside") and the losses of the two storages
must repeat from cycle to cycle (same parameters restored, same inputs).
usage: python tools/capture_soak.py [cycles=50]          (prints one JSON line; exit code 1 on any failure)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BUSY = r"""
import sys, time, torch
a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
t0 = time.time()
while time.time() - t0 < float(sys.argv[1]):
    for _ in range(20): c = a @ b
    torch.cuda.synchronize()
    sum(i * i for i in range(20000))
"""


def main():
    cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    os.environ.setdefault("HIG_POISON", "1")
    os.environ["HIG_FORCE_EXCHANGE"] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # the busy neighbour is started BEFORE this process touches the GPU
    busy = subprocess.Popen([sys.executable, "-c", BUSY, str(20 + 6 * cycles)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    import torch
    import torch.distributed as dist
    from hig_amd import parallel
    import dp_worker

    out = {"cycles": cycles, "forms": {}, "errors": [], "losses_repeat": True}

    def body(rank, world):
        t0 = time.time()
        sets = {s: dp_worker.build_and_inputs(s) for s in ("f32", "bf16")}
        first = {}
        for cyc in range(cycles):
            for storage, (c, m, tr, gi, x0, noise) in sets.items():
                st = tr.fused_state()
                st["graphs"].clear()                     # force warm-up + capture again
                keep = (m.flat_params().flat.clone(), st["m"].clone(), st["v"].clone(), st["step"].clone())
                losses = [tr.train_step_captured(x0[k % 2], gi["t"], gi["length"], gi["xf_proj"], gi["xf_out"], noise=noise[k % 2]).item()
                          for k in range(3)]
                form = st.get("captured_form")
                out["forms"][form] = out["forms"].get(form, 0) + 1
                if form != "one graph, exchange inside":
                    out["errors"].append("cycle %d %s: %s (%s)" % (cyc, storage, form, st.get("capture_exchange_error")))
                if not all(l == l for l in losses):
                    out["errors"].append("cycle %d %s: non-finite loss %s" % (cyc, storage, losses))
                if storage in first and losses != first[storage]:
                    out["losses_repeat"] = False
                    out["errors"].append("cycle %d %s: losses %s != first cycle's %s" % (cyc, storage, losses, first[storage]))
                first.setdefault(storage, losses)
                with torch.no_grad():                    # back to the start: every cycle captures the same step
                    m.flat_params().flat.copy_(keep[0]); st["m"].copy_(keep[1]); st["v"].copy_(keep[2]); st["step"].copy_(keep[3])
                m.params_changed()
                if getattr(m, "storage", "f32") == "bf16":
                    m.flat_params().shadow16(m._param_version())
        torch.cuda.synchronize()
        out["seconds"] = round(time.time() - t0, 1)
        out["neighbour_alive_at_end"] = busy.poll() is None

    os.environ["MASTER_PORT"] = str(parallel._free_port())
    try:
        parallel.run_distributed(body, 0, 1, "nccl")
    except Exception as e:   # noqa: BLE001
        out["errors"].append("%s: %s" % (type(e).__name__, e))
    finally:
        if busy.poll() is None:
            busy.kill()
        busy.wait()
    out["ok"] = not out["errors"]
    print(json.dumps(out))
    sys.exit(0 if out["ok"] else 1)


if __name__ == "__main__":
    main()
