#!/bin/bash
# Two ranks of bench.py on ONE GPU over gloo (the RCCL call pattern with a host-staged transport): rank 0 runs under
# rocprofv3 (kernel + memory-copy trace), so the timeline shows WHEN each gradient bucket starts moving relative to
# the backward kernels.  usage: tools/dp_overlap_trace.sh <tag>
set -u
TAG=${1:-dp_overlap}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29531 WORLD_SIZE=2 HIG_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
RANK=1 LOCAL_RANK=1 python3 $R/bench.py --gpus 2 --steps 3 --warmup 1 --no-extra --no-cpu-baseline > $O/rank1.log 2>&1 &
P1=$!
export RANK=0 LOCAL_RANK=0
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 $R/bench.py --gpus 2 --steps 3 --warmup 1 --no-extra --no-cpu-baseline > $O/rank0.json 2> $O/rank0.err
wait $P1
cat $O/rank0.json
python3 - <<PY
import csv, glob
k = sorted(glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True))[-1]
m = sorted(glob.glob("$O/trace/**/*memory_copy_trace.csv", recursive=True))
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(k))]
rows.sort()
# training steps: from q_sample_kernel to clip_adam_kernel
starts = [s for s, e, n in rows if "q_sample_kernel" in n]
ends = [e for s, e, n in rows if "clip_adam_kernel" in n]
copies = []
if m:
    for r in csv.DictReader(open(m[-1])):
        copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Kind", "")), int(float(r.get("Bytes", r.get("Size", 0)) or 0))))
copies.sort()
lines = ["# data-parallel step, 2 ranks on one GPU over gloo: rank-0 timeline (rocprofv3 kernel + memory-copy trace)", ""]
for i, (s0, e1) in enumerate(zip(starts, ends)):
    bw = [(s, e, n) for s, e, n in rows if s0 <= s <= e1]
    last_bwd = max(e for s, e, n in bw if "clip_adam" not in n and "sumsq" not in n)
    first_bwd = min((s for s, e, n in bw if "ln_bwd" in n or "apply_bwd" in n), default=s0)
    cs = [c for c in copies if s0 <= c[0] <= e1 and c[3] >= (1 << 20)]
    d2h = [c for c in cs if "DEVICE_TO_HOST" in c[2].upper() or "D2H" in c[2].upper()]
    during = [c for c in d2h if c[0] < last_bwd]
    lines.append("step %d: forward+backward kernels %.2f ms (backward starts at +%.2f ms); %d device-to-host bucket copies >= 1 MiB, "
                 "%d of them (%.0f MB of %.0f MB) START before the last backward kernel ends; first at +%.2f ms; optimizer at +%.2f ms"
                 % (i, (last_bwd - s0) / 1e6, (first_bwd - s0) / 1e6, len(d2h), len(during), sum(c[3] for c in during) / 1e6,
                    sum(c[3] for c in d2h) / 1e6, ((d2h[0][0] - s0) / 1e6 if d2h else -1), (e1 - s0) / 1e6))
open("$O/timeline.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
