#!/bin/bash
# BASELINE config 5 (T = 300, d = 1024, L = 12, head dim 128, bf16 storage): forward time under the attention kernels' tile /
# occupancy variants (environment knobs, read once per process: one process per setting), then the LDS / register footprint
# of every kernel of the default forward from a rocprofv3 kernel trace.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
run() { echo -n "$1: "; env $1 python3 $R/tools/fwd_cfg5_time.py 2>&1 | grep config-5 | sed 's/.*bf16s: //'; }
run "DEFAULT=1"
run "HIG_CTX16=0"            # context build: fp32-MFMA kernel with bf16 loads instead of the bf16-MFMA kernel
run "HIG_CTX16_NB2=1"        # context build: DMA ring of two chunks instead of three (less LDS)
run "HIG_FUSE_APPLY=0"       # apply + stylization front as two kernels (apply on the bf16 cores, LayerNorm row kernel)
run "HIG_FUSE_APPLY=1"       # fused, products on the fp32 matrix cores
run "HIG_FUSE_APPLY=2"       # fused, products on the bf16 matrix cores (default)
run "HIG_LNFOLD1024=0"       # LayerNorm kernels in front of q/k/v and the cross-attention query instead of the fold
run "HIG_FWD16_FORK=0"       # text side and embedding chain on the caller's stream
run "HIG_BF16_WIDE256=1"     # q/k/v on the 256 x 256 tiled kernel (implies no fold for it)
run "DEFAULT=1"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/cfg5_sweep -o t -- python3 $R/tools/fwd_cfg5_time.py > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/cfg5_sweep/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:88]
    a = agg.setdefault(k, [0, 0.0, r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["Workgroup_Size_X"], r["Grid_Size_X"], r["Grid_Size_Y"]])
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("%-90s %6s %9s %8s %5s %5s %5s %10s" % ("kernel", "calls", "avg us", "LDS B", "vgpr", "agpr", "wg", "grid"))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if a[1] / a[0] < 4.5 and a[0] < 20: continue
    print("%-90s %6d %9.1f %8s %5s %5s %5s %6sx%-4s" % (k, a[0], a[1] / a[0], a[2], a[3], a[4], a[5], a[6], a[7]))
PY
