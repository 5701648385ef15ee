#!/bin/bash
# Runs on the GPU box (through gpurun): bench line + rocprofv3 kernel trace of the same command +
# separate PMC passes for the dominant kernel (FFN linear1 GEMM).  Writes under gpurun_out/$1/.
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_under_rocprof.json 2> $O/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/tools/ffn_gemm_pmc.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/tools/ffn_gemm_pmc.py > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- python3 $R/tools/ffn_gemm_pmc.py > $O/pmc_sq.log 2>&1
cat $O/bench.json
# round 2: bf16-storage sampling step and the fp32 training step (kernel stats), bf16 FFN GEMM counters, HBM-bound kernels
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bf16 -- python3 $R/tools/fwd16_time.py 32 bf16 > $O/fwd16_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_train -- python3 $R/tools/train_step_time.py > $O/train_under_rocprof.log 2>&1
bash $R/tools/pmc_gemm16.sh $TAG/pmc16_ffn1 ffn1 64 > $O/pmc16.log 2>&1
bash $R/tools/pmc_hbm.sh $TAG/pmc_hbm > $O/pmc_hbm.log 2>&1
# round 3: s_memtime stamps of the weight-stationary bf16 GEMM (FFN linear1 at M = 12 544, cross-attention query at M = 6 272),
# of the fused apply + stylization kernel, and the per-shape timings of the bf16 GEMM
python3 $R/tools/gemm_ws16_stamps.py ffn1 64 2>&1 | grep -v amdgpu.ids | tail -13 > $O/ws16_stamps.txt
python3 $R/tools/gemm_ws16_stamps.py ca_q 32 2>&1 | grep -v amdgpu.ids | tail -10 >> $O/ws16_stamps.txt
python3 $R/tools/apply16_stamps.py 32 2>&1 | grep -v amdgpu.ids | tail -10 > $O/apply16_stamps.txt
python3 $R/tools/gemm16_bench.py 64 2>&1 | grep -v amdgpu.ids | tail -8 > $O/gemm16_shapes.txt
python3 $R/tools/gemm16_bench.py 32 2>&1 | grep -v amdgpu.ids | tail -8 >> $O/gemm16_shapes.txt
python3 $R/tools/train_step_time.py 2>&1 | grep -v amdgpu.ids | tail -3 > $O/train_step_eager_vs_captured.txt
HIG_BWD_OVERLAP=0 python3 $R/tools/train_step_time.py 2>&1 | grep -v amdgpu.ids | tail -2 | sed 's/^/HIG_BWD_OVERLAP=0 /' >> $O/train_step_eager_vs_captured.txt


# round 4: the bf16-storage training step (kernel stats of the eager step; eager vs captured; solo kernel timings), the
# RCCL world-1 exchange-cost probe, the store-policy / text-fork / LayerNorm-fold A/B lines
STORAGE=bf16 NO_CAPTURE=1 STEPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_train16 -- python3 $R/tools/train16_time.py > $O/train16_under_rocprof.log 2>&1
python3 $R/tools/train16_time.py 2>&1 | grep -v amdgpu.ids | tail -3 > $O/train16_time.txt
python3 $R/tools/train16_kernels_time.py 2>&1 | grep -v amdgpu.ids > $O/train16_kernels_solo.txt
python3 $R/tools/rccl_world1_probe.py 2>/dev/null | grep "^{" > $O/rccl_world1_probe.json
{ for f in 1 0; do HIG_LNFOLD32=$f python3 $R/tools/fwd_text_time.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/HIG_LNFOLD32=$f /"; done;
  for f in 1 0; do HIG_TEXT_FORK=$f python3 $R/tools/fwd_text_time.py 2>&1 | grep -v amdgpu.ids | tail -1; done; } > $O/fwd32_fold_textfork.txt
python3 $R/tools/fwd_cfg5_time.py 2>&1 | grep -v amdgpu.ids | tail -1 > $O/cfg5_time.txt

# round 5: micro-probes behind the specialised-wave kernels (L2 -> CU fetch, VALU next to MFMA, LDS read rate), stamps of
# gemm_wsp16 and of the sliced weight-gradient kernel, per-shape weight-gradient timings, the two-person bf16 step
for p in l2_fetch_probe coissue_probe lds_read_probe; do [ -x $R/tools/$p ] && $R/tools/$p > $O/$p.txt 2>&1; done
python3 $R/tools/gemm_wsp16_stamps.py ffn1 64 2>&1 | grep -v amdgpu.ids | tail -14 > $O/wsp16_stamps.txt
python3 $R/tools/gemm_wsp16_stamps.py qkv 64 2>&1 | grep -v amdgpu.ids | tail -14 >> $O/wsp16_stamps.txt
python3 $R/tools/wgrad_stamps.py 2>&1 | grep -v amdgpu.ids | tail -12 > $O/wgrad_stamps.txt
python3 $R/tools/wgrad_time.py 2>&1 | grep -v amdgpu.ids | tail -8 > $O/wgrad_time.txt
python3 $R/tools/two_person16_time.py 2>&1 | grep -v amdgpu.ids | tail -4 > $O/two_person16_time.txt

# round 6: the exact-fp32 weight-stationary kernels (gemm_wsp32 / wgrad_wsp32): per-shape A/B against the tiled kernel, stamps of the
# diagnostic instance, the micro-probes behind them, the steady-state counter pass, the capture soak
{ HIG_F32_WSP=0 python3 $R/tools/gemm32_bench.py; HIG_F32_WSP=1 python3 $R/tools/gemm32_bench.py; M=50176 WGRAD=0 python3 $R/tools/gemm32_bench.py; } 2>&1 | grep "WSP=" > $O/gemm32_shapes.txt
python3 $R/tools/gemm_wsp32_stamps.py none 64 hot 2>&1 | grep -v amdgpu.ids | grep -v "workgroup start" | tail -17 > $O/wsp32_stamps.txt
python3 $R/tools/gemm_wsp32_stamps.py ffn1 64 cold 2>&1 | grep -v amdgpu.ids | grep -v "workgroup start" | tail -17 >> $O/wsp32_stamps.txt
for p in coissue32_probe mfma32_stream_probe; do [ -x $R/tools/$p ] && $R/tools/$p > $O/$p.txt 2>&1; done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq_warm -- python3 $R/tools/ffn_gemm_pmc.py warm > $O/pmc_sq_warm.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/kt_warm -- python3 $R/tools/ffn_gemm_pmc.py warm > $O/kt_warm.log 2>&1
python3 $R/tools/rowkernels_time.py 2>&1 | grep -v amdgpu.ids > $O/rowkernels_time.txt
python3 $R/tools/capture_soak.py 50 2> $O/capture_soak.err | grep "^{" > $O/capture_soak.json
