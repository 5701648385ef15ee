cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bf16_storage.py -x -q -k "folded or oracle or captured or gemm_ws16 or updates" 2>&1 | tail -8
for B in 32 64; do
  for f in 0 1; do HIG_LNFOLD=$f python tools/fwd16_time.py $B 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/B=$B lnfold=$f /"; done
done
