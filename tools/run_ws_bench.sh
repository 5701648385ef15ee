cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1700 python -m pytest tests/test_gpu_bf16_storage.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -4
for B in 32 64; do
  HIG_BF16_WS=0 python tools/fwd16_time.py $B 2>&1 | grep -v amdgpu.ids | tail -1
  python tools/fwd16_time.py $B 2>&1 | grep -v amdgpu.ids | tail -1
done
