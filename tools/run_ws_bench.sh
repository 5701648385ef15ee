cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_bf16_storage.py tests/test_gpu_interaction.py -x -q -k "fused_apply or linear_attention or context or oracle or captured or bf16" 2>&1 | tail -5
python3 tools/apply16_stamps.py 32 2>&1 | grep -v amdgpu.ids | tail -9
python tools/fwd16_time.py 32 2>&1 | grep -v amdgpu.ids | tail -1
python tools/fwd16_time.py 64 2>&1 | grep -v amdgpu.ids | tail -1
python tools/fwd_cfg5_time.py 2>&1 | grep -v amdgpu.ids | tail -1
HIG_FUSE_APPLY=0 python tools/fwd_cfg5_time.py 2>&1 | grep -v amdgpu.ids | tail -1
