cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bf16_storage.py tests/test_gpu_interaction.py -x -q -k "joint_embed or oracle or captured or bf16" 2>&1 | tail -5
for f in 1 2; do python tools/fwd16_time.py 32 2>&1 | grep -v amdgpu.ids | tail -3 | tr '\n' '|'; echo; done
