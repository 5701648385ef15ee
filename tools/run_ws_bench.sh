cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_bf16_storage.py -x -q -k "context or oracle or captured" 2>&1 | tail -3
python tools/fwd16_time.py 32 2>&1 | grep -v amdgpu.ids | tail -1
python tools/fwd16_time.py 64 2>&1 | grep -v amdgpu.ids | tail -1
python tools/fwd16_time.py 128 2>&1 | grep -v amdgpu.ids | tail -1
