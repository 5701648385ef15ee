cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_bf16_storage.py -x -q -k "context or oracle or captured or fused_apply" 2>&1 | tail -3
for c in 1 0 1 0; do HIG_CTX16=$c python tools/fwd_cfg5_time.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/ctx16=$c /"; done
python tools/fwd16_time.py 32 2>&1 | grep -v amdgpu.ids | tail -1
python tools/fwd16_time.py 64 2>&1 | grep -v amdgpu.ids | tail -1
