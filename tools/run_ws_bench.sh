cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bf16_storage.py -x -q -k "fused_apply or oracle or captured" 2>&1 | tail -5
python3 tools/apply16_stamps.py 32 2>&1 | grep -v amdgpu.ids | tail -9
HIG_APPLY16_NW=4 python3 tools/apply16_stamps.py 32 2>&1 | grep -v amdgpu.ids | tail -9
for n in 8 4; do HIG_APPLY16_NW=$n python tools/fwd16_time.py 32 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/B=32 nw=$n /"; done
