cd $GRAFT_REPO_ROOT
timeout 2500 python -m pytest tests/test_gpu_bf16_storage.py tests/test_gpu_full_size.py -x -q -k "one_kernel or fused_into or oracle or captured or sampling" 2>&1 | tail -4
for f in 1 0 1 0; do HIG_FUSE_OUT=$f python tools/fwd16_time.py 32 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/B=32 fuse_out=$f /"; done
