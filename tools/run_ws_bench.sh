cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_interaction.py -x -q 2>&1 | tail -3
for f in 2 0 2 0; do HIG_FUSE_APPLY=$f python tools/two_person16_time.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/fuse=$f /"; done
