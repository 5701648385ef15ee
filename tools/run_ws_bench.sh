cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_interaction.py tests/test_gpu_bf16_storage.py -x -q -k "bf16 or oracle or captured or folded" 2>&1 | tail -3
for f in 1 0 1 0; do HIG_LNFOLD=$f python tools/two_person16_time.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/fold=$f /"; done
python tools/fwd16_time.py 32 2>&1 | grep -v amdgpu.ids | tail -1
