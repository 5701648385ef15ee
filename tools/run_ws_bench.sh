cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for s in 0 1 0 1; do HIG_FWD_SPLIT=$s python tools/fwd_time.py 2>&1 | tail -1 | sed "s/^/split=$s /"; done
for s in 0 1; do HIG_FWD_SPLIT=$s python tools/train_step_time.py 2>&1 | tail -3 | sed "s/^/split=$s /"; done
timeout 1500 python -m pytest tests/test_gpu_denoiser.py tests/test_gpu_full_size.py tests/test_gpu_trainer_state.py -x -q 2>&1 | tail -5
