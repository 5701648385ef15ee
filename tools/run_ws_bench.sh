cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bf16_storage.py -x -q -k "all_epilogues or oracle or captured" 2>&1 | tail -5
for f in 1 0; do HIG_BF16_FEWROW=$f python tools/fwd16_time.py 32 2>&1 | grep -v amdgpu.ids | tail -3 | tr '\n' '|' | sed "s/^/fewrow=$f /"; echo; done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_fwd16 -o f -- python3 $GRAFT_REPO_ROOT/tools/fwd16_time.py 32 > $GRAFT_REPO_ROOT/gpurun_out/prof_fwd16.log 2>&1; rm -f $GRAFT_REPO_ROOT/gpurun_out/prof_fwd16/f_kernel_trace.csv
