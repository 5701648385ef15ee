cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_bf16_storage.py tests/test_gpu_interaction.py -x -q -k "layernorm or oracle or captured or bf16" 2>&1 | tail -3
python tools/fwd16_time.py 32 2>&1 | grep -v amdgpu.ids | tail -1
python tools/fwd16_time.py 512 2>&1 | grep -v amdgpu.ids | tail -1
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_fwd16 -o f -- python3 $GRAFT_REPO_ROOT/tools/fwd16_time.py 32 > $GRAFT_REPO_ROOT/gpurun_out/prof_fwd16.log 2>&1; rm -f $GRAFT_REPO_ROOT/gpurun_out/prof_fwd16/f_kernel_trace.csv
