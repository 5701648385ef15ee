cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for q in 1 2 4 8; do echo "== DEBUG_HIP_FORCE_GRAPH_QUEUES=$q"; DEBUG_HIP_FORCE_GRAPH_QUEUES=$q python tools/train_step_time.py 2>&1 | grep -v amdgpu.ids | tail -1; done
for b in 1 8 64 1024; do echo "== DEBUG_HIP_GRAPH_BATCH_SIZE=$b"; DEBUG_HIP_GRAPH_BATCH_SIZE=$b python tools/train_step_time.py 2>&1 | grep -v amdgpu.ids | tail -1; done
echo "== AMD_LOG_LEVEL=4 grep"; AMD_LOG_LEVEL=4 python tools/train_step_time.py 2>&1 | grep -i "hipGraph\] Creat\|max_streams" | sort | uniq -c | head
