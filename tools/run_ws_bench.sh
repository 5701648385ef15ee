cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bf16_storage.py -x -q -k "gemm_ws16" 2>&1 | tail -3
HIG_BF16_WS_NWJ=44 python tools/gemm_ws16_stamps.py ffn1 64 2>&1 | tail -14
for B in 32 64; do
  python tools/gemm16_bench.py $B 2>&1 | grep -v emb_ss | grep -v te2 | grep -v amdgpu.ids
  HIG_BF16_WS_NWJ=44 python tools/gemm16_bench.py $B 2>&1 | grep -v emb_ss | grep -v te2 | grep -v amdgpu.ids
done
