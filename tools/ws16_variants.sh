cd $GRAFT_REPO_ROOT
for v in 8 4 2 44; do
  for sh in ffn1 qkv; do
    echo "=== NWJ=$v $sh"; HIG_BF16_WS_NWJ=$v python tools/gemm_ws16_stamps.py $sh 64 2>&1 | grep -v amdgpu.ids | tail -14
  done
done
echo "=== bench shapes default"; python tools/gemm16_bench.py 64 2>&1 | grep -v amdgpu.ids | tail -8
for v in 8 4 2 44; do echo "=== bench NWJ=$v"; HIG_BF16_WS_NWJ=$v python tools/gemm16_bench.py 64 2>&1 | grep -v amdgpu.ids | tail -8 | head -5; done
