cd $GRAFT_REPO_ROOT
for n in 0 8 4 2 44; do
  echo "== NWJ=$n"
  HIG_BF16_WS_NWJ=$n python tools/gemm16_bench.py 64 2>&1 | grep -v amdgpu.ids | grep -E "ffn1|sty_out|qkv|ffn2|ca_q" | tail -5
done
