cd $GRAFT_REPO_ROOT
P=human-interaction-generation_amd
for v in old new old new; do
  cp $P/libhig_$v.so $P/libhig.so
  echo "== $v"
  python tools/gemm16_bench.py 64 2>&1 | grep -v amdgpu.ids | grep -E "ffn1|sty_out|qkv" | tail -3
  python tools/fwd16_time.py 32 2>&1 | grep -v amdgpu.ids | tail -1
  python tools/fwd16_time.py 64 2>&1 | grep -v amdgpu.ids | tail -1
done
cp $P/libhig_new.so $P/libhig.so
