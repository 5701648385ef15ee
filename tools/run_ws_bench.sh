cd $GRAFT_REPO_ROOT
P=human-interaction-generation_amd
for v in old new old new; do
  cp $P/libhig_$v.so $P/libhig.so
  echo "== $v"
  python tools/attn_quick.py 64 2>&1 | grep -v amdgpu.ids | tail -1
  python tools/attn_quick.py 32 2>&1 | grep -v amdgpu.ids | tail -1
  python tools/fwd_time.py 2>&1 | grep -v amdgpu.ids | tail -2
done
cp $P/libhig_new.so $P/libhig.so
