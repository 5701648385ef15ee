cd $GRAFT_REPO_ROOT
for pol in 0 1 2; do echo "=== HIG_WS16_STORE=$pol"; HIG_WS16_STORE=$pol python tools/gemm16_bench.py 64 2>&1 | grep -v amdgpu.ids | tail -8 | head -6; done
for pol in 0 1 2; do echo "=== fwd16 B=64 HIG_WS16_STORE=$pol"; HIG_WS16_STORE=$pol python tools/fwd16_time.py 64 bf16 2>&1 | grep -v amdgpu.ids | tail -3; done
for pol in 0 1; do echo "=== fwd16 B=32 HIG_WS16_STORE=$pol"; HIG_WS16_STORE=$pol python tools/fwd16_time.py 32 bf16 2>&1 | grep -v amdgpu.ids | tail -3; done
for pol in 0 1; do echo "=== train16 HIG_WS16_STORE=$pol"; HIG_WS16_STORE=$pol STORAGE=bf16 NO_CAPTURE=1 python tools/train16_time.py 2>&1 | grep -v amdgpu.ids | tail -1; done
