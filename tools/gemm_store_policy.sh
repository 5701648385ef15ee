cd $GRAFT_REPO_ROOT
for rep in 1 2; do for pol in 0 1 2; do echo -n "HIG_GEMM_STORE=$pol  "; HIG_GEMM_STORE=$pol python tools/fwd_time.py 2>&1 | grep -v amdgpu.ids | tail -1; done; done
for pol in 0 1 2; do echo -n "HIG_GEMM_STORE=$pol HIG_FWD_SPLIT=0 "; HIG_FWD_SPLIT=0 HIG_GEMM_STORE=$pol python tools/fwd_time.py 2>&1 | grep -v amdgpu.ids | tail -1; done
for pol in 0 1; do echo -n "HIG_GEMM_STORE=$pol bf16x3 "; HIG_PREC=bf16x3 HIG_GEMM_STORE=$pol python tools/fwd_time.py 2>&1 | grep -v amdgpu.ids | tail -1; done
for pol in 0 1; do echo "HIG_GEMM_STORE=$pol train f32: "; HIG_GEMM_STORE=$pol STORAGE=f32 NO_CAPTURE=1 python tools/train16_time.py 2>&1 | grep -v amdgpu.ids | tail -1; done
for pol in 0 1; do echo "HIG_GEMM_STORE=$pol gemm_bench:"; HIG_GEMM_STORE=$pol python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tail -8; done
