cd $GRAFT_REPO_ROOT
timeout 300 python tools/gemm_1x1_probe.py 2>&1 | grep -v amdgpu.ids
