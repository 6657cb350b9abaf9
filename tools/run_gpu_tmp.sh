cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; timeout 900 python tools/ubench/xcd_order_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_xcd_order.txt
