cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -x --timeout 900 -p no:cacheprovider > gpurun_out/r03a_pytest_kernels.log 2>&1
echo "pytest kernels rc=$?"; tail -15 gpurun_out/r03a_pytest_kernels.log
timeout 900 python tests/ab_env.py links: nolinks:MU_GRAD_LINKS=0,MU_ATTN_FUSED_ADD=0 > gpurun_out/r03a_ab_links.txt 2>&1
cat gpurun_out/r03a_ab_links.txt
timeout 600 python tests/bench_layers.py 64 > gpurun_out/r03a_layers.md 2>&1
tail -30 gpurun_out/r03a_layers.md
