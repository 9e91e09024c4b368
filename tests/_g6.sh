cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -x --timeout 900 -p no:cacheprovider > gpurun_out/r03e_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r03e_pytest.log
timeout 600 python tests/bench_layers.py 64 > gpurun_out/r03e_layers.md 2>&1; head -4 gpurun_out/r03e_layers.md; tail -4 gpurun_out/r03e_layers.md
timeout 1200 python tests/ab_step.py pre cur curslp > gpurun_out/r03e_ab.txt 2>&1; cat gpurun_out/r03e_ab.txt
