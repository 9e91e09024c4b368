cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q --timeout 1500 -p no:cacheprovider -x > gpurun_out/r03g_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r03g_pytest.log
timeout 1200 python tests/ab_step.py pre cur > gpurun_out/r03g_ab.txt 2>&1; cat gpurun_out/r03g_ab.txt
