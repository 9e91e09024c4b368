cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_kernels.py "tests/test_gpu_modules.py::test_unet_fp32_gradients_within_3x_of_the_references_own_fp32_noise" "tests/test_gpu_modules.py::test_eval_mode_batchnorm_folded_into_conv_epilogue" -q --timeout 1500 -p no:cacheprovider > gpurun_out/r03d_pytest.log 2>&1
echo "pytest rc=$?"; grep -n "noise-floor\|passed\|failed\|FAILED" gpurun_out/r03d_pytest.log | head -20
timeout 900 python tests/ab_step.py pre cur > gpurun_out/r03d_ab_fepi.txt 2>&1; cat gpurun_out/r03d_ab_fepi.txt
