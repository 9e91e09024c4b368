cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_next.py "tests/test_gpu_modules.py::test_unet_fp32_gradients_within_3x_of_the_references_own_fp32_noise" "tests/test_gpu_modules.py::test_unet_b8_train_mode_vs_oracle" "tests/test_gpu_modules.py::test_kernels_are_bitwise_deterministic_at_bench_shapes" tests/test_gpu_dp.py -q --timeout 1500 -p no:cacheprovider > gpurun_out/r03b_pytest.log 2>&1
echo "pytest rc=$?"; tail -40 gpurun_out/r03b_pytest.log
( timeout 1200 python tests/stress_determinism.py 300 > gpurun_out/r03b_stress300.log 2>&1; echo "stress rc=$?" >> gpurun_out/r03b_stress300.log )
tail -3 gpurun_out/r03b_stress300.log
