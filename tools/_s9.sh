cd $GRAFT_REPO_ROOT
timeout 900 python tools/stress_determinism.py 20 --fp32x > gpurun_out/r06i_stress_fp32x.log 2>&1; tail -3 gpurun_out/r06i_stress_fp32x.log
timeout 900 python tools/train_soak.py --fp32x > gpurun_out/r06i_soak_fp32x.log 2>&1; tail -6 gpurun_out/r06i_soak_fp32x.log
timeout 600 python -m pytest tests/test_gpu_dp.py -m gpu -q -k "multi_rank_path" --timeout 500 -p no:cacheprovider 2>&1 | tail -3
