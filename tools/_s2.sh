cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -k "fp32x" --timeout 900 -p no:cacheprovider > gpurun_out/r06b_fp32x.log 2>&1; tail -30 gpurun_out/r06b_fp32x.log
