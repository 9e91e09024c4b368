cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q --timeout 800 -p no:cacheprovider -k "bn or batchnorm or norm" 2>&1 | tail -3
timeout 600 python tests/ab_bench.py bn nopk pk 2>&1 | tail -12
