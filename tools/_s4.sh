cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -k "fp32x or conv" --timeout 900 -p no:cacheprovider > gpurun_out/r06d_fp32x.log 2>&1; tail -4 gpurun_out/r06d_fp32x.log
timeout 600 python bench.py --no-cpu-baseline --no-host-inclusive --no-parity-gate --dtype fp32x --steps 10 --warmup 3 > gpurun_out/r06d_bench_fp32x.json 2> gpurun_out/r06d_bench_fp32x.err; cut -c1-300 gpurun_out/r06d_bench_fp32x.json
