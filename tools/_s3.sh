cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q --timeout 1500 -p no:cacheprovider > gpurun_out/r06c_pytest.log 2>&1; tail -8 gpurun_out/r06c_pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-host-inclusive --dtype fp32x --steps 10 --warmup 3 > gpurun_out/r06c_bench_fp32x.json 2> gpurun_out/r06c_bench_fp32x.err; cut -c1-400 gpurun_out/r06c_bench_fp32x.json
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06c_prof -o a -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity-gate --no-clock-probe --no-fp32x-line --no-host-inclusive --dtype fp32x --steps 8 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/r06c_prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r06c_prof -name '*kernel_stats.csv' | head -1)
python tools/prof_summary.py $f 10 40 > gpurun_out/r06c_prof_summary.txt; cp $f gpurun_out/r06c_kernel_stats.csv; rm -rf gpurun_out/r06c_prof
head -50 gpurun_out/r06c_prof_summary.txt
