cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd /tmp
for v in pre cur; do
  MU_LIB_PATH=$ROOT/gpurun_variants/libmu_$v.so timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r03f_prof_$v -o a -- python3 $ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 2 > $ROOT/gpurun_out/r03f_prof_$v.log 2>&1
  echo "prof $v rc=$?"
done
cd $ROOT
A=$(find gpurun_out/r03f_prof_pre -name '*kernel_stats.csv' | head -1); B=$(find gpurun_out/r03f_prof_cur -name '*kernel_stats.csv' | head -1)
python tests/prof_diff.py $A $B 12 40 | tee gpurun_out/r03f_prof_diff.txt
timeout 3000 python -m pytest tests -m gpu -q --timeout 1500 -p no:cacheprovider > gpurun_out/r03g_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r03g_pytest.log
