#!/bin/bash
# One gpurun session: GPU test suite, the bench line, rocprofv3 kernel-trace stats and the PMC passes (separate passes, counters
# only with --kernel-trace, every profiler command under `timeout`).  Usage (from the repo root on the GPU box):
#     bash tools/gpu_session.sh TAG [tests|bench|prof|pmc ...]      (default: all four; MU_SESSION_FLAGS="--dtype fp32x" adds bench flags)
# Outputs land in gpurun_out/<TAG>_*.
TAG=${1:-r02}; shift
WHAT=${*:-tests bench prof pmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
for w in $WHAT; do
  case $w in
    tests)
      timeout 3000 python -m pytest tests -m gpu -q --timeout 1500 -p no:cacheprovider > $OUT/${TAG}_pytest.log 2>&1
      echo "pytest rc=$?"; tail -5 $OUT/${TAG}_pytest.log ;;
    bench)
      timeout 900 python bench.py $MU_SESSION_FLAGS --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
      echo "bench rc=$?"; cat $OUT/${TAG}_bench.json ;;
    prof)
      cd /tmp
      timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o a -- python3 $ROOT/bench.py --no-cpu-baseline --no-parity-gate --no-clock-probe --no-fp32x-line --no-host-inclusive $MU_SESSION_FLAGS --steps 8 --warmup 2 > $OUT/${TAG}_prof.log 2>&1
      echo "prof rc=$?"
      cd $ROOT
      f=$(find $OUT/${TAG}_prof -name '*kernel_stats.csv' | head -1)
      [ -n "$f" ] && python tools/prof_summary.py $f 10 24 > $OUT/${TAG}_prof_summary.txt && head -12 $OUT/${TAG}_prof_summary.txt ;;
    pmc)
      cd /tmp
      timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_sq -o a -- python3 $ROOT/bench.py --no-cpu-baseline --no-parity-gate --no-clock-probe --no-fp32x-line --no-host-inclusive $MU_SESSION_FLAGS --steps 2 --warmup 1 > $OUT/${TAG}_pmc_sq.log 2>&1
      echo "pmc sq rc=$?"
      timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o a -- python3 $ROOT/bench.py --no-cpu-baseline --no-parity-gate --no-clock-probe --no-fp32x-line --no-host-inclusive $MU_SESSION_FLAGS --steps 2 --warmup 1 > $OUT/${TAG}_pmc_fetch.log 2>&1
      echo "pmc fetch rc=$?"
      timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o a -- python3 $ROOT/bench.py --no-cpu-baseline --no-parity-gate --no-clock-probe --no-fp32x-line --no-host-inclusive $MU_SESSION_FLAGS --steps 2 --warmup 1 > $OUT/${TAG}_pmc_write.log 2>&1
      echo "pmc write rc=$?"
      cd $ROOT
      python tools/pmc_table.py $OUT/${TAG}_pmc_table.md 3 $OUT/${TAG}_pmc_sq $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write | head -30
      ;;
  esac
done
