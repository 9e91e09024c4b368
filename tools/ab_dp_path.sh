#!/bin/bash
# What the N > 1 code path of bench.py costs by itself, on ONE rank over RCCL: plain / forced-DP, eager / graph.
#   bash tools/ab_dp_path.sh TAG   -> gpurun_out/<TAG>_ab_dp_path.txt
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/${TAG}_ab_dp_path.txt
: > $OUT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
B="--no-cpu-baseline --no-parity-gate --no-clock-probe --no-fp32x-line --no-host-inclusive --steps 20 --warmup 5"
run() { name=$1; shift; r=$(timeout 600 env "$@" python bench.py $B $EXTRA 2>>gpurun_out/${TAG}_ab_dp_path.err | grep '^{' | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['config'].get('mask_mode'))"); echo "$name $r" | tee -a $OUT; }
for rnd in 1 2; do
  EXTRA="" run plain_eager MU_X=0
  EXTRA="" run forcedp_eager_resample MU_BENCH_FORCE_DP=1
  EXTRA="--mask-mode fixed" run forcedp_eager_fixed MU_BENCH_FORCE_DP=1
  EXTRA="--graph" run plain_graph MU_X=0
  EXTRA="--graph" run forcedp_graph_resample MU_BENCH_FORCE_DP=1
  EXTRA="--graph --mask-mode fixed" run forcedp_graph_fixed MU_BENCH_FORCE_DP=1
done
