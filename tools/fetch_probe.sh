#!/bin/bash
# FETCH_SIZE / duration of the attention kernels under tools/bench_attn.py for library variants (debug aid):
#   bash tools/fetch_probe.sh TAG [variant ...]      ("intree" = the in-tree library)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = intree ]; then unset MU_LIB_PATH; else export MU_LIB_PATH=$ROOT/gpurun_variants/libmu_$v.so; fi
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_$v -o a -- python3 $ROOT/tools/bench_attn.py > $OUT/${TAG}_$v.log 2>&1
  echo "$v rc=$?"
  cd $ROOT
  python3 - $OUT/${TAG}_$v/a_counter_collection.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if "attn" not in r["Kernel_Name"]: continue
    a = acc[r["Kernel_Name"][:40]]
    a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k, (n, f, t) in acc.items():
    print(f"  {k:40s} calls {n:3d}  HBM read {2 * f * 1024 / n / 1e9:.3f} GB/launch (2 x FETCH_SIZE KB)  {t / n:.3f} ms/launch")
PY
done
