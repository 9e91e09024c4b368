#!/bin/bash
# LDS bank-conflict share per kernel over one bench step (debug aid): bash tools/lds_conflicts.sh TAG [bench flags, e.g. --dtype fp32x]   (on the GPU box)
TAG=${1:-lds}
shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $ROOT/gpurun_out/${TAG}_lds -o a -- python3 $ROOT/bench.py --no-cpu-baseline --no-parity-gate --no-clock-probe --no-fp32x-line --no-host-inclusive --steps 2 --warmup 1 "$@" > $ROOT/gpurun_out/${TAG}_lds.log 2>&1
echo rc=$?
cd $ROOT
python3 - $ROOT/gpurun_out/${TAG}_lds/a_counter_collection.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
rows = []
for k, c in acc.items():
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    if c["SQ_LDS_IDX_ACTIVE"] <= 0: continue
    rows.append((cyc, k, c["SQ_LDS_IDX_ACTIVE"] / (cyc * 256), c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], n[k]))
print("| kernel | launches | LDS active share of CU cycles | bank-conflict share of LDS cycles |\n|---|---|---|---|")
for cyc, k, a, b, m in sorted(rows, reverse=True)[:28]:
    print(f"| `{k}` | {m} | {a:.3f} | {b:.3f} |")
PY
