#!/bin/bash
# roofline.traffic of the bench line per workload shape: FETCH_SIZE / WRITE_SIZE (separate rocprofv3 --pmc passes, counters only with
# --kernel-trace, every profiler command under `timeout`) of the dominant kernel -- the dK/dV sweep of self_attention6 -- inside the
# bench step.  Usage (repo root on the GPU box):  bash tools/dkv_traffic.sh TAG KEY [bench.py args ...]
#   -> gpurun_out/<TAG>_dkv_traffic_<KEY>.json   (KEY = bench.py's traffic key: b{batch}_c{c_out}_hw{hw}_{dtype}[_3head])
TAG=$1; KEY=$2; shift 2
export MU_SESSION_TAG=$TAG
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_traf_${KEY}_$c -o a -- python3 $ROOT/bench.py --no-cpu-baseline --no-parity-gate --no-clock-probe --no-fp32x-line --no-host-inclusive --steps 2 --warmup 1 "$@" > $OUT/${TAG}_traf_${KEY}_$c.log 2>&1
  echo "$KEY $c rc=$?"
done
cd $ROOT
python3 - $OUT/${TAG}_traf_${KEY}_FETCH_SIZE $OUT/${TAG}_traf_${KEY}_WRITE_SIZE $OUT/${TAG}_dkv_traffic_${KEY}.json "$KEY" "$*" <<'PY'
import csv, glob, json, os, sys
def launches(d, counter):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "attn_bwd_dkv3" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    if not rows:
        return None, 0, None
    grid = max(int(r.get("Grid_Size", 0) or 0) for r in rows)
    big = [r for r in rows if int(r.get("Grid_Size", 0) or 0) == grid]
    return sum(float(r["Counter_Value"]) for r in big) / len(big), len(big), big[0]["Kernel_Name"]
f, nf, name = launches(sys.argv[1], "FETCH_SIZE")
w, nw, _ = launches(sys.argv[2], "WRITE_SIZE")
if f is None or w is None:
    print("no dK/dV launches found"); sys.exit(1)
rec = {"kernel": name, "workload_key": sys.argv[4], "bench_args": sys.argv[5], "session": os.environ.get("MU_SESSION_TAG", ""),
       "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --no-cpu-baseline --no-parity-gate --no-clock-probe --no-fp32x-line --no-host-inclusive --steps 2 --warmup 1 "
              + sys.argv[5] + f" (tools/dkv_traffic.sh); mean over the {nf} largest-grid launches of the kernel (self_attention6)",
       "FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
       "correction": "gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) streaming reads (MI355X_MICROARCH.md, HBM section): bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
       "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
json.dump(rec, open(sys.argv[3], "w"), indent=1)
print(json.dumps(rec))
PY
