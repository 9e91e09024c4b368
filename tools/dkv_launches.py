#!/usr/bin/env python3
"""Per-launch durations of the roofline kernel (the dK/dV sweep of self_attention6 = the largest-grid launches of attn_bwd_dkv3_kernel)
from a rocprofv3 --kernel-trace run:  python tools/dkv_launches.py <dir with *kernel_trace.csv> [bench json]  """
import csv, glob, json, os, statistics, sys

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "attn_bwd_dkv3" in r["Kernel_Name"]]
if not rows:
    print("no dK/dV launches in the trace"); sys.exit(1)
def gsz(r):                                     # kernel-trace CSVs carry Grid_Size or Grid_Size_X/Y/Z depending on the rocprofv3 build
    if r.get("Grid_Size"):
        return int(r["Grid_Size"])
    n = 1
    for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"):
        n *= int(r.get(k) or 1)
    return n


grid = max(gsz(r) for r in rows)
big = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if gsz(r) == grid]
small = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if gsz(r) != grid]
name = next(r["Kernel_Name"] for r in rows if gsz(r) == grid)
print(f"{name[:80]}: largest-grid launches (self_attention6, N = 16384) in the kernel trace")
print(f"launches {len(big)}, mean {statistics.mean(big):.3f} ms, median {statistics.median(big):.3f} ms, min {min(big):.3f}, max {max(big):.3f}")
print("all: " + " ".join(f"{t:.3f}" for t in big))
if small:
    print(f"the same kernel's other launches (the smaller C = 64 block): {len(small)}, mean {statistics.mean(small):.3f} ms")
if len(sys.argv) > 2 and os.path.exists(sys.argv[2]):
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    r = d["roofline"]
    print(f"bench.py's roofline.ms_per_launch (HIP events, un-profiled run in the same session): {r['ms_per_launch']} ms -> achieved {r['achieved']} "
          f"TF/s executed = {r['frac']} of {r['peak']} TF/s ({r['frac_at_measured_clock']} at the clock the chip held in the MFMA probe)")
