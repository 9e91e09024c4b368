#!/usr/bin/env python3
"""Where a step's wall time goes between its kernels, from a rocprofv3 --kernel-trace run (one stream: launches do not overlap):
    python tools/step_gaps.py <dir with *kernel_trace.csv> [steps]
Steps are cut at the largest-grid launch of the dK/dV sweep (once per step).  Per step: launches, sum of kernel durations, idle time
between kernels, and the launches below 12 us with their share.  Context figure for the small-batch lines (B = 8 / 16)."""
import csv, glob, os, statistics, sys

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
if not rows:
    print("no kernel trace"); sys.exit(1)
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 6


def gsz(r):
    if r.get("Grid_Size"):
        return int(r["Grid_Size"])
    n = 1
    for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"):
        n *= int(r.get(k) or 1)
    return n


ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], gsz(r)) for r in rows), key=lambda e: e[0])
dkv = [e for e in ev if "attn_bwd_dkv3" in e[2]]
grid = max(e[3] for e in dkv)
marks = [i for i, e in enumerate(ev) if "attn_bwd_dkv3" in e[2] and e[3] == grid]
marks = marks[-(nsteps + 1):]
print(f"{len(marks) - 1} steps cut at the self_attention6 dK/dV launch; times in ms per step (median over the steps)")
wall, busy, idle, n, small_n, small_t, gaps_all = [], [], [], [], [], [], []
for a, b in zip(marks[:-1], marks[1:]):
    seg = ev[a:b + 1]
    w = (seg[-1][0] - seg[0][0]) / 1e6
    bz = sum(e[1] - e[0] for e in seg[:-1]) / 1e6
    gaps = [max(0, seg[i + 1][0] - seg[i][1]) / 1e3 for i in range(len(seg) - 1)]      # us
    wall.append(w); busy.append(bz); idle.append(sum(gaps) / 1e3); n.append(len(seg) - 1)
    sm = [e for e in seg[:-1] if (e[1] - e[0]) < 12000]
    small_n.append(len(sm)); small_t.append(sum(e[1] - e[0] for e in sm) / 1e6)
    gaps_all += gaps
med = statistics.median
print(f"wall {med(wall):.3f}  kernels {med(busy):.3f}  idle between kernels {med(idle):.3f}  launches {int(med(n))}")
print(f"launches < 12 us: {int(med(small_n))}, {med(small_t):.3f} ms of kernel time")
gaps_all.sort()
q = lambda p: gaps_all[min(len(gaps_all) - 1, int(p * len(gaps_all)))]
print(f"gap between consecutive kernels (us): median {q(0.5):.2f}, p90 {q(0.9):.2f}, p99 {q(0.99):.2f}, max {gaps_all[-1]:.1f}")
