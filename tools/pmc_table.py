#!/usr/bin/env python3
"""Per-kernel counter table from rocprofv3 --pmc passes (debug/profiling aid, not a test).

    python tools/pmc_table.py OUT.md steps PASS_DIR [PASS_DIR ...]

Each PASS_DIR holds one `*_counter_collection.csv` of a `rocprofv3 --kernel-trace --pmc <counters> --output-format csv` run of the
SAME command (bench.py with `steps` steps in total, warm-up included).  Counters of all passes are merged per kernel name; the
table lists the top kernels by summed duration (End-Start timestamps of the pass that holds SQ_BUSY_CYCLES, else the first pass).

Columns (per step): time, MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES * 4 SIMDs per CU-cycle ... see note), VALU active,
wave wait, HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950: FETCH_SIZE reports half of wide streaming reads, MI355X_MICROARCH.md).
"""
import collections
import csv
import glob
import os
import sys


def load(pass_dir):
    files = glob.glob(os.path.join(pass_dir, "**", "*counter_collection.csv"), recursive=True)
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    return rows


def main():
    out, steps, dirs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
    val = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = collections.defaultdict(float)
    calls = collections.defaultdict(set)
    timed = False
    for d in dirs:
        rows = load(d)
        names = {r["Counter_Name"] for r in rows}
        use_time = (not timed) and ("SQ_BUSY_CYCLES" in names or d == dirs[-1])
        seen = set()
        for r in rows:
            k = r["Kernel_Name"]
            val[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if use_time and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                calls[k].add(r["Dispatch_Id"])
        timed = timed or use_time
    top = sorted(dur, key=dur.get, reverse=True)[:16]
    lines = ["| kernel | calls/step | ms/step (under PMC) | MFMA busy % | VALU active % of wave cycles | waves parked % (SQ_WAIT_ANY) | "
             "issue-stalled % (SQ_WAIT_INST_ANY) | HBM read GB/step (2xFETCH_SIZE) | HBM write GB/step | HBM GB/s |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    for k in top:
        v = val[k]
        ms = dur[k] / steps / 1e6
        busy = v.get("SQ_BUSY_CYCLES", 0.0)
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; SQ_BUSY_CYCLES counts cycles summed over the chip's SQs
        # (one per shader engine, 32).  GRBM_GUI_ACTIVE (summed over 8 XCDs) / 8 = kernel cycles; MFMA busy = cycles / (kernel
        # cycles * 1024 SIMDs), the rocprof MfmaUtil expression.
        gui = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        mfma = 100.0 * mf / (gui * 1024.0) if gui else float("nan")
        wc = v.get("SQ_WAVE_CYCLES", 0.0)
        pct = lambda c: (100.0 * v.get(c, 0.0) / wc) if wc else float("nan")        # noqa: E731
        rd = 2.0 * v.get("FETCH_SIZE", 0.0) * 1024 / steps / 1e9
        wr = v.get("WRITE_SIZE", 0.0) * 1024 / steps / 1e9
        short = k.replace("void ", "")[:70]
        lines.append(f"| `{short}` | {len(calls[k]) / steps:.1f} | {ms:.3f} | {mfma:.1f} | {pct('SQ_ACTIVE_INST_VALU'):.1f} | "
                     f"{pct('SQ_WAIT_ANY'):.1f} | {pct('SQ_WAIT_INST_ANY'):.1f} | {rd:.3f} | {wr:.3f} | {(rd + wr) / (ms * 1e-3) if ms else 0:.0f} |")
        if busy and mf:
            lines[-1] += f" <!-- SQ_VALU_MFMA_BUSY_CYCLES/SQ_BUSY_CYCLES = {mf / busy:.3f} -->"
    total = sum(dur.values()) / steps / 1e6
    hdr = [f"Per-kernel PMC summary, {steps} steps, passes: {', '.join(os.path.basename(d.rstrip('/')) for d in dirs)}", "",
           f"GPU kernel time per step under the counter passes: {total:.2f} ms (counter collection serialises dispatches; use the "
           "kernel-trace CSV for times)", ""]
    open(out, "w").write("\n".join(hdr + lines) + "\n")
    print("\n".join(hdr + lines))


if __name__ == "__main__":
    main()
