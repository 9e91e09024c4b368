"""Summarise a rocprofv3 kernel_stats CSV per bench step: python tools/prof_summary.py <csv> <steps+warmup> [topN]."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"GPU busy per step: {tot / steps / 1e6:.3f} ms")
# (order matters: the first matching group takes the kernel; the fp32x operand-encoding kernels are the library's own, not torch ops)
groups = {"fp32x operand encodings": ("split_encode", "attn_dy_encode", "colsum_enc", "dyh_", "encode_hl"),
          "attention": ("attn_",), "conv fwd/dgrad": ("conv_nt",), "conv wgrad": ("wgrad",), "BN/LN": ("bn_", "lns_"),
          "layout/prep": ("transpose", "prep_weight", "cast_kernel", "u8_to", "prep_qkv", "compact_keys"),
          "pool/upcat/dropout/add": ("maxpool", "upcat", "dropout", "add_kernel"),
          "loss / optimiser / column sums": ("ce_", "adamw", "colsum", "mean_iou", "inst_triplet"), "clock probe": ("clock_probe",)}
acc = {k: 0.0 for k in groups}
other = 0.0
for r in rows:
    t = float(r["TotalDurationNs"]) / steps / 1e6
    for k, pats in groups.items():
        if any(p in r["Name"] for p in pats):
            acc[k] += t
            break
    else:
        other += t
for k, v in acc.items():
    print(f"  {k:30s} {v:7.3f} ms")
print(f"  {'other (torch / RCCL ops)':24s} {other:7.3f} ms")
for r in rows[:top]:
    print(f"{r['Name'][:72]:72s} calls/step {int(r['Calls']) / steps:6.1f} ms/step {float(r['TotalDurationNs']) / steps / 1e6:7.3f} "
          f"avg_us {float(r['AverageNs']) / 1e3:8.1f}")
