#!/usr/bin/env python3
"""Which torch (aten) operations still run inside one training step of the bench configuration (debug aid): torch.profiler over two
steps, aten ops grouped by name + input shapes with the Python source line that issued them.
    python tools/torch_ops_in_step.py [B]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maskunet_amd
from torch.profiler import profile, ProfilerActivity

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = "cuda"
torch.manual_seed(0)
model = maskunet_amd.UNet(3, 150).to(dev)
model.set_compute_dtype(torch.float16).train()
crit = maskunet_amd.CrossEntropyLoss()
x = torch.rand(B, 3, 128, 128, device=dev)
lab = torch.randint(0, 150, (B, 128, 128), device=dev)


def step():
    model.zero_grad(set_to_none=True)
    out = model(x)
    loss = crit(out, lab) * 1024.0
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(2):
        step()
    torch.cuda.synchronize()
rows = {}
for e in prof.events():
    if not e.name.startswith("aten::"):
        continue
    if e.name in ("aten::empty", "aten::empty_like", "aten::view", "aten::as_strided", "aten::empty_strided", "aten::detach", "aten::reshape",
                  "aten::alias", "aten::select", "aten::slice", "aten::expand", "aten::t", "aten::transpose", "aten::permute", "aten::unsqueeze",
                  "aten::squeeze", "aten::_unsafe_view", "aten::result_type", "aten::item", "aten::_local_scalar_dense", "aten::lift_fresh",
                  "aten::is_nonzero", "aten::resolve_conj", "aten::resolve_neg", "aten::view_as", "aten::unbind", "aten::narrow", "aten::contiguous"):
        continue
    src = next((s for s in (e.stack or []) if "maskunet_amd" in s or "bench" in s or "tools/" in s), (e.stack or ["?"])[0] if e.stack else "?")
    key = (e.name, str(e.input_shapes)[:80], src[-90:])
    r = rows.setdefault(key, [0, 0.0])
    r[0] += 1
    r[1] += e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
for (name, shp, src), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{n / 2:6.1f}/step {t / 2:9.1f} us/step  {name:28s} {shp:80s} {src}")
