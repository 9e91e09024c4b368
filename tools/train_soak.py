"""End-to-end sanity check (debug aid, GPU): 300 fp16 training steps (HIP forward/backward, maskunet_amd.CrossEntropyLoss, FusedAdamW with a
static loss scale) on a learnable synthetic task; the loss must fall and every parameter stay finite.  python tools/train_soak.py
`--scaler`: the same under torch.amp.GradScaler (starting at 2^24, so the first steps overflow: FusedAdamW skips them on the device and the
scaler backs off) -- the fp16 training-safety path of round 4.  `--fp32x`: fp32 storage with split-bf16 matrix products instead of fp16."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maskunet_amd
torch.manual_seed(0)
dev = torch.device("cuda", 0)
C = 8
FP32X, SCALER = "--fp32x" in sys.argv, "--scaler" in sys.argv
model = maskunet_amd.UNet(3, C).to(dev); model.set_compute_dtype(torch.float32 if FP32X else torch.float16).train()
if FP32X:
    maskunet_amd.set_float32_matmul_precision("high")
opt = maskunet_amd.FusedAdamW(model.parameters(), lr=2e-4, weight_decay=1e-2)
scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 24, growth_interval=50) if SCALER else None
STATIC = 1.0 if FP32X else 1024.0
skipped = 0
crit = maskunet_amd.CrossEntropyLoss()
def batch(B=16):
    x = torch.rand(B, 3, 128, 128, device=dev)
    # learnable synthetic task: class = quantised local brightness of a blurred copy
    g = torch.nn.functional.avg_pool2d(x.mean(1, keepdim=True), 9, 1, 4)
    y = (g.squeeze(1) * C * 1.999 - C * 0.5).clamp(0, C - 1).long()
    return x, y
losses = []
for it in range(300):
    x, y = batch()
    loss = crit(model(x), y)
    if scaler is not None:
        scaler.scale(loss).backward()
        s0 = scaler.get_scale()
        scaler.step(opt)
        scaler.update()
        skipped += scaler.get_scale() < s0
    else:
        (loss * STATIC).backward()
        opt.step(grad_scale=STATIC)
    model.zero_grad(set_to_none=True)
    if it % 25 == 0 or it == 299:
        losses.append(round(float(loss.detach()), 4)); print(it, losses[-1], flush=True)
bad = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
print("non-finite params:", bad, "| steps skipped for overflow:", int(skipped), "| final scale:", scaler.get_scale() if scaler else STATIC)
model.eval()
x, y = batch(16)
with torch.no_grad():
    out = model(x)
print("eval pixel acc", float((out.argmax(1) == y).float().mean()), "miou", float(maskunet_amd.mean_iou(out, y, C)))
assert not bad and losses[-1] < 0.8 * losses[0], (bad, losses)
print("soak OK")
