"""fused multi-tensor AdamW -- the optimiser step that follows the backward (optim.AdamW(model.parameters(),
lr, weight_decay), ade_semantic.py:379,401).  torch.optim.AdamW semantics (decoupled decay, bias correction); one kernel
launch updates every parameter, and ``grad_scale`` un-scales fp16-loss-scaled gradients inside the same pass.

fp16 training safety: a loss-scaled backward can overflow, and one inf / NaN gradient written into the fp32 master weights is not
recoverable.  ``step(grad_scale=s)`` therefore checks every gradient on the device first (one read of the gradients, ``check_finite``,
default on whenever a scale is in use) and an overflowed step updates NOTHING -- parameters, moments and the effective step count of
the bias corrections stay as they were; no host sync is involved (``last_step_skipped()`` reads the flag when the caller wants to
know, e.g. to lower a static scale).  ``torch.cuda.amp.GradScaler`` works as with torch's fused optimisers: the class sets
``_step_supports_amp_scaling``, so ``scaler.step(opt)`` hands over its device-side ``grad_scale`` / ``found_inf`` tensors and the
kernel un-scales and skips by them (``scaler.update()`` then adapts the scale)."""
from __future__ import annotations

import torch

from . import _lib
from ._lib import call, ptr, stream


class FusedAdamW(torch.optim.Optimizer):
    _step_supports_amp_scaling = True          # GradScaler.step(): pass grad_scale / found_inf as device tensors, no .item() sync

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._plans = {}
        self._found = {}                        # device -> float32 [1] overflow flag of this optimiser's own finite check
        self._last_found = []                   # the flags the LAST step() decided by (its own, or the ones GradScaler handed in)

    def _plan(self, gi, params):
        key = (gi, tuple(p.data_ptr() for p in params))
        plan = self._plans.get(key)
        if plan is None:
            chunk = _lib.load().mu_adamw_chunk()
            bt, bc = [], []
            for i, p in enumerate(params):
                n = (p.numel() + chunk - 1) // chunk
                bt += [i] * n
                bc += list(range(n))
            dev = params[0].device
            # pointer-table staging: a small ring of pinned host buffers (+ the event of their last upload) and one device
            # table per slot, allocated once -- pinning memory per step costs milliseconds of host time
            ring = [(torch.empty((len(params), 6), dtype=torch.int64).pin_memory(),
                     torch.empty((len(params), 6), dtype=torch.int64, device=dev), torch.cuda.Event()) for _ in range(4)]
            plan = (torch.tensor(bt, dtype=torch.int32, device=dev), torch.tensor(bc, dtype=torch.int32, device=dev), len(bt), ring, [0],
                    torch.zeros(len(params), dtype=torch.int32, device=dev))      # skipped steps per tensor (device side)
            for k in [k for k in self._plans if k[0] == gi]:      # a re-allocated parameter list of THIS group (model.to() / .half() /
                old = self._plans.pop(k)                          # .float()): its device-side skip counts move into the host step
                if len(k[1]) == len(params):                      # counters first -- dropping them would over-count `step` for good
                    for q, n in zip(params, old[5].tolist()):
                        if n and q in self.state and "step" in self.state[q]:
                            self.state[q]["step"] -= n
            self._plans[key] = plan
        return plan

    def last_step_skipped(self) -> bool:
        """True when the last step() found a non-finite gradient and therefore changed nothing (one D2H read).  Under
        GradScaler.step(opt) the flag read is the scaler's own found_inf tensor of that step (valid until scaler.update() resets it)."""
        return any(bool(f.item() != 0) for f in self._last_found)

    def effective_steps(self, p) -> int:
        """Updates really applied to parameter p: attempted steps minus the skipped ones (one D2H read)."""
        for gi, group in enumerate(self.param_groups):
            params = [q for q in group["params"] if q.requires_grad]
            for i, q in enumerate(params):
                if q is p:
                    plan = self._plans.get((gi, tuple(t.data_ptr() for t in params)))
                    return self.state[p].get("step", 0) - (int(plan[5][i].item()) if plan else 0)
        raise KeyError("parameter not in this optimizer")

    def _fold_skips(self):
        """Move the device-side skip counts into the host step counters (one D2H read per group; checkpoint time only)."""
        for gi, group in enumerate(self.param_groups):
            params = [q for q in group["params"] if q.requires_grad]
            plan = self._plans.get((gi, tuple(t.data_ptr() for t in params)))
            if plan is None:
                continue
            sk = plan[5].tolist()
            if any(sk):
                for q, n in zip(params, sk):
                    if n:
                        self.state[q]["step"] -= n
                plan[5].zero_()

    def state_dict(self):
        """torch.optim.AdamW-compatible state; `step` holds the updates really applied (skipped overflow steps folded in)."""
        self._fold_skips()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """The loaded `step` counters already are "updates really applied": skips counted on the device since the last state_dict() belong to
        the state being replaced and must not be subtracted from the loaded counters (a rollback would otherwise reach t <= 0)."""
        super().load_state_dict(state_dict)
        for plan in self._plans.values():
            plan[5].zero_()

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0, check_finite=None):
        """grad_scale: static loss scale the gradients carry (divided out inside the kernel).  check_finite: look for inf / NaN gradients
        on the device first and skip the whole update if there is one; default = whenever a loss scale is in use.  Under
        torch.cuda.amp.GradScaler.step(opt) the scaler's own device-side scale and overflow flag are used instead."""
        loss = closure() if closure is not None else None
        amp_scale = getattr(self, "grad_scale", None)        # set by GradScaler.step() around this call (device float tensors)
        amp_found = getattr(self, "found_inf", None)
        if check_finite is None:
            check_finite = amp_found is None and float(grad_scale) != 1.0
        jobs = []
        self._last_found = []
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.requires_grad]
            if not params:
                continue
            for p in params:
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                    raise RuntimeError("FusedAdamW needs contiguous fp32 CUDA parameters")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
            # (before the step counters are read: a re-planned group folds its old skip counts into them)
            bt, bc, nblocks, ring, cursor, skipped = self._plan(gi, params)
            rows = []
            for p in params:
                st = self.state[p]
                g = p.grad
                if g is not None:
                    st["step"] += 1                  # one counter per parameter, as torch.optim.AdamW (attempted updates)
                    if g.dtype != torch.float32 or not g.is_contiguous():
                        g = g.float().contiguous()
                        p.grad = g
                rows.append((p.data_ptr(), g.data_ptr() if g is not None else 0, st["exp_avg"].data_ptr(),
                             st["exp_avg_sq"].data_ptr(), p.numel(), max(st["step"], 1)))
            # pointer table: {p, g, m, v, n, step} = 6 x 8 bytes per tensor, uploaded asynchronously (pinned staging)
            dev = params[0].device
            if amp_found is not None:
                found = amp_found.to(device=dev, dtype=torch.float32).reshape(-1)[:1]
            elif check_finite:
                found = self._found.get(dev)
                if found is None:
                    found = self._found[dev] = torch.zeros(1, dtype=torch.float32, device=dev)
            else:
                found = None
            if found is not None and not any(f is found for f in self._last_found):
                self._last_found.append(found)
            scale_dev = amp_scale.to(device=dev, dtype=torch.float32).reshape(-1)[:1] if amp_scale is not None else None
            host, table, ev = ring[cursor[0] % len(ring)]
            cursor[0] += 1
            ev.synchronize()                 # the upload that last used this slot (4 steps ago) is long done
            host.copy_(torch.tensor(rows, dtype=torch.int64))
            table.copy_(host, non_blocking=True)
            ev.record()
            jobs.append((group, params, table, bt, bc, nblocks, skipped, found, scale_dev))
        own_check = bool(check_finite) and amp_found is None
        if own_check and len(jobs) > 1:
            # several parameter groups: ONE decision for the whole step -- every group's gradients are checked into the same flag
            # (mode 2: check only) before any group is updated
            for f in {id(j[7]): j[7] for j in jobs}.values():
                f.zero_()
            for group, params, table, bt, bc, nblocks, skipped, found, scale_dev in jobs:
                call("mu_adamw_multi", ptr(table), ptr(bt), ptr(bc), nblocks, len(params), 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, None, ptr(found),
                     2, None, stream())
        for group, params, table, bt, bc, nblocks, skipped, found, scale_dev in jobs:
            b1, b2 = group["betas"]
            call("mu_adamw_multi", ptr(table), ptr(bt), ptr(bc), nblocks, len(params), float(group["lr"]), float(b1), float(b2),
                 float(group["eps"]), float(group["weight_decay"]), 1.0 / float(grad_scale), ptr(scale_dev), ptr(found),
                 int(own_check and len(jobs) == 1), ptr(skipped), stream())     # `skipped` always: t = step - skipped also in a step without a check
            for p in params:                 # the kernel wrote the parameters through raw pointers: tell autograd / the weight-layout cache
                if p.grad is not None:
                    torch.autograd.graph.increment_version(p)
        return loss
