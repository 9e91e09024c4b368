"""fused multi-tensor AdamW -- the optimiser step that follows the backward (optim.AdamW(model.parameters(),
lr, weight_decay), ade_semantic.py:379,401).  torch.optim.AdamW semantics (decoupled decay, bias correction); one kernel
launch updates every parameter, and ``grad_scale`` un-scales fp16-loss-scaled gradients inside the same pass."""
from __future__ import annotations

import math
import struct

import torch

from . import _lib
from ._lib import call, ptr, stream


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._plans = {}

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
            plan = (torch.tensor(bt, dtype=torch.int32, device=dev), torch.tensor(bc, dtype=torch.int32, device=dev), len(bt), ring, [0])
            for k in [k for k in self._plans if k[0] == gi]:      # a re-allocated parameter list of THIS group: drop its old plan
                del self._plans[k]
            self._plans[key] = plan
        return plan

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        loss = closure() if closure is not None else None
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
            b1, b2 = group["betas"]
            rows = []
            for p in params:
                st = self.state[p]
                g = p.grad
                if g is not None:
                    st["step"] += 1                  # one counter per parameter, as torch.optim.AdamW
                    if g.dtype != torch.float32 or not g.is_contiguous():
                        g = g.float().contiguous()
                        p.grad = g
                t = max(st["step"], 1)
                bc = struct.unpack("q", struct.pack("ff", 1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t)))[0]
                rows.append((p.data_ptr(), g.data_ptr() if g is not None else 0, st["exp_avg"].data_ptr(),
                             st["exp_avg_sq"].data_ptr(), p.numel(), bc))
            # pointer table: {p, g, m, v, n, (bc1, bc2_sqrt)} = 6 x 8 bytes per tensor, uploaded asynchronously (pinned staging)
            bt, bc, nblocks, ring, cursor = self._plan(gi, params)
            host, table, ev = ring[cursor[0] % len(ring)]
            cursor[0] += 1
            ev.synchronize()                 # the upload that last used this slot (4 steps ago) is long done
            host.copy_(torch.tensor(rows, dtype=torch.int64))
            table.copy_(host, non_blocking=True)
            ev.record()
            call("mu_adamw_multi", ptr(table), ptr(bt), ptr(bc), nblocks, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                 float(group["weight_decay"]), 1.0 / float(grad_scale), stream())
            for p in params:                 # the kernel wrote the parameters through raw pointers: tell autograd / the weight-layout cache
                if p.grad is not None:
                    torch.autograd.graph.increment_version(p)
        return loss
