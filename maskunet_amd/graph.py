"""Whole training step (forward, criterion, backward) captured in ONE HIP graph.

One step of the path is ~430 kernel launches; enqueueing them from Python costs ~8 ms of host time, which is hidden behind the GPU at
B >= 32 per GPU and becomes the bound below B ~ 16 (profiles/r06_step_gaps_by_batch.txt: 2.7 ms of idle GPU per
step at B = 8 eager, 0.04 ms replayed).  Every launch of the path is stream-ordered on torch's current
stream with no host synchronisation, so the step can be captured (torch.cuda.CUDAGraph == hipGraph) and replayed with a single launch:

    step = maskunet_amd.GraphedStep(model, criterion, example_inputs, example_labels, loss_scale=1024.0)
    for inputs, labels in loader:
        loss = step(inputs, labels)          # gradients are in p.grad afterwards (the same tensors every replay)
        optimizer.step()                     # eager, after the replay; zero_grad() in between is harmless: every call re-attaches
                                             # the graph's gradient tensors to p.grad, and the replay overwrites them

What changes under capture: the dropout seed drawn on the host is baked into the graph, so a device step counter -- incremented inside the
graph -- is mixed into it (mu_dropout_step); BatchNorm running statistics / step counters are updated on the device as always; the
attention keep-masks are whatever the modules hold at capture time (set_keep_masks / the lazily drawn, cached masks of the reference)
-- or, with mask_mode = "resample" (what the reference does under multi-GPU nn.DataParallel: a fresh draw per replica forward,
ade_semantic.py:177-181 + :373), REDRAWN BY EVERY REPLAY: the torch.randint draws and the key compactions (mu_compact_keys) are captured
with the step, and torch's CUDA generator gives a captured graph a fresh Philox offset per replay (round 6).  After a replay the
modules' `_keep` tensors hold the masks that replay used.

With a maskunet_amd.DataParallel model the replica's step is captured (the graph holds no collective) and the gradient exchange runs
eagerly after every replay: the same bucketed all-reduce on the comm stream, without overlap with the backward -- the hooks of
DataParallel sit on the parameters' AccumulateGrad nodes, which the captured torch.autograd.grad on fresh leaves never runs.  The
captured weight-gradient kernels write straight into DataParallel's bucket slices (its gradient arena, laid out after the first warm-up
step), the buckets are reduced in place and p.grad is the reduced slice -- no flattening and no copy back per replay.  The capture is
SEGMENTED (round 5): two graphs split inside the one backward pass at ops.cut_point (encoder | bottleneck + decoder), so the all-reduce
of the buckets the first segment completes runs on the comm stream beside the second segment's replay (see _capture_segmented).  Gradient accumulation is NOT available through a graph: a
replay overwrites the graph's gradient tensors instead of adding to them, so a call inside `DataParallel.no_sync()` raises.
"""
from __future__ import annotations

import torch

from . import ops
from .dp import DataParallel


class GraphedStep:
    def __init__(self, model, criterion, example_inputs, example_labels, loss_scale: float = 1.0, warmup: int = 2, segments: int = 2):
        if not example_inputs.is_cuda:
            raise RuntimeError("GraphedStep needs CUDA example inputs (the HIP path has no CPU fallback)")
        self.dp = model if isinstance(model, DataParallel) else None
        model = model.module if self.dp is not None else model
        self.model, self.criterion, self.scale = model, criterion, float(loss_scale)
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        self.names, self.params = [n for n, _ in named], [p for _, p in named]
        self.inputs = example_inputs.detach().clone()
        self.labels = example_labels.detach().clone()
        dev = self.inputs.device
        self.step_counter = torch.zeros(1, dtype=torch.int64, device=dev)
        # eager warm-up on a side stream (allocator pools, workspaces, lazily drawn masks), as torch's graph recipe prescribes
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for it in range(max(int(warmup), 1)):
                self._body()
                if it == 0 and self.dp is not None:
                    self.dp.prepare_arena()      # from here on the large gradients are written into their all-reduce bucket slices
                model.zero_grad(set_to_none=True)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        self.graph2 = None                      # second segment (DataParallel): the backward behind ops.cut_point
        self._seg1 = set()                      # ids of the parameters whose gradients are complete when the first segment ends
        ops.SEED_STEP = self.step_counter
        try:
            # thread_local: with an initialised nccl (RCCL) process group the backend's watchdog thread polls the events of finished
            # collectives (hipEventQuery) at any time; under the default "global" capture mode that call from ANOTHER thread is an
            # error ("operation not permitted when stream is capturing") and takes the process down
            if self.dp is not None and self.dp.multi and segments >= 2:
                try:
                    self._capture_segmented()
                except Exception as e:              # noqa: BLE001 -- fall back to ONE graph + the exchange behind it (segments=1)
                    import warnings
                    warnings.warn(f"GraphedStep: segmented capture failed ({e!r}); capturing the step as one graph, the gradient "
                                  "exchange then runs after the replay without overlap")
                    torch.cuda.synchronize(dev)
                    self.graph, self.graph2, self._seg1 = torch.cuda.CUDAGraph(), None, set()
                    ops.CUT_HOOK = None
                    with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                        self.step_counter.add_(1)
                        self.loss, self.outputs = self._body()
            else:
                with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                    self.step_counter.add_(1)
                    self.loss, self.outputs = self._body()
        finally:
            ops.SEED_STEP = None
            ops.CUT_HOOK = None

    def _capture_segmented(self):
        """The step as TWO graphs sharing one memory pool, split INSIDE the single backward pass (a collective cannot be captured on this
        stack: tools/probe_graph_rccl.py dumps core): the gradient hook of ops.cut_point -- it fires when every node behind the cut has
        run -- ends the first capture and begins the second.  One backward pass, so ops.GradLink / EncLink pairs across the cut keep
        working.  Replay: graph 1, the all-reduces of the buckets it completed (comm stream), graph 2 beside them, the rest."""
        import gc
        dev = self.inputs.device
        g1, g2 = self.graph, torch.cuda.CUDAGraph()
        state = {"cut": False, "ended1": False, "fired": set()}

        def cut(_grad):
            if not state["cut"] and not state["ended1"]:
                self._seg1 = set(state["fired"])
                g1.capture_end()
                state["ended1"] = True                   # from here on g1 must not be ended again
                g2.capture_begin(pool=g1.pool(), capture_error_mode="relaxed")
                state["cut"] = True                      # only now is g2 the capture to end
            return None

        torch.cuda.synchronize(dev)
        gc.collect()
        torch.cuda.empty_cache()
        cap = torch.cuda.Stream(dev)
        cap.wait_stream(torch.cuda.current_stream(dev))
        ops.CUT_HOOK = cut
        self._leaf_fired = state["fired"]
        with torch.cuda.stream(cap):
            # "relaxed": the hook runs on the autograd engine's device thread, and a capture begun in the stricter modes may only be
            # ended by the thread that began it (relaxed also tolerates the RCCL watchdog's event queries, see above).  The price:
            # relaxed mode switches OFF the runtime's capture safety checks -- an allocation or a synchronising call made by any
            # thread while the capture runs is no longer rejected -- so nothing else may touch the device during construction.
            g1.capture_begin(capture_error_mode="relaxed")
            err = None
            try:
                self.step_counter.add_(1)
                self.loss, self.outputs = self._body()
            except BaseException as e:                   # keep the ORIGINAL error: ending an invalidated capture raises its own
                err = e
            finally:
                ops.CUT_HOOK = None
                try:
                    if state["cut"]:
                        g2.capture_end()
                    elif not state["ended1"]:
                        g1.capture_end()
                except Exception:
                    if err is None:
                        raise
            if err is not None:
                raise err
        torch.cuda.current_stream(dev).wait_stream(cap)
        self._leaf_fired = None
        if state["cut"]:
            self.graph2 = g2

    def _body(self):
        # The forward runs on fresh leaves aliasing the parameters (functional_call), and the gradients are taken with
        # torch.autograd.grad.  A parameter's AccumulateGrad node is bound to the stream of the forward that created it; one kept alive
        # by an earlier eager step (e.g. through a loss tensor the caller still holds) makes the engine synchronise the capture stream
        # with that stream, which is not capturable (segfault in hipStreamEndCapture on this stack).  Fresh leaves have fresh nodes.
        fresh = {n: p.detach().requires_grad_(True) for n, p in zip(self.names, self.params)}
        fired = getattr(self, "_leaf_fired", None)
        if fired is not None:                    # segmented capture: note the order in which the parameter gradients complete
            for n, p in zip(self.names, self.params):
                fresh[n].register_hook(lambda g, k=id(p): fired.add(k))
        # DataParallel's gradient arena is keyed on the parameters; the kernels here see the fresh leaves: register them for the call
        arena_keys = []
        ops.GRAD_HANDED.clear()
        if self.dp is not None and ops.GRAD_ARENA is not None:
            for n, p in zip(self.names, self.params):
                v = self.dp.gradient_slice(p)
                if v is not None:
                    ops.GRAD_ARENA[id(fresh[n])] = v
                    arena_keys.append(id(fresh[n]))
        try:
            out = torch.func.functional_call(self.model, fresh, (self.inputs,))
            sem = out[0] if isinstance(out, (tuple, list)) else out
            loss = self.criterion(sem, self.labels)
            grads = torch.autograd.grad(loss * self.scale if self.scale != 1.0 else loss, list(fresh.values()), allow_unused=True)
        finally:
            for k in arena_keys:
                ops.GRAD_ARENA.pop(k, None)
        self.grads = grads
        for p, g in zip(self.params, grads):
            p.grad = g
        return loss.detach(), sem.detach()

    def __call__(self, inputs, labels):
        if self.dp is not None and not self.dp._sync:
            raise RuntimeError("GraphedStep inside DataParallel.no_sync(): a replay overwrites the gradients of the previous micro-batch "
                               "instead of accumulating into them -- run accumulation steps eagerly")
        self.inputs.copy_(inputs, non_blocking=True)
        self.labels.copy_(labels, non_blocking=True)
        self.graph.replay()
        for p, g in zip(self.params, self.grads):     # optimizer.zero_grad(set_to_none=True) between replays detaches them
            p.grad = g
        if self.dp is not None:                       # one exchange per step (ade_semantic.py:373)
            self.dp._arm()
            if self.graph2 is not None:
                # the buckets whose gradients the first segment completed (bottleneck, decoder, heads: most of the bytes) are reduced on
                # the comm stream WHILE the second segment -- the encoder's backward -- replays on this one
                self.dp.launch_complete_buckets(self._seg1)
                self.graph2.replay()
            # the rest.  No copy back: the large gradients ARE their bucket slices (the captured kernels wrote there), the small ones
            # are moved into theirs by the exchange, and p.grad is rebound to the reduced slice -- the graph's own tensors are
            # re-attached (and overwritten) by the next replay anyway
            self.dp.finish_gradient_sync()
        elif self.graph2 is not None:
            self.graph2.replay()
        return self.loss.clone()                      # self.loss is overwritten by the next replay
