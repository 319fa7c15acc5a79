"""A whole training step -- forward, backward, optimizer -- as ONE hipGraph replay.

The fused 1vsAll step at the FB15k-237 shape is ~150 us of kernels (profiles/r5_train_step_kernels.txt) issued by
~240 us of Python: autograd bookkeeping, nine launches through ctypes, the optimizer's loop over parameters.  The
GPU waits for the host.  A batch of a fixed shape is the same launch sequence every time, so it is captured once
(torch.cuda.CUDAGraph; HIP graphs underneath) and replayed: the host copies the batch's indexes into static buffers and
issues one graph launch.

    step = GraphedStep(lambda t: model.loss_sp_po_sum(t[:, 0], t[:, 1], t[:, 2], 1.0 / len(t)), optimizer)
    for batch in loader:                                       # batch: [n, 3] triples on the device
        loss = step(batch)                                     # a 0-d tensor (static: read it before the next call)

Mirrors the step of TrainingJob.run_epoch (kge/job/train.py:452-474: zero_grad, forward + backward of the batch,
optimizer.step) for the jobs whose batches are index tensors of a fixed shape: 1vsAll and negative sampling; the last,
shorter batch of an epoch and anything else that does not fit runs eagerly through the same callables.

Used by the LibKGE plugin (`hip_1vsAll.graph_step`, default true: kge_amd/libkge_plugin/train_job.py), by bench.py's
roofline_train leg and by tools/graph_step_probe.py.  Verified: tests/test_gpu_train_graph.py (losses to 1e-6, SGD
parameters to 2e-5 of an eager run, recapture at a learning-rate change, eager fallback for a short batch), 200 steps
at the FB15k-237 shape (tools/graph_step_probe.py), and through an unmodified LibKGE two epochs against the eager job
(tests/test_gpu_libkge_plugin.py test a: second-epoch loss to 1e-5, parameters 5e-5).  Single steps of an Adagrad run
differ from the eager run's by up to 3e-4 (a replay orders the float atomics of the gradient scatter differently and
Adagrad turns a coordinate whose gradient is that noise into a +-lr move); epoch means agree to 2e-6.

A trap this module's first version fell into, for whoever captures library calls into graphs: a hipMemsetAsync
captured into a hipGraph becomes a memset node that ROCm 7 replays with its blit fill kernel from a 16-byte pattern the
graph does not own.  After ~100 replays the "zeroed" relation-gradient accumulator came back as a repeating 16-byte
pattern with one garbage dword, and every replayed step added that to the relation table.  The
library no longer calls hipMemsetAsync anywhere (common.hpp fill_words_async: a kernel carries its value as a launch
argument, which the graph owns).

What a captured step cannot follow, and what is done about it:
  * a changed learning rate (schedulers; kge/job/train.py:406-431): the kernels take lr as a launch argument, so the
    graph is captured again when any group's lr differs from the captured one;
  * Adagrad's lr_decay makes the step size depend on the step count: refused (eager);
  * the first `warmup` calls run eagerly (they are real steps), on the stream the capture will use: lazy allocations
    -- optimizer state, the engine's per-stream workspaces and their one-time clearing, the bf16 scoring copies --
    happen there, outside the capture.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch

__all__ = ["GraphedStep"]


class GraphedStep:
    def __init__(self, loss_fn: Callable[..., torch.Tensor], optimizer: torch.optim.Optimizer, warmup: int = 3,
                 enabled: bool = True):
        """loss_fn(*index_tensors) -> 0-d loss (its backward() fills the parameters' .grad); optimizer: its step()
        must be capturable -- kernels only, no host read of device memory (kge_amd.optim.Adagrad and SGD are)."""
        self.loss_fn = loss_fn
        self.optimizer = optimizer
        self.warmup = int(warmup)
        self.enabled = bool(enabled)
        self.calls = 0
        self.replays = 0
        self.captures = 0
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._static_in: Sequence[torch.Tensor] = ()
        self._static_loss: Optional[torch.Tensor] = None
        self._lrs = None
        self._stream: Optional[torch.cuda.Stream] = None  # the capture stream; the warm-up steps run on it too
        self.static_grads = []
        self.disabled_reason: Optional[str] = None
        for g in optimizer.param_groups:
            if g.get("lr_decay", 0) != 0:
                self._disable("the optimizer's lr_decay makes the step size a function of the step count")
        # A captured step() freezes every launch argument the host computed.  Adam's bias correction (lr / (1 - b1^t),
        # sqrt(1 - b2^t)) is such an argument: a replay would keep the values of the capture step for ever.
        # An optimizer says so itself (`graph_capturable`, kge_amd.optim); of torch's own only SGD is known to be safe.
        cap = getattr(optimizer, "graph_capturable", None)
        if cap is None:
            cap = type(optimizer) is torch.optim.SGD
        if not cap:
            self._disable(f"{type(optimizer).__name__}.step() computes launch arguments from the step count on the host")

    def _disable(self, why: str) -> None:
        self.enabled = False
        self.disabled_reason = why

    def _eager(self, inputs) -> torch.Tensor:
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.loss_fn(*inputs)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def _side_stream(self, device) -> torch.cuda.Stream:
        if self._stream is None:
            self._stream = torch.cuda.Stream(device)
        return self._stream

    def _eager_on_capture_stream(self, inputs) -> torch.Tensor:
        """A warm-up step on the stream the capture will run on.  The engine keeps its scratch per (device, stream)
        -- workspaces that must be ZEROED ONCE before their first use --: what the warm-up allocates and clears there
        is what the captured calls find.  Warmed up on the caller's stream instead, the capture allocated a second
        set and the clearing fill (60-100 MB) became a node of the graph: 7-10 us of every replay until round 5."""
        dev = inputs[0].device
        side, cur = self._side_stream(dev), torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            loss = self._eager(inputs)
        cur.wait_stream(side)
        loss.record_stream(cur)
        return loss

    def _hyper(self):
        """What a captured optimizer step froze into its launch arguments, per parameter group: a change of any of it
        (a scheduler's new learning rate, another weight decay / eps, a parameter added to a group) captures again.
        The parameters' own storage and the gradient buffers must stay where they are while the step is graphed."""
        return [(g.get("lr"), g.get("weight_decay", 0), g.get("eps", None), g.get("momentum", None), len(g["params"]))
                for g in self.optimizer.param_groups]

    def _signature(self, inputs):
        return tuple((tuple(x.shape), x.dtype, x.device) for x in inputs)

    def _capture(self, inputs) -> None:
        self._static_in = tuple(torch.empty_like(x, memory_format=torch.contiguous_format) for x in inputs)
        for dst, src in zip(self._static_in, inputs):
            dst.copy_(src)
        self.optimizer.zero_grad(set_to_none=True)
        # the root gradient of backward(): a static 1, filled here, instead of the ones_like launch of every replay
        one = torch.ones((), dtype=torch.float32, device=inputs[0].device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=self._side_stream(inputs[0].device)):
            self.optimizer.zero_grad(set_to_none=True)
            loss = self.loss_fn(*self._static_in)
            if loss.dim() == 0 and loss.dtype == one.dtype and loss.device == one.device:
                loss.backward(gradient=one)
            else:
                loss.backward()
            self.optimizer.step()
            self._static_loss = loss.detach()
        # the gradient buffers the captured backward writes and the captured optimizer reads (static: a replay fills
        # them again; the parameters' .grad attributes may be set to None by the caller in between)
        self.static_grads = [p.grad for g in self.optimizer.param_groups for p in g["params"]]
        self._stepped = [p for g in self.optimizer.param_groups for p in g["params"] if p.grad is not None]
        self._graph = graph
        self._root_one = one
        self._sig = self._signature(inputs)
        self._lrs = self._hyper()
        self.captures += 1

    @property
    def static_inputs(self) -> Sequence[torch.Tensor]:
        """The captured step's input buffers (empty before the capture).  A caller that writes the next batch INTO them
        -- and hands them back as the call's arguments -- saves the device-to-device copy in front of every replay
        (4.5 us of a 150 us step at the FB15k-237 shape); host tensors handed to the call are copied host-to-device
        straight into them."""
        return tuple(self._static_in) if self._graph is not None else ()

    def _on_device(self, inputs):
        """Host tensors (the batch as the loader hands it over) -> the device of the parameters; device tensors as they are."""
        if all(isinstance(x, torch.Tensor) and x.is_cuda for x in inputs):
            return inputs
        dev = next((p_.device for g in self.optimizer.param_groups for p_ in g["params"]), None)
        if dev is None or dev.type != "cuda" or not all(isinstance(x, torch.Tensor) for x in inputs):
            return inputs
        return tuple(x.to(dev, non_blocking=True) for x in inputs)

    def __call__(self, *inputs: torch.Tensor) -> torch.Tensor:
        self.calls += 1
        replayable = (self.enabled and self._graph is not None and self.calls > self.warmup
                      and all(isinstance(x, torch.Tensor) for x in inputs)
                      and self._hyper() == self._lrs
                      and tuple((tuple(x.shape), x.dtype) for x in inputs) == tuple((s_[0], s_[1]) for s_ in self._sig))
        if not replayable:
            inputs = self._on_device(inputs)
        if not self.enabled or not all(isinstance(x, torch.Tensor) and (x.is_cuda or replayable) for x in inputs):
            return self._eager(inputs)
        if self.calls <= self.warmup:
            return self._eager_on_capture_stream(inputs)
        lrs = self._hyper()
        fresh = False
        if self._graph is None or lrs != self._lrs:
            fresh = True
            if self._graph is not None and self._signature(inputs) != self._sig:
                return self._eager(inputs)  # (a short batch right at a learning-rate change: the next full one captures)
            try:
                self._capture(inputs)
            except Exception as exc:  # a step that cannot be captured stays eager -- loudly, once
                import warnings
                self._graph = None
                self._disable(f"capture failed: {exc!r}")
                warnings.warn(f"kge_amd.GraphedStep: {self.disabled_reason}; the step runs eagerly")
                torch.cuda.synchronize()
                return self._eager(inputs)
        elif not replayable:
            return self._eager(inputs)  # e.g. the epoch's last, shorter batch
        elif all(src is dst for dst, src in zip(self._static_in, inputs)):
            pass  # the caller wrote the batch into static_inputs itself: nothing to copy
        elif len(inputs) > 1 and hasattr(torch, "_foreach_copy_") and all(x.is_cuda for x in inputs):
            # one multi-tensor launch where torch can (dense tensors of one layout), else its own loop of copies.  Each
            # copy is a launch of ~4.7 us in front of the replay: hand a batch over as ONE tensor where there is a choice
            torch._foreach_copy_(list(self._static_in), list(inputs), non_blocking=True)
        else:
            for dst, src in zip(self._static_in, inputs):
                if src is not dst:
                    dst.copy_(src, non_blocking=True)  # (a host tensor: ONE host-to-device copy, no device-side hop)
        self._graph.replay()
        self.replays += 1
        after = getattr(self.optimizer, "after_graph_replay", None)
        if after is not None and not fresh:
            # host-side bookkeeping the captured kernels do not carry: the per-parameter step counts of the checkpoint
            # (the capture itself ran step() once on the host, so the first replay is already counted)
            after(self._stepped)
        return self._static_loss
