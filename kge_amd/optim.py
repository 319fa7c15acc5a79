"""torch.optim.Adagrad / torch.optim.Adam with the update done by one HIP kernel per parameter
(kge_adagrad_step, kge_adagrad_step_rows for row-sparse gradients, kge_adam_step), optionally
maintaining the bf16 copies of the tables that mixed-precision scoring reads.

Drop-in for `torch.optim.Adagrad` (same constructor arguments, same `state_dict`: per-parameter
"step" and "sum"), so LibKGE's optimizer checkpoints load either way.  Parameters the kernel does
not cover (CPU tensors, non-float32, sparse gradients) are stepped by torch's own functional
Adagrad.  kge/util/optimizer.py:15-20 resolves `train.optimizer.default.type` with
`getattr(torch.optim, ...)`; the LibKGE plugin registers this class there as `HipAdagrad`.
"""
import torch
from torch.optim.adagrad import Adagrad as _TorchAdagrad
from torch.optim.adagrad import adagrad as _functional_adagrad

from . import _lib, engine

# (bf16 tensor, version of the parameter it was made from, data_ptr of the parameter's storage then),
# set on the parameter
BF16_ATTR = "_kge_bf16_copy"


def bf16_copy_of(param: torch.Tensor):
    """The optimizer-maintained bf16 copy of `param`, or None if there is none / it is stale.
    Stale = the version counter moved OR the storage was replaced: `weight.data = ...`
    (LookupEmbedder._normalize_embeddings, lookup_embedder.py:64-69, run as a post-batch hook when
    normalize.p > 0) swaps the storage without bumping the counter."""
    rec = getattr(param, BF16_ATTR, None)
    if rec is None or rec[1] != param._version or rec[2] != param.data_ptr() or rec[0].data_ptr() == 0:
        return None
    return rec[0]


class Adagrad(_TorchAdagrad):
    # kge_amd.train_graph.GraphedStep: the step's launch arguments do not depend on the step count (lr_decay != 0 is
    # refused there), so a captured step() replays correctly; `after_graph_replay` keeps state["step"] -- which the
    # checkpoint carries -- counting.
    graph_capturable = True

    def after_graph_replay(self, params):
        """`params`: the parameters the captured step() updated."""
        for p in params:
            st = self.state.get(p)
            if st is not None and "step" in st:
                st["step"] += 1

    def __init__(self, params, lr=1e-2, lr_decay=0, weight_decay=0, initial_accumulator_value=0, eps=1e-10,
                 bf16_copies: bool = False, **kw):
        """bf16_copies=True: after every step each 2-D float32 parameter carries a fresh bf16 copy
        (written by the same kernel pass) that `kge_amd.model.BF16Shadow` picks up instead of
        re-casting the table before the next scoring call."""
        kw.pop("foreach", None)
        kw.pop("fused", None)
        super().__init__(params, lr=lr, lr_decay=lr_decay, weight_decay=weight_decay,
                         initial_accumulator_value=initial_accumulator_value, eps=eps, foreach=False, **kw)
        self.bf16_copies = bool(bf16_copies)
        self._penalties = {}       # parameter -> (kind, p, gradient factor, row_dim, value factor)
        self._pen_acc = {}         # device -> float64 accumulators, one per parameter with a penalty
        self._pen_slot = {}        # parameter -> index into its device's accumulators
        self._pen_off_once = False

    # ---- the embedders' unweighted penalty terms inside the step (kge_adagrad_step_multi_penalty) ------------------
    def set_penalty(self, param, kind: str, p: int, weight: float, times: float = 1.0):
        """From now on every step() adds to `param`'s gradient the gradient of
            times * weight / p * sum |x|^p                      kind "lp", p in (1, 2, 3)
            times * weight / 3 * sum |z|^3 over complex coordinates   kind "n3_complex" (z = row[c] + i row[c + dim / 2])
        (LookupEmbedder.penalty unweighted, lookup_embedder.py:122-147; `times` = 2 for an entity embedder that serves the
        subject and the object slot, kge_model.py:620-625) inside the update's own pass, and `penalty_value(param)`
        holds the term's value at the parameters the step started from.  The caller must then NOT back-propagate the
        term itself.  kind None removes it."""
        if kind is None:
            self._penalties.pop(param, None)
        else:
            if kind not in ("lp", "n3_complex") or (kind == "lp" and p not in (1, 2, 3)):
                raise ValueError(f"kge_amd.optim.Adagrad: no fused form of penalty {kind} p={p}")
            if not (param.is_cuda and param.dtype == torch.float32 and param.is_contiguous()):
                raise ValueError("kge_amd.optim.Adagrad: fused penalties are for dense float32 GPU parameters")
            if kind == "n3_complex":
                p = 3
                if param.dim() != 2 or param.shape[1] % 8 != 0:
                    raise ValueError("kge_amd.optim.Adagrad: n3_complex needs a [*, dim] table with dim % 8 == 0")
            self._penalties[param] = (1 if kind == "lp" else 2, int(p), float(times) * float(weight),
                                      int(param.shape[-1]), float(times) * float(weight) / float(p))
        self._pen_acc, self._pen_slot = {}, {}
        for q in self._penalties:
            acc = self._pen_acc.get(q.device)
            self._pen_slot[q] = 0 if acc is None else acc.numel()
            self._pen_acc[q.device] = torch.zeros(self._pen_slot[q] + 1, dtype=torch.float64, device=q.device)

    def has_penalties(self) -> bool:
        return bool(self._penalties)

    def skip_penalties_once(self):
        """The next step() is the plain update (the caller back-propagated the terms itself for that batch)."""
        self._pen_off_once = True

    def penalty_value(self, param) -> torch.Tensor:
        """0-d float32 device tensor: the term's value the LAST step() saw (pre-step parameters)."""
        spec = self._penalties[param]
        return (self._pen_acc[param.device][self._pen_slot[param]] * spec[4]).to(torch.float32)

    @staticmethod
    def _kernel_ok(p: torch.Tensor) -> bool:
        g = p.grad
        return (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g is not None
                and not g.is_sparse and g.dtype == torch.float32 and g.is_contiguous())

    @staticmethod
    def _rows_ok(p: torch.Tensor, group) -> bool:
        """Row-sparse gradient of a 2-D table (lookup_embedder.sparse: True -> nn.Embedding(sparse=True),
        lookup_embedder.py:44-46): kge_adagrad_step_rows touches only the rows that have a gradient."""
        g = p.grad
        return (g is not None and g.is_sparse and p.is_cuda and p.dim() == 2 and p.dtype == torch.float32
                and p.is_contiguous() and g.dtype == torch.float32 and group["weight_decay"] == 0)

    def _step_rows(self, p, group):
        state = self.state[p]
        state["step"] += 1
        step = float(state["step"])
        minus_clr = -float(group["lr"]) / (1.0 + (step - 1.0) * group["lr_decay"])
        g = p.grad.coalesce()  # unique row ids, summed value rows (what torch's sparse Adagrad does first)
        rows = g.indices()[0].contiguous()
        vals = g.values().contiguous()
        if rows.numel() == 0:
            return
        s = state["sum"]
        copy = None
        if self.bf16_copies:
            # only the touched rows are rewritten below: the copy must be FRESH before (a parameter changed behind
            # the optimizer's back -- load_state_dict into the same Parameter, a re-init, a dense torch step --
            # leaves every other row stale); a stale or missing one is re-cast once, into the old buffer if it fits
            copy = bf16_copy_of(p)
            if copy is None:
                rec = getattr(p, BF16_ATTR, None)
                if rec is not None and rec[0].shape == p.shape and rec[0].data_ptr() != 0:
                    copy = rec[0]
                    copy.copy_(p.detach())
                else:
                    copy = p.detach().to(torch.bfloat16)
        with torch.cuda.device(p.device):
            _lib.check(_lib.lib().kge_adagrad_step_rows(
                p.data_ptr(), p.stride(0), vals.data_ptr(), vals.stride(0), s.data_ptr(), s.stride(0), rows.data_ptr(),
                rows.numel(), p.shape[1], minus_clr, float(group["eps"]), None if copy is None else copy.data_ptr(),
                0 if copy is None else copy.stride(0), engine._stream(p.device)), "kge_adagrad_step_rows")
        torch.autograd.graph.increment_version(p)
        if copy is not None:
            setattr(p, BF16_ATTR, (copy, p._version, p.data_ptr()))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        dense = {}  # device -> [(p, copy, segment)]: every dense table of a device in ONE launch (kge_adagrad_step_multi)
        for group in self.param_groups:
            if group.get("maximize") or group.get("differentiable"):
                raise NotImplementedError("kge_amd.optim.Adagrad: maximize / differentiable are not supported")
            rest = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if self._rows_ok(p, group):
                    self._step_rows(p, group)
                    continue
                if not self._kernel_ok(p):
                    rest.append(p)
                    continue
                state = self.state[p]
                state["step"] += 1
                step = float(state["step"])
                minus_clr = -float(group["lr"]) / (1.0 + (step - 1.0) * group["lr_decay"])
                s = state["sum"]
                copy = None
                if self.bf16_copies and p.dim() == 2:
                    rec = getattr(p, BF16_ATTR, None)
                    copy = rec[0] if rec is not None and rec[0].shape == p.shape else \
                        torch.empty(p.shape, dtype=torch.bfloat16, device=p.device)
                seg = _lib.KgeAdagradSeg(p.data_ptr(), p.grad.data_ptr(), s.data_ptr(),
                                         None if copy is None else copy.data_ptr(), p.numel(), minus_clr,
                                         float(group["weight_decay"]), float(group["eps"]))
                dense.setdefault(p.device, []).append((p, copy, seg))
            if rest:  # torch's own update for everything else
                if any(p in self._penalties for p in rest) and not self._pen_off_once:
                    raise RuntimeError("kge_amd.optim.Adagrad: a parameter with a fused penalty left the kernel path")
                grads = [p.grad for p in rest]
                sums = [self.state[p]["sum"] for p in rest]
                steps = [self.state[p]["step"] for p in rest]
                _functional_adagrad(rest, grads, sums, steps, has_sparse_grad=any(g.is_sparse for g in grads),
                                    foreach=False, lr=group["lr"], weight_decay=group["weight_decay"],
                                    lr_decay=group["lr_decay"], eps=group["eps"], maximize=False)
        with_pen = bool(self._penalties) and not self._pen_off_once
        self._pen_off_once = False
        if with_pen:
            for acc in self._pen_acc.values():
                acc.zero_()  # (a fill kernel: captured with the step)
        for device, items in dense.items():
            with torch.cuda.device(device):
                for i in range(0, len(items), _lib.ADAGRAD_MAX_SEGS):
                    chunk = items[i:i + _lib.ADAGRAD_MAX_SEGS]
                    segs = (_lib.KgeAdagradSeg * len(chunk))(*(c[2] for c in chunk))
                    if with_pen and any(c[0] in self._penalties for c in chunk):
                        pens = []
                        for c in chunk:
                            spec = self._penalties.get(c[0])
                            if spec is None:
                                pens.append(_lib.KgePenaltySeg(0, 0, 0.0, 0, None))
                            else:
                                acc = self._pen_acc[device]
                                pens.append(_lib.KgePenaltySeg(spec[0], spec[1], spec[2], spec[3],
                                                               acc.data_ptr() + 8 * self._pen_slot[c[0]]))
                        _lib.check(_lib.lib().kge_adagrad_step_multi_penalty(
                            segs, (_lib.KgePenaltySeg * len(chunk))(*pens), len(chunk), engine._stream(device)),
                            "kge_adagrad_step_multi_penalty")
                        continue
                    _lib.check(_lib.lib().kge_adagrad_step_multi(segs, len(chunk), engine._stream(device)),
                               "kge_adagrad_step_multi")
            for p, copy, _seg in items:
                torch.autograd.graph.increment_version(p)  # the kernel wrote through the raw pointer
                if copy is not None:
                    setattr(p, BF16_ATTR, (copy, p._version, p.data_ptr()))
        return loss


class Adam(torch.optim.Adam):
    """torch.optim.Adam (same constructor arguments and `state_dict`: "step", "exp_avg", "exp_avg_sq")
    with the dense update of float32 GPU parameters done by kge_adam_step: one pass instead of the
    multi-tensor sequence, optionally writing the bf16 scoring copies in the same pass.  amsgrad,
    maximize, capturable and differentiable are not supported; parameters the kernel does not cover
    are stepped by torch's own Adam."""

    # step() computes lr / (1 - beta1^t) and sqrt(1 - beta2^t) on the host from the step count and hands them to the
    # kernel as launch arguments: a hipGraph capture would freeze them at the capture step.  GraphedStep refuses.
    graph_capturable = False

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False,
                 bf16_copies: bool = False, **kw):
        kw.pop("foreach", None)
        kw.pop("fused", None)
        if amsgrad:
            raise NotImplementedError("kge_amd.optim.Adam: amsgrad is not supported")
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, foreach=False,
                         **kw)
        self.bf16_copies = bool(bf16_copies)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        done, held = [], False
        for group in self.param_groups:
            if group.get("maximize") or group.get("differentiable") or group.get("capturable"):
                raise NotImplementedError("kge_amd.optim.Adam: maximize / differentiable / capturable are not supported")
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not Adagrad._kernel_ok(p):
                    held = True
                    continue
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = torch.tensor(0.0, dtype=torch.float32)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                t = float(state["step"])
                step_size = float(group["lr"]) / (1.0 - beta1 ** t)
                bc2_sqrt = (1.0 - beta2 ** t) ** 0.5
                copy = None
                if self.bf16_copies and p.dim() == 2:
                    rec = getattr(p, BF16_ATTR, None)
                    copy = rec[0] if rec is not None and rec[0].shape == p.shape else \
                        torch.empty(p.shape, dtype=torch.bfloat16, device=p.device)
                with torch.cuda.device(p.device):
                    _lib.check(_lib.lib().kge_adam_step(
                        p.data_ptr(), p.grad.data_ptr(), state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr(),
                        p.numel(), step_size, bc2_sqrt, float(beta1), float(beta2), float(group["weight_decay"]),
                        float(group["eps"]), None if copy is None else copy.data_ptr(), engine._stream(p.device)),
                        "kge_adam_step")
                torch.autograd.graph.increment_version(p)
                if copy is not None:
                    setattr(p, BF16_ATTR, (copy, p._version, p.data_ptr()))
                done.append(p)
        if held:  # torch's own update for everything else: hide the stepped parameters' gradients for the call
            stash = [(q, q.grad) for q in done]
            for q, _ in stash:
                q.grad = None
            try:
                torch.optim.Adam.step(self)
            finally:
                for q, g in stash:
                    q.grad = g
        return loss
