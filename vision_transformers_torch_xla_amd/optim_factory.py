"""Optimizer construction for the ViT training step, with AdamW fused into one HIP launch per parameter group.

Mirrors optim_factory.py of the reference: get_parameter_groups (:70-211, non-TPU branch :151-195: a parameter goes to the
"no_decay" group when it is 1-D, a ".bias", or named by model.no_weight_decay(); groups carry "lr_scale"),
create_optimizer(args, model, get_num_layer, get_layer_scale, filter_bias_and_bn, skip_list) (:214-296) and
LayerDecayValueAssigner (:59-67).  Only the optimizer the reference's scripts select, AdamW (:248-249, main.py:171),
is implemented; other --opt names raise.
"""
from __future__ import annotations

import os

import torch

from . import ops
from ._lib import VitkError


class LayerDecayValueAssigner(object):
    def __init__(self, values, get_layer_id=None):
        self.values = values
        self._get_layer_id = get_layer_id

    def get_scale(self, layer_id):
        return self.values[layer_id]

    def get_layer_id(self, var_name):
        if self._get_layer_id is None:
            raise VitkError("LayerDecayValueAssigner: pass get_layer_id= (the reference only ships a ConvNeXt layer map)")
        return self._get_layer_id(var_name)


def get_parameter_groups(model, weight_decay=1e-5, skip_list=(), get_num_layer=None, get_layer_scale=None):
    """The optimizer's parameter groups, with the reference's membership rule (optim_factory.py:151-195): a trainable
    parameter is exempt from weight decay when it is a vector (norm scales, biases), is named "*.bias", or is listed in
    `skip_list` (the model's no_weight_decay() names); everything else decays with `weight_decay`.  With `get_num_layer`
    the two classes are further split per layer id and each group carries get_layer_scale(layer id) as "lr_scale" (1.0
    otherwise).  Groups are returned in the order their first parameter appears in model.named_parameters(), parameters in
    that order within a group - the order optimizer state dicts are indexed by."""
    groups = {}
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        exempt = param.ndim == 1 or name.endswith(".bias") or name in skip_list
        layer = get_num_layer(name) if get_num_layer is not None else None
        key = ("no_decay" if exempt else "decay") if layer is None else "layer_%d_%s" % (layer, "no_decay" if exempt else "decay")
        group = groups.get(key)
        if group is None:
            group = groups[key] = {"weight_decay": 0.0 if exempt else weight_decay, "params": [],
                                   "lr_scale": get_layer_scale(layer) if get_layer_scale is not None else 1.0}
        group["params"].append(param)
    return list(groups.values())


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction, eps outside the sqrt) executed by
    vitk_adamw_step over contiguous ranges of the model's flat parameter buffer: reads p, g, m, v and writes p, m, v
    plus the bf16 weight copy in one pass.  `param_groups` behave as usual (the training loop overwrites "lr" and
    "weight_decay" every iteration, engine.py:98-103).  Parameters that do not belong to a flat-buffer model are
    updated by one launch each."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, model=None):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("FusedAdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._model = model
        self._steps = 0
        self.grad_scale = 1.0          # extra factor on the gradients (e.g. 1/world_size after a SUM all-reduce)
        self._plan = None
        self._m = self._v = None

    # -- planning: map every param group onto contiguous ranges of the flat buffer
    def _build_plan(self):
        model = self._model
        plan = []
        if model is not None and hasattr(model, "_flat") and not model._views_ok():
            model._flatten()
        flat_ok = model is not None and hasattr(model, "_flat")
        # A re-plan (the model was re-flattened by .to() / .cuda(), or its state was loaded) must not lose the moments
        # gathered so far while the step count - and with it the bias correction - carries on: keep them and copy them
        # into the new flat buffers below.
        carried = {p: (st["exp_avg"].detach().clone(), st["exp_avg_sq"].detach().clone())
                   for p, st in self.state.items() if "exp_avg" in st and "exp_avg_sq" in st}
        if flat_ok:
            flat_p = model._flat["p"]
            base = flat_p.data_ptr()
            self._m = torch.zeros_like(flat_p)
            self._v = torch.zeros_like(flat_p)
            name_of = {id(p): n for n, p in zip(model._pnames, model._plist)}
            order = {n: i for i, n in enumerate(model._pnames)}
        for gi, group in enumerate(self.param_groups):
            ranges, loose = [], []
            if flat_ok:
                inside = sorted((p for p in group["params"] if id(p) in name_of), key=lambda p: order[name_of[id(p)]])
                loose = [p for p in group["params"] if id(p) not in name_of]
                cur = None
                for p in inside:
                    i = order[name_of[id(p)]]
                    start = (p.data_ptr() - base) // 4
                    # the range may run up to the next parameter's offset (alignment padding holds zeros)
                    nxt = model._offs[model._pnames[i + 1]] if i + 1 < len(model._pnames) else model._total
                    if cur is not None and cur[2] == i - 1 and cur[1] == start:
                        cur[1], cur[2] = nxt, i
                    else:
                        cur = [start, nxt, i]
                        ranges.append(cur)
                    self.state[p] = {"step": torch.tensor(float(self._steps)), "exp_avg": self._m[start:start + p.numel()].view(p.shape),
                                     "exp_avg_sq": self._v[start:start + p.numel()].view(p.shape)}
                    if p in carried:
                        for dst, src in zip((self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"]), carried[p]):
                            if tuple(src.shape) != tuple(dst.shape):
                                raise VitkError(f"FusedAdamW: optimizer state of shape {tuple(src.shape)} does not fit parameter "
                                                f"{name_of[id(p)]} of shape {tuple(dst.shape)}")
                            dst.copy_(src.to(dst.device))
            else:
                loose = list(group["params"])
            for p in loose:
                if p not in self.state or "exp_avg" not in self.state[p]:
                    self.state[p] = {"step": torch.tensor(0.0), "exp_avg": torch.zeros_like(p, dtype=torch.float32),
                                     "exp_avg_sq": torch.zeros_like(p, dtype=torch.float32)}
            plan.append(([(r[0], r[1]) for r in ranges], loose))
        self._plan = plan
        self._plan_key = (model._flat["p"].data_ptr() if flat_ok else None, tuple(len(g["params"]) for g in self.param_groups))

    def _plan_stale(self):
        if self._plan is None:
            return True
        model = self._model
        key = (model._flat["p"].data_ptr() if (model is not None and hasattr(model, "_flat")) else None,
               tuple(len(g["params"]) for g in self.param_groups))
        return key != self._plan_key

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._plan_stale():
            self._build_plan()
        self._steps += 1
        model = self._model
        if model is not None and hasattr(model, "_await_transposes"):
            model._await_transposes()            # the kernels below rewrite the bf16 weights a deferred transposition may still read
        for group, (ranges, loose) in zip(self.param_groups, self._plan):
            b1, b2 = group["betas"]
            kw = dict(lr=group["lr"], beta1=b1, beta2=b2, eps=group["eps"], weight_decay=group["weight_decay"], step=self._steps,
                      grad_scale=self.grad_scale)
            for (a, b) in ranges:
                f = model._flat
                ops.adamw_step(f["p"][a:b], model._ensure_grad_buffer()[a:b], self._m[a:b], self._v[a:b], p_bf16=f["b"][a:b], **kw)
            for p in loose:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise VitkError("FusedAdamW: parameters must be on the GPU")
                st = self.state[p]
                ops.adamw_step(p.data.view(-1), p.grad.contiguous().view(-1), st["exp_avg"].view(-1), st["exp_avg_sq"].view(-1), **kw)
        stamp = torch.tensor(float(self._steps))            # one tensor for every parameter's "step" entry
        for st in self.state.values():
            if "step" in st:
                st["step"] = stamp
        if model is not None and hasattr(model, "_refresh_transposes"):
            # bf16 copies were written by the kernel itself; the transposed ones go to the side stream (VITK_DEFER_WT=0: this stream)
            model._refresh_transposes(defer=os.environ.get("VITK_DEFER_WT", "1") != "0")
            model._weights_version = model._version_key()
        return loss

    def zero_grad(self, set_to_none: bool = True):
        """For the flat-buffer model this only marks the gradient buffer stale (the next backward overwrites it)."""
        model = self._model
        if model is not None and hasattr(model, "_flat"):
            model.zero_grad(set_to_none=False)
            own = {id(p) for p in model._plist}
            for group in self.param_groups:
                for p in group["params"]:
                    if id(p) not in own and p.grad is not None:
                        p.grad = None if set_to_none else p.grad.zero_()
            return
        super().zero_grad(set_to_none=set_to_none)

    def load_state_dict(self, state_dict):
        """Restores torch-AdamW-format state into the flat moment buffers."""
        steps = [float(s["step"]) for s in state_dict.get("state", {}).values() if "step" in s]
        if self._plan_stale():
            self._build_plan()
        views = {id(p): (st["exp_avg"], st["exp_avg_sq"]) for p, st in self.state.items() if "exp_avg" in st}
        super().load_state_dict(state_dict)
        with torch.no_grad():
            for p, st in self.state.items():
                if id(p) in views and "exp_avg" in st:
                    m, v = views[id(p)]
                    if tuple(st["exp_avg"].shape) != tuple(m.shape) or tuple(st["exp_avg_sq"].shape) != tuple(v.shape):
                        # state dicts are matched by parameter INDEX: a checkpoint of a model with another parameter order
                        # or size must not be broadcast into the wrong moments
                        raise VitkError(f"FusedAdamW.load_state_dict: moment of shape {tuple(st['exp_avg'].shape)} does not fit a "
                                        f"parameter of shape {tuple(m.shape)} (the checkpoint belongs to another model layout)")
                    m.copy_(st["exp_avg"]); v.copy_(st["exp_avg_sq"])
                    st["exp_avg"], st["exp_avg_sq"] = m, v
        if steps:
            self._steps = int(max(steps))


def create_optimizer(args, model, get_num_layer=None, get_layer_scale=None, filter_bias_and_bn=True, skip_list=None):
    """args.opt / lr / weight_decay / opt_eps / opt_betas -> optimizer, with the reference's call contract
    (optim_factory.py:214-296).  filter_bias_and_bn: decay lives in the parameter groups (the exempt group gets 0), so the
    optimizer-wide default is 0; otherwise one group over all parameters with args.weight_decay.  An "a_b_adamw" style name
    selects by its last part.  AdamW - the only optimizer the reference's recipes use (main.py:171) - is the fused one."""
    if filter_bias_and_bn:
        exempt_names = skip_list if skip_list is not None else (model.no_weight_decay() if hasattr(model, "no_weight_decay") else ())
        params = get_parameter_groups(model, args.weight_decay, exempt_names, get_num_layer, get_layer_scale)
        default_decay = 0.0
    else:
        params, default_decay = model.parameters(), args.weight_decay
    hyper = {"lr": args.lr, "weight_decay": default_decay}
    for attr, key, conv in (("opt_eps", "eps", float), ("opt_betas", "betas", tuple)):
        if getattr(args, attr, None) is not None:
            hyper[key] = conv(getattr(args, attr))
    kind = args.opt.lower().split("_")[-1]
    if kind not in ("adamw", "fusedadamw"):
        raise VitkError(f"optimizer {args.opt!r} is not implemented: the ViT recipes of the reference use adamw (main.py:171)")
    return FusedAdamW(params, model=model if hasattr(model, "_flat") else None, **hyper)
