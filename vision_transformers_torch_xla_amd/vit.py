"""Vision Transformer whose forward AND backward run as hand-written gfx950 HIP kernels (libvitk), behind the
reference's timm-style surface.

Mirrors (reference file:line, under /root/reference):
  models/vision_transformer.py:444-623  VisionTransformer constructor rules (use_fc_norm :529, dpr linspace :581,
                                        norm / fc_norm placement :603,616), :634-648 + :998-1010 weight init,
                                        :666-668 no_weight_decay(), :743-780 _pos_embed, :934-995 forward, :419-441 pool
  models/vision_transformer.py:109-178  Block: x += dp(attn(ln1 x)); x += dp(mlp(ln2 x))
  models/deit.py:28-119                 distilled variant (dist_token, head_dist, two prefix tokens)
  models/_factory.py:46-155             create_model(name, pretrained=False, **kwargs), unknown name -> RuntimeError
  models/vision_transformer.py:2690-2860, models/my_vit.py:84-165   model size table
State-dict keys are timm's (cls_token, pos_embed, patch_embed.proj.*, blocks.N.norm1.*, blocks.N.attn.qkv.*, ...), so
checkpoints move between the reference and this model unchanged.

Design: all parameters live in ONE flat fp32 buffer (master weights) with a same-layout flat fp32 gradient buffer
and a flat bf16 copy that the MFMA GEMMs read; nn.Parameters are views.  The layout is
[decay weights: head, blocks L-1..0, patch | no-decay: tokens, pos, biases, norms] so that (a) fused AdamW is two
launches, (b) gradients become final in increasing address order during backward, which is what the data-parallel
reducer buckets on.  One torch.autograd.Function spans the whole network; its backward writes straight into the flat
gradient buffer (param.grad are views of it).
There is no CPU path: tensors must live on the GPU and libvitk.so must be built.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
import torch.nn as nn

from . import ops
from ._lib import (EPI_BIAS, EPI_BIAS_F32, EPI_BIAS_GELU_DG, EPI_BIAS_RES, EPI_MUL, EPI_PATCH, EPI_PLAIN, VitkError)

bf16, f32 = torch.bfloat16, torch.float32
_ALIGN = 64   # flat-buffer offsets are multiples of 64 elements (256 B fp32 / 128 B bf16)


class _Linear(nn.Module):
    def __init__(self, in_f, out_f):
        super().__init__()
        self.in_features, self.out_features = in_f, out_f
        self.weight = nn.Parameter(torch.empty(out_f, in_f))
        self.bias = nn.Parameter(torch.zeros(out_f))


class _LayerNorm(nn.Module):
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class _Attention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = _Linear(dim, 3 * dim)
        self.proj = _Linear(dim, dim)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = _Linear(dim, hidden)
        self.fc2 = _Linear(hidden, dim)


class Block(nn.Module):
    """Parameter holder of one pre-norm block (models/vision_transformer.py:109-178); the math runs in libvitk."""

    def __init__(self, dim, heads, mlp_ratio, drop_path):
        super().__init__()
        self.norm1 = _LayerNorm(dim)
        self.attn = _Attention(dim, heads)
        self.norm2 = _LayerNorm(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.drop_path = float(drop_path)


class _PatchProj(nn.Module):
    def __init__(self, in_chans, dim, patch):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(dim, in_chans, patch, patch))
        self.bias = nn.Parameter(torch.zeros(dim))


class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch, in_chans, dim):
        super().__init__()
        self.img_size, self.patch_size = (img_size, img_size), (patch, patch)
        self.grid_size = (img_size // patch, img_size // patch)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = _PatchProj(in_chans, dim, patch)


class _Branch:
    """Stochastic depth of ONE residual branch for one step.  `scale` (fp32 [B] on the GPU) is 0 for the samples that drop the
    branch and 1/keep for the others.  timm's DropPath computes the branch for every sample and multiplies by that factor
    (models/vision_transformer.py:161,173,176-177); here, when `compact` is set, the branch is computed for the kept samples
    only - LayerNorm gathers their rows into a compact batch, the projection GEMM scatters its residual epilogue back, the
    dropped samples' rows are copied - which gives the same activations and gradients (0 * branch = 0) for less work.
    kept / dropped: int32 sample lists; inv[sample] = index in kept or -1; scale_c = scale[kept]."""
    __slots__ = ("scale", "compact", "nk", "kept", "inv", "dropped", "scale_c")

    @staticmethod
    def from_masks(mask, keep, device, compact, ring):
        """mask bool [R, B] and keep fp32 [R] on the host -> R branches; two host->device copies in all"""
        R, B = mask.shape
        nk = mask.sum(1).tolist()
        order = torch.argsort((~mask).to(torch.int8), dim=1, stable=True).to(torch.int32)       # kept samples first, ascending
        inv = torch.where(mask, torch.cumsum(mask, 1, dtype=torch.int32) - 1, torch.full((), -1, dtype=torch.int32))
        scale = mask.to(f32) / keep[:, None]
        ints, flts = ring.next(2 * R * B)
        ints[:R * B].view(R, B).copy_(order); ints[R * B:].view(R, B).copy_(inv)
        flts[:R * B].view(R, B).copy_(scale); flts[R * B:].view(R, B).copy_((1.0 / keep)[:, None].expand(R, B))
        if torch.device(device).type == "cuda" and (2 * R * B) % 4 == 0:
            gi = ops.copy_from_pinned(ints, device)
            gf = ops.copy_from_pinned(flts, device)
        else:
            gi = ints.to(device, non_blocking=True)
            gf = flts.to(device, non_blocking=True)
        ring.copied()
        out = []
        for r in range(R):
            b = _Branch()
            b.scale, b.nk = gf[r * B:(r + 1) * B], int(nk[r])
            b.compact = bool(compact and b.nk < B)
            b.kept = gi[r * B:r * B + b.nk]
            b.dropped = gi[r * B + b.nk:(r + 1) * B]
            b.inv = gi[(R + r) * B:(R + r + 1) * B]
            b.scale_c = gf[(R + r) * B:(R + r) * B + b.nk]
            out.append(b)
        return out

    @staticmethod
    def from_scale(scale_cpu, device, compact):
        """an explicit per-sample factor (tests): samples with factor 0 drop the branch"""
        b = _Branch()
        mask = scale_cpu != 0
        B = scale_cpu.numel()
        b.scale, b.nk = scale_cpu.to(device).contiguous(), int(mask.sum())
        b.compact = bool(compact and b.nk < B)
        order = torch.argsort((~mask).to(torch.int8), stable=True).to(torch.int32)
        b.kept, b.dropped = order[:b.nk].to(device), order[b.nk:].to(device)
        b.inv = torch.where(mask, torch.cumsum(mask, 0, dtype=torch.int32) - 1, torch.full((), -1, dtype=torch.int32)).to(device)
        b.scale_c = scale_cpu[mask].to(device).contiguous()
        return b


_WGRAD_SIDE = {}     # device index -> (side stream of the weight-gradient GEMMs, its workspace)


class _PinRing:
    """A few pinned host buffers reused round-robin for the per-step index tables.  The copies that read them are asynchronous
    and the host may run steps ahead of the GPU, so a slot is reused only after the event recorded behind its copies has
    completed (a host-side wait that only triggers when the host is a whole ring ahead)."""

    def __init__(self, slots=8):
        self.slots, self.i, self.bufs, self.events, self.last = slots, 0, {}, {}, None

    def next(self, n):
        key = (n, self.i % self.slots)
        self.i += 1
        if key not in self.bufs:
            pin = torch.cuda.is_available()
            self.bufs[key] = (torch.empty(n, dtype=torch.int32, pin_memory=pin), torch.empty(n, dtype=f32, pin_memory=pin))
        ev = self.events.get(key)
        if ev is not None:
            ev.synchronize()
        self.last = key
        return self.bufs[key]

    def copied(self):
        """call after the copies out of the slot handed out last have been enqueued"""
        if torch.cuda.is_available() and self.last is not None:
            ev = self.events.get(self.last)
            if ev is None:
                ev = self.events[self.last] = torch.cuda.Event()
            ev.record()


class _ViTFunction(torch.autograd.Function):
    """forward = all layers through libvitk, keeping what backward needs; backward fills the flat gradient buffer."""

    @staticmethod
    def forward(ctx, img, model, scales, *params):
        outs, saved = model._forward_impl(img, scales, save=True)
        ctx.model, ctx.saved, ctx.nout = model, saved, len(outs)
        return outs[0] if len(outs) == 1 else tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        model, saved = ctx.model, ctx.saved
        ctx.saved = None
        model._backward_impl(saved, list(douts))
        return (None, None, None) + (None,) * len(model._plist)


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, global_pool="token", embed_dim=768,
                 depth=12, num_heads=12, mlp_ratio=4.0, qkv_bias=True, drop_path_rate=0.0, distilled=False,
                 weight_init="", _lazy_grad=False, **kwargs):
        super().__init__()
        # what rebuilds an identically shaped model (copy.deepcopy goes through the constructor, see __deepcopy__)
        self._ctor = dict(img_size=img_size, patch_size=patch_size, in_chans=in_chans, num_classes=num_classes, global_pool=global_pool,
                          embed_dim=embed_dim, depth=depth, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                          drop_path_rate=drop_path_rate, distilled=distilled)
        self._lazy_grad = bool(_lazy_grad)       # the flat gradient buffer is allocated by the first backward, not up front
        unsupported = {k: v for k, v in kwargs.items() if v not in (None, False, 0, 0.0, "")}
        for k in ("class_token", "final_norm", "pretrained_cfg", "pretrained_cfg_overlay", "cache_dir"):
            unsupported.pop(k, None)
        if unsupported:
            raise VitkError(f"VisionTransformer: options outside the ViT training hot path are not implemented: {sorted(unsupported)}")
        if global_pool not in ("avg", "token"):
            raise VitkError(f"global_pool={global_pool!r} not implemented (have 'avg', 'token')")
        if distilled and global_pool != "token":
            raise VitkError("distilled models require global_pool='token' (models/deit.py:38)")
        if not qkv_bias:
            raise VitkError("qkv_bias=False not implemented")
        if embed_dim % num_heads or (embed_dim // num_heads) % 8 or embed_dim // num_heads > 96:
            raise VitkError(f"head dim must be a multiple of 8 up to 96 (embed_dim={embed_dim}, heads={num_heads}: {embed_dim / num_heads:g})")
        if patch_size % 8 or img_size % patch_size:
            raise VitkError("patch_size must be a multiple of 8 and divide img_size")
        if num_classes < 1:
            raise VitkError("num_classes must be positive (a headless model is not implemented)")
        if embed_dim > 1024 or embed_dim % 8:
            raise VitkError("embed_dim must be a multiple of 8 and <= 1024 (LayerNorm kernel)")
        self.num_classes, self.global_pool, self.distilled = num_classes, global_pool, distilled
        # The head GEMMs want a class count that is a multiple of 8 (16-byte rows of the bf16 operands).  Any other count - the
        # reference's --nb_classes is free, and fine-tuning onto 37 or 100 classes is its stated use (main.py:653-674) - runs on
        # ZERO-PADDED bf16 copies of the head weights (padding rows zero, padding bias entries zero): the padded logits are cut
        # off before anything reads them, the padded columns of the logit gradient are zero, and the weight / bias gradients of the
        # real rows are copied out of the padded product.  Parameters, gradients, optimizer state and state dicts keep timm's shapes.
        self._cls_pad = (num_classes + 7) // 8 * 8
        self._pad_head = self._cls_pad != num_classes
        self.num_features = self.head_hidden_size = self.embed_dim = embed_dim
        self.depth, self.num_heads, self.patch_size, self.in_chans = depth, num_heads, patch_size, in_chans
        self.num_prefix_tokens = 2 if distilled else 1
        self.has_class_token = True
        self.drop_path_rate = drop_path_rate
        self.distilled_training = False
        use_fc_norm = global_pool == "avg"                                   # vision_transformer.py:529

        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.num_patches = self.patch_embed.num_patches
        self.num_tokens = self.num_patches + self.num_prefix_tokens
        self.head_dim = embed_dim // num_heads
        # Head dims 64 and 96 run on the MFMA attention kernels (64: up to 608 tokens, 96: up to 224).  Other head dims up to 96
        # (48: my_vit_mini, 72: my_vit_xs, models/my_vit.py:84-125) run on them too, ZERO-PADDED to 64 / 96: the qkv and proj GEMMs
        # read bf16 weight copies with that many rows / columns per head, the padding rows, columns and bias entries zero, so the
        # padding lanes of q, k, v, the attention output and every gradient that flows through them are exactly zero and no
        # result changes; parameters, gradients and state dicts keep timm's shapes.  Beyond 224 tokens a head between 64 and 96
        # wide takes the vector-arithmetic kernels (up to 256 tokens), unpadded.
        hd = self.head_dim
        self._hd_pad = 64 if hd <= 64 else (96 if self.num_tokens <= 224 else hd)
        self._pad_heads = self._hd_pad != hd
        limit = 608 if hd <= 64 else 256
        if self.num_tokens > limit:
            raise VitkError(f"{self.num_tokens} tokens exceed the attention kernels' limit for head dim {self.head_dim} ({limit})")
        # registration order = timm's named_parameters() order (cls_token, pos_embed, dist_token: models/deit.py:41-43 adds the
        # distillation token after the base class registered the other two), which optimizer state dicts are matched by
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.num_tokens, embed_dim))
        if distilled:
            self.dist_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]  # vision_transformer.py:581
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, dpr[i]) for i in range(depth)])
        if use_fc_norm:
            self.fc_norm = _LayerNorm(embed_dim)                             # :616
        else:
            self.norm = _LayerNorm(embed_dim)                                # :603
        self.head = _Linear(embed_dim, num_classes)
        if distilled:
            self.head_dist = _Linear(embed_dim, num_classes)
        self._forced_drop_scales = None      # tests inject per-layer DropPath factors here
        self._compact_droppath = os.environ.get("VITK_DROPPATH_COMPACT", "1") != "0"   # compute a branch for its kept samples only
        self._pin_ring = _PinRing()
        self._branch_samples = [0, 0, 0, 0]  # samples a residual branch was computed for / could have been (benchmark accounting)
        self._bwd_taps = None                # tests: dict that receives per-layer intermediate gradients
        self._grad_ready_hook = None         # data-parallel reducer: called with the flat offset below which grads are final
        self._static_drop = None             # stochastic depth from tables at fixed device addresses (a captured step replays them)
        self._ws = ops.Workspace()
        self._flat = None
        if weight_init != "skip":
            self.init_weights()
        self._flatten()

    # ------------------------------------------------------------------------------------------- parameters
    def init_weights(self, mode: str = ""):
        """timm-mode init: trunc-normal(.02) for pos_embed and every Linear weight, zero biases, cls ~ N(0, 1e-6)
        (vision_transformer.py:634-648, 998-1010; deit.py:49-51).  The patch Conv2d keeps torch's default init."""
        with torch.no_grad():
            nn.init.trunc_normal_(self.pos_embed, std=0.02, a=-2.0, b=2.0)
            nn.init.normal_(self.cls_token, std=1e-6)
            if self.distilled:
                nn.init.trunc_normal_(self.dist_token, std=0.02, a=-2.0, b=2.0)
            for m in self.modules():
                if isinstance(m, _Linear):
                    nn.init.trunc_normal_(m.weight, std=0.02, a=-2.0, b=2.0)
                    nn.init.zeros_(m.bias)
            w = self.patch_embed.proj.weight
            nn.init.kaiming_uniform_(w, a=math.sqrt(5))                      # nn.Conv2d.reset_parameters
            bound = 1.0 / math.sqrt(w[0].numel())
            nn.init.uniform_(self.patch_embed.proj.bias, -bound, bound)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_embed", "cls_token", "dist_token"}                      # vision_transformer.py:668

    def set_distilled_training(self, enable=True):
        self.distilled_training = enable                                     # deit.py:72-73

    def get_classifier(self):
        return (self.head, self.head_dist) if self.distilled else self.head

    def _flat_order(self):
        """[(name, param)] in flat-buffer order and the number of leading 'decay' entries.  Decay = what
        optim_factory.get_parameter_groups puts in its weight-decay group (optim_factory.py:164)."""
        named = dict(self.named_parameters())
        skip = self.no_weight_decay()
        is_nd = lambda n, p: p.dim() == 1 or n.endswith(".bias") or n in skip
        decay = ["head.weight"] + (["head_dist.weight"] if self.distilled else [])
        for i in reversed(range(self.depth)):
            decay += [f"blocks.{i}.mlp.fc2.weight", f"blocks.{i}.mlp.fc1.weight", f"blocks.{i}.attn.proj.weight", f"blocks.{i}.attn.qkv.weight"]
        decay += ["patch_embed.proj.weight"]
        assert all(not is_nd(n, named[n]) for n in decay)
        first_nd = ["cls_token"] + (["dist_token"] if self.distilled else []) + ["pos_embed"]
        rest = [n for n, p in named.items() if n not in decay and n not in first_nd]
        assert all(is_nd(n, named[n]) for n in first_nd + rest), "unexpected weight-decay parameter outside the layout"
        order = decay + first_nd + rest
        return [(n, named[n]) for n in order], len(decay)

    def _flatten(self, device=None):
        """(Re)build the flat master / gradient / bf16 buffers on `device` and re-point every Parameter into them."""
        order, ndecay = self._flat_order()
        device = torch.device(device) if device is not None else order[0][1].device
        offs, off = {}, 0
        for k, (n, p) in enumerate(order):
            if k == ndecay:
                decay_end = off
            offs[n] = off
            nel = p.numel()
            # cls/dist tokens must be adjacent (they are read as one [prefix, D] table): D is a multiple of 8 and
            # D*4 bytes keeps 16-byte alignment, so no padding between them
            pad = nel if n in ("cls_token",) and self.distilled else ((nel + _ALIGN - 1) // _ALIGN) * _ALIGN
            off += pad
        total = off
        flat_p = torch.zeros(total, dtype=f32, device=device)
        flat_g = None if self._lazy_grad else torch.zeros(total, dtype=f32, device=device)
        with torch.no_grad():
            for n, p in order:
                flat_p[offs[n]:offs[n] + p.numel()].copy_(p.detach().reshape(-1).to(device=device, dtype=f32))
        self._offs, self._total, self._decay_end = offs, total, decay_end
        self._flat = {"p": flat_p, "g": flat_g, "b": torch.zeros(total, dtype=bf16, device=device)}
        self._plist, self._gview = [], {}
        for n, p in order:
            sl = slice(offs[n], offs[n] + p.numel())
            p.data = flat_p[sl].view(p.shape)
            p.grad = None
            self._plist.append(p)
        self._pnames = [n for n, _ in order]
        self._grad_valid = False
        # transposed weight copies and their descriptor table belong to THIS bf16 buffer: a table kept across a re-flatten
        # would name freed memory if the allocator ever handed the old address out again
        pending = getattr(self, "_wt_ready", None)
        if pending is not None:     # a transposition queued on the side stream still writes the copies dropped below
            torch.cuda.current_stream().wait_event(pending)
        self._wt = {}
        self._wpad = {}
        self._hpad = {}
        self._wt_plan = None
        self._wt_plan_key = None
        self._wt_ready = None       # event behind a transposition that runs on the side stream; backward waits for it
        self._weights_version = None
        # bucket boundaries for gradient reduction: grads below boundary k are final once stage k of backward is done
        self._layer_end = {}
        for i in range(self.depth):
            n = f"blocks.{i}.attn.qkv.weight"
            self._layer_end[i] = offs[n] + dict(order)[n].numel()
        n = "head_dist.weight" if self.distilled else "head.weight"
        self._head_end = offs[n] + dict(order)[n].numel()
        self._build_views()

    def _apply(self, fn, recurse=True):
        super()._apply(fn)
        if self._flat is not None:
            self._flatten()
        return self

    def zero_grad(self, set_to_none: bool = True):
        """Marks the flat gradient buffer stale: the next backward overwrites it instead of accumulating (no memset)."""
        self._grad_valid = False
        if set_to_none:
            for p in self._plist:
                p.grad = None

    def _version_key(self):
        return sum(p._version for p in self._plist)

    def _sync_weights(self, force=False):
        """bf16 copy of every weight + transposed copies for the data-gradient GEMMs, refreshed when the fp32 masters
        changed through torch (load_state_dict, a foreign optimizer).  FusedAdamW refreshes them itself."""
        if not self._views_ok():            # e.g. after copy.deepcopy(model): parameters no longer alias the flat buffer
            self._flatten()
        key = self._version_key()
        if not force and self._weights_version == key:
            return
        if not self._flat["p"].is_cuda:
            raise VitkError("the model must be on the GPU (model.cuda()): libvitk is the only implementation")
        self._await_transposes()
        ops.cast_f32_to_bf16(self._flat["p"], self._flat["b"])
        self._refresh_transposes()
        self._weights_version = key

    def _await_transposes(self):
        """A transposition that an optimizer step left on the side stream READS the bf16 weight buffer: whoever overwrites that
        buffer next (another optimizer step with no backward in between, a load_state_dict followed by a forward pass) first
        makes the current stream wait for it."""
        if self._wt_ready is not None:
            torch.cuda.current_stream().wait_event(self._wt_ready)
            self._wt_ready = None

    def _build_views(self):
        """2-D views into the flat buffers: bf16 GEMM weights [out, in], the prefix-token table, the position table."""
        self._wb = {}
        for n, p in zip(self._pnames, self._plist):
            if p.dim() >= 2 and n not in ("cls_token", "dist_token", "pos_embed"):
                off, rows = self._offs[n], p.shape[0]
                self._wb[n] = self._flat["b"][off:off + p.numel()].view(rows, p.numel() // rows)
        D, npre = self.embed_dim, self.num_prefix_tokens
        o = self._offs["cls_token"]
        self._tokens = self._flat["p"][o:o + npre * D].view(npre, D)
        o = self._offs["pos_embed"]
        self._pos2d = self._flat["p"][o:o + self.num_tokens * D].view(self.num_tokens, D)
        self._gview, self._gtokens, self._gpos2d = {}, None, None
        if self._flat["g"] is not None:
            self._build_grad_views()

    def _build_grad_views(self):
        flat_g, D, npre = self._flat["g"], self.embed_dim, self.num_prefix_tokens
        self._gview = {n: flat_g[self._offs[n]:self._offs[n] + p.numel()].view(p.shape) for n, p in zip(self._pnames, self._plist)}
        o = self._offs["cls_token"]
        self._gtokens = flat_g[o:o + npre * D].view(npre, D)
        o = self._offs["pos_embed"]
        self._gpos2d = flat_g[o:o + self.num_tokens * D].view(self.num_tokens, D)

    def _ensure_grad_buffer(self):
        """the flat fp32 gradient buffer of a model built with a lazy one (deep copies: an EMA copy never trains)"""
        if self._flat["g"] is None:
            self._flat["g"] = torch.zeros(self._total, dtype=f32, device=self._flat["p"].device)
            self._build_grad_views()
        return self._flat["g"]

    def __deepcopy__(self, memo):
        """copy.deepcopy(model) (timm ModelEma does it, engine.py:198-202 / main.py:744-754): a fresh model of the same
        configuration with the same parameter VALUES, training flags and requires_grad flags.  Runtime objects are not
        copied but rebuilt empty - pinned-table ring and its events, workspaces, the side stream, the transposed-weight
        table, a data-parallel reducer's hook (the copy is not wrapped) - and the gradient buffer of the copy is allocated
        only if it is ever trained."""
        new = VisionTransformer(**self._ctor, weight_init="skip", _lazy_grad=True)
        memo[id(self)] = new
        new._flatten(self._flat["p"].device)
        with torch.no_grad():
            new._flat["p"].copy_(self._flat["p"])
        for pn, po in zip(new._plist, self._plist):
            pn.requires_grad_(po.requires_grad)
        new.train(self.training)
        new.distilled_training = self.distilled_training
        new._compact_droppath = self._compact_droppath
        return new

    def _refresh_transposes(self, defer=False):
        """[in, out] bf16 copies of the linears: the data-gradient GEMM dX = dY @ W runs as gemm_nt(dY, W^T).
        defer: nothing before the next backward reads the copies, so the optimizer asks for them on the side stream of the
        weight-gradient GEMMs (idle during the forward pass), where the launch fills compute units that the forward GEMMs' last
        tile rounds leave empty; the next backward waits for `_wt_ready`."""
        key = (self._flat["b"].data_ptr(), str(self._flat["b"].device))
        H, hd, hp = self.num_heads, self.head_dim, self._hd_pad
        cur = torch.cuda.current_stream()
        if self._wt_ready is not None:           # a transposition still queued on the side stream reads the padded copies rewritten
            cur.wait_event(self._wt_ready)       # below and writes the transposed ones: wait for it before touching either
            self._wt_ready = None
        if getattr(self, "_wt_plan_key", None) != key:
            pairs = []
            self._wpad = {}
            self._hpad = {}
            for n, src in self._wb.items():
                if n == "patch_embed.proj.weight":      # the image needs no gradient
                    continue
                if self._pad_head and n in ("head.weight", "head_dist.weight"):          # [C, D] -> [Cp, D], zero rows behind
                    src = self._hpad[n] = torch.zeros((self._cls_pad, src.shape[1]), dtype=bf16, device=src.device)
                    self._hpad[n[:-6] + "bias"] = torch.zeros(self._cls_pad, dtype=f32, device=src.device)
                if self._pad_heads and n.endswith("attn.qkv.weight"):        # [3 H hd, D] -> [3 H hp, D], zero rows between
                    src = self._wpad[n] = torch.zeros((3 * H * hp, src.shape[1]), dtype=bf16, device=src.device)
                    self._wpad[n[:-6] + "bias"] = torch.zeros(3 * H * hp, dtype=f32, device=src.device)
                elif self._pad_heads and n.endswith("attn.proj.weight"):     # [D, H hd] -> [D, H hp], zero columns between
                    src = self._wpad[n] = torch.zeros((src.shape[0], H * hp), dtype=bf16, device=src.device)
                dst = torch.empty((src.shape[1], src.shape[0]), dtype=bf16, device=src.device)
                self._wt[n] = dst
                pairs.append((src, dst))
            self._wt_plan = ops.transpose_plan(pairs)   # one launch for all of them
            self._wt_plan_key = key
        if self._pad_heads:
            # refresh the padded copies (strided device copies of a few hundred KB; torch's copy kernels)
            named = dict(zip(self._pnames, self._plist))
            for n, dst in self._wpad.items():
                if n.endswith("qkv.weight"):
                    dst.view(3 * H, hp, -1)[:, :hd].copy_(self._wb[n].view(3 * H, hd, -1))
                elif n.endswith("qkv.bias"):
                    dst.view(3 * H, hp)[:, :hd].copy_(named[n].detach().view(3 * H, hd))
                else:
                    dst.view(-1, H, hp)[:, :, :hd].copy_(self._wb[n].view(-1, H, hd))
        if self._pad_head:
            named = dict(zip(self._pnames, self._plist))
            C = self.num_classes
            for n, dst in self._hpad.items():
                dst[:C].copy_(self._wb[n] if n.endswith("weight") else named[n].detach())
        # (a captured step that is replayed does not see events recorded outside its graph: no deferral then)
        side = self._wgrad_stream() if (defer and self._static_drop is None) else None
        if side is None:
            ops.transpose_bf16_batched(self._wt_plan)
            return
        ev = torch.cuda.Event()
        ev.record(cur)                           # the bf16 weights are complete once everything queued so far has run
        side.wait_event(ev)
        with torch.cuda.stream(side):
            ops.transpose_bf16_batched(self._wt_plan)
            self._wt_ready = torch.cuda.Event()
            self._wt_ready.record(side)

    def _views_ok(self):
        p0, off0 = self._plist[0], self._offs[self._pnames[0]]
        p1, off1 = self._plist[-1], self._offs[self._pnames[-1]]
        base = self._flat["p"].data_ptr()
        return p0.data_ptr() == base + 4 * off0 and p1.data_ptr() == base + 4 * off1

    # ---------------------------------------------------------------------------------------------- forward
    def _draw_drop_scales(self, B, device):
        """DropPath (timm drop_path, scale_by_keep): per sample Bernoulli(keep)/keep, one draw per residual branch.
        Returns one (_Branch | None, _Branch | None) pair per block.  The draws are made on the HOST (torch's CPU generator, so
        torch.manual_seed governs them) because the launch shapes depend on them: a branch is computed for the samples that
        keep it only (see _Branch); all index tables of a step travel to the GPU in two asynchronous copies."""
        if self._forced_drop_scales is not None:
            return [tuple(None if s is None else _Branch.from_scale(s.to(dtype=f32).cpu(), device, self._compact_droppath)
                          for s in pair) for pair in self._forced_drop_scales]
        rates = [blk.drop_path if self.training else 0.0 for blk in self.blocks]
        active = [i for i, r in enumerate(rates) if r > 0.0]
        out = [(None, None)] * self.depth
        st = self._branch_samples          # [attention computed, attention total, MLP computed, MLP total] since the last reset
        st[1] += B * self.depth; st[3] += B * self.depth
        st[0] += B * (self.depth - len(active)); st[2] += B * (self.depth - len(active))
        if not active:
            return out
        keep = torch.tensor([1.0 - rates[i] for i in active for _ in range(2)], dtype=f32)          # [2A]
        mask = torch.rand(len(keep), B) < keep[:, None]
        plans = _Branch.from_masks(mask, keep, device, self._compact_droppath, self._pin_ring)
        for k, i in enumerate(active):
            out[i] = (plans[2 * k], plans[2 * k + 1])
            st[0] += plans[2 * k].nk if plans[2 * k].compact else B
            st[2] += plans[2 * k + 1].nk if plans[2 * k + 1].compact else B
        return out

    # -- stochastic depth for a step that is captured in a HIP graph and replayed
    def begin_static_droppath(self, B, device):
        """From now on training forwards of batch size B read their per-sample DropPath factors from ONE device table at a fixed
        address instead of drawing them: launch shapes no longer depend on the draws (every branch is computed for every sample
        and scaled, timm's own arithmetic), so a captured forward + backward can be replayed.  The caller draws new factors with
        refresh_static_droppath() before every step, the first included (one torch.rand per step, exactly as the per-step path
        draws, so both see the same random stream).  end_static_droppath() returns to per-step draws."""
        rates = [blk.drop_path for blk in self.blocks]
        active = [i for i, r in enumerate(rates) if r > 0.0]
        table = torch.zeros(max(2 * len(active) * B, 4), dtype=f32, device=device)
        scales = [(None, None)] * self.depth
        for k, i in enumerate(active):
            pair = []
            for j in range(2):
                b = _Branch()
                b.scale, b.nk, b.compact = table[(2 * k + j) * B:(2 * k + j + 1) * B], B, False
                b.kept = b.dropped = b.inv = b.scale_c = None
                pair.append(b)
            scales[i] = tuple(pair)
        self._static_drop = dict(B=B, table=table, scales=scales, keep=torch.tensor([1.0 - rates[i] for i in active for _ in range(2)], dtype=f32))

    def refresh_static_droppath(self):
        """new Bernoulli(keep) / keep factors for every residual branch, written into the static table by one copy kernel (the
        draws come from torch's CPU generator, as in the per-step path)"""
        sd = self._static_drop
        keep, B = sd["keep"], sd["B"]
        if keep.numel() == 0:
            return
        mask = torch.rand(len(keep), B) < keep[:, None]
        _, flts = self._pin_ring.next(max(len(keep) * B, 4))
        flts[:len(keep) * B].view(len(keep), B).copy_(mask.to(f32) / keep[:, None])
        if sd["table"].is_cuda and flts.is_pinned() and (len(keep) * B) % 4 == 0:
            ops.copy_from_pinned(flts, sd["table"].device, out=sd["table"])
        else:
            sd["table"].copy_(flts[:sd["table"].numel()], non_blocking=True)
        self._pin_ring.copied()

    def end_static_droppath(self):
        self._static_drop = None

    def is_patch_matrix(self, x) -> bool:
        """x is the bf16 patch matrix [B * num_patches, in_chans * patch^2] the device input pipeline writes for the model
        (ops.preprocess_images / ops.resample_images with patch=...) rather than an image batch"""
        return (x.dim() == 2 and x.dtype == bf16 and x.shape[1] == self.in_chans * self.patch_size ** 2 and x.shape[0] > 0
                and x.shape[0] % self.num_patches == 0)

    def forward(self, x: torch.Tensor):
        """x: the image batch [B, C, H, W] (fp32, as datasets.py:301-314 hands it to the reference's model) - or the bf16 patch
        matrix of that batch (is_patch_matrix), which the patch-embedding GEMM reads as it is."""
        if not x.is_cuda:
            raise VitkError("VisionTransformer.forward: input must be a GPU tensor; this model has no CPU implementation")
        if self.is_patch_matrix(x):
            if not x.is_contiguous():
                x = x.contiguous()
            batch = x.shape[0] // self.num_patches
        else:
            if x.dim() != 4 or x.shape[1] != self.in_chans or x.shape[2] != self.patch_embed.img_size[0] or x.shape[3] != self.patch_embed.img_size[1]:
                raise VitkError(f"input shape {tuple(x.shape)} does not match [B,{self.in_chans},{self.patch_embed.img_size[0]},{self.patch_embed.img_size[1]}]"
                                f" (or a bf16 patch matrix [B*{self.num_patches},{self.in_chans * self.patch_size ** 2}])")
            if x.dtype != f32 or not x.is_contiguous():
                x = x.float().contiguous()
            batch = x.shape[0]
        self._sync_weights()
        if self._static_drop is not None and self.training and self._static_drop["B"] == batch:
            scales = self._static_drop["scales"]         # refreshed by refresh_static_droppath(), never drawn in here
        else:
            scales = self._draw_drop_scales(batch, x.device)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._plist):
            return _ViTFunction.apply(x, self, scales, *self._plist)
        outs, _ = self._forward_impl(x, scales, save=False)
        return outs[0] if len(outs) == 1 else tuple(outs)

    def _forward_impl(self, img, scales, save):
        N, D, H, npre, P = self.num_tokens, self.embed_dim, self.num_heads, self.num_prefix_tokens, self.num_patches
        as_patches = img.dim() == 2          # the bf16 patch matrix itself (see forward)
        B = img.shape[0] // P if as_patches else img.shape[0]
        M = B * N
        wb = self._wb
        wa = self._wpad if self._pad_heads else wb      # qkv / proj weights as the attention branch reads them
        scale = self.head_dim ** -0.5
        patches = img if as_patches else ops.im2col_patches(img, self.patch_size)
        x = torch.empty((M, D), dtype=bf16, device=img.device)
        ops.gemm_nt(patches, wb["patch_embed.proj.weight"], epilogue=EPI_PATCH, bias=self.patch_embed.proj.bias, out=x,
                    pos=self._pos2d, patches=P, prefix=npre)
        ops.prefix_tokens_fwd(self._tokens, self._pos2d, x, B, N, D, npre)
        layers = []
        for i, blk in enumerate(self.blocks):
            s1, s2 = scales[i]
            pre = f"blocks.{i}."
            # ---- attention branch: x2 = x + s1 * proj(attn(qkv(ln1(x))))
            if s1 is not None and s1.compact and s1.nk == 0:
                y1 = m1 = r1 = qkv = o = lse = None
                x2 = x                                              # every sample drops the branch
            else:
                c1 = s1 is not None and s1.compact
                nb = s1.nk if c1 else B
                y1, m1, r1 = ops.layernorm_fwd(x, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps, smap=s1.kept if c1 else None,
                                               rows_per_sample=N)
                wq, bq, wp = wa[pre + "attn.qkv.weight"], (wa[pre + "attn.qkv.bias"] if self._pad_heads else blk.attn.qkv.bias), wa[pre + "attn.proj.weight"]
                qkv = ops.gemm_nt(y1, wq, epilogue=EPI_BIAS, bias=bq)
                o, lse = ops.attention_fwd(qkv, nb, N, H, scale)
                if c1:
                    x2 = torch.empty_like(x)
                    ops.gemm_nt(o, wp, epilogue=EPI_BIAS_RES, bias=blk.attn.proj.bias, res=x, rowscale=s1.scale_c,
                                rows_per_group=N, out=x2, smap=s1.kept, rows_per_sample=N)
                    ops.copy_samples(x, x2, s1.dropped, N * D)
                else:
                    x2 = ops.gemm_nt(o, wp, epilogue=EPI_BIAS_RES, bias=blk.attn.proj.bias, res=x,
                                     rowscale=None if s1 is None else s1.scale, rows_per_group=N)
            # ---- MLP branch: x3 = x2 + s2 * fc2(gelu(fc1(ln2(x2))))
            if s2 is not None and s2.compact and s2.nk == 0:
                y2 = m2 = r2 = h = a = None
                x3 = x2
            else:
                c2 = s2 is not None and s2.compact
                y2, m2, r2 = ops.layernorm_fwd(x2, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps, smap=s2.kept if c2 else None,
                                               rows_per_sample=N)
                # fc1's epilogue writes the activation a = gelu(h) and, for the backward pass, gelu'(h) in bf16 (named h below: it
                # takes the pre-activation's place; multiplying by a stored derivative costs the backward GEMM's epilogue a third
                # of what evaluating two exponentials per element there does)
                a = torch.empty((y2.shape[0], blk.mlp.fc1.out_features), dtype=bf16, device=img.device)
                h = ops.gemm_nt(y2, wb[pre + "mlp.fc1.weight"], epilogue=EPI_BIAS_GELU_DG, bias=blk.mlp.fc1.bias, out2=a)
                if c2:
                    x3 = torch.empty_like(x2)
                    ops.gemm_nt(a, wb[pre + "mlp.fc2.weight"], epilogue=EPI_BIAS_RES, bias=blk.mlp.fc2.bias, res=x2, rowscale=s2.scale_c,
                                rows_per_group=N, out=x3, smap=s2.kept, rows_per_sample=N)
                    ops.copy_samples(x2, x3, s2.dropped, N * D)
                else:
                    x3 = ops.gemm_nt(a, wb[pre + "mlp.fc2.weight"], epilogue=EPI_BIAS_RES, bias=blk.mlp.fc2.bias, res=x2,
                                     rowscale=None if s2 is None else s2.scale, rows_per_group=N)
            if save:
                layers.append((x, m1, r1, y1, qkv, o, lse, x2, m2, r2, y2, h, a, s1, s2))
            x = x3
        tail = {}

        def head_fwd(feat, wname, lin):
            """logits [B, num_classes] fp32; a class count that is not a multiple of 8 goes through the zero-padded copies"""
            if not self._pad_head:
                return ops.gemm_nt(feat, wb[wname], epilogue=EPI_BIAS_F32, bias=lin.bias)
            padded = ops.gemm_nt(feat, self._hpad[wname], epilogue=EPI_BIAS_F32, bias=self._hpad[wname[:-6] + "bias"])
            return padded[:, :self.num_classes].contiguous()

        if self.global_pool == "avg":
            pooled = ops.pool_avg_fwd(x, B, N, D, npre)
            f, mf, rf = ops.layernorm_fwd(pooled, self.fc_norm.weight, self.fc_norm.bias, self.fc_norm.eps)
            logits = head_fwd(f, "head.weight", self.head)
            outs = [logits]
            tail = dict(pooled=pooled, f=f, mf=mf, rf=rf)
        else:
            c = ops.gather_rows(x, B, N, D, 0)
            f, mf, rf = ops.layernorm_fwd(c, self.norm.weight, self.norm.bias, self.norm.eps)
            logits = head_fwd(f, "head.weight", self.head)
            outs = [logits]
            tail = dict(c=c, f=f, mf=mf, rf=rf)
            if self.distilled:
                c2 = ops.gather_rows(x, B, N, D, 1)
                f2, mf2, rf2 = ops.layernorm_fwd(c2, self.norm.weight, self.norm.bias, self.norm.eps)
                logits2 = head_fwd(f2, "head_dist.weight", self.head_dist)
                tail.update(c2=c2, f2=f2, mf2=mf2, rf2=rf2)
                if self.distilled_training and self.training:
                    outs = [logits, logits2]                                 # deit.py:114-116
                    tail["mode"] = "both"
                else:
                    outs = [(logits + logits2) / 2]                          # deit.py:119
                    tail["mode"] = "mean"
        saved = dict(B=B, patches=patches, layers=layers, tail=tail) if save else None
        return outs, saved

    # --------------------------------------------------------------------------------------------- backward
    def _wgrad_stream(self):
        """side stream of the weight-gradient GEMMs, None: everything on the current stream (VITK_WGRAD_STREAM=0 turns it off,
        =2 forces it on).  Beside a gradient reducer it is used only when the process has at least 8 hardware queues
        (GPU_MAX_HW_QUEUES, read by HIP when it starts): with HIP's default of 4 the launch stream, this stream, the reducer's
        stream and RCCL's own share queues, and a stream that shares the launch stream's queue serialises against it (measured
        with RCCL collectives in flight: 4 queues -0.7 %, 8 queues +1.2 % against no side stream).  The stream has the default
        priority - a high-priority stream that waits on events of a lower-priority one can starve that one of the queue slot
        it needs to record them."""
        mode = os.environ.get("VITK_WGRAD_STREAM", "1")
        if mode == "0":
            return None
        if self._grad_ready_hook is not None and mode != "2":
            try:
                queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
            except ValueError:
                queues = 4
            if queues < 8:
                return None
        if ops.prof_enabled():       # an event-bracketed launch is timed with the chip to itself
            return None
        dev = torch.cuda.current_device()
        if dev not in _WGRAD_SIDE:       # one per device and process: HIP streams beyond the few hardware queues share them
            _WGRAD_SIDE[dev] = (torch.cuda.Stream(), ops.Workspace())
        self._wgrad_side, self._ws_side = _WGRAD_SIDE[dev]
        return self._wgrad_side

    def _backward_impl(self, saved, douts):
        B = saved["B"]
        N, D, npre = self.num_tokens, self.embed_dim, self.num_prefix_tokens
        H = self.num_heads
        M = B * N
        wb, wt, g, ws = self._wb, self._wt, self._gview, self._ws
        scale = self.head_dim ** -0.5
        self._ensure_grad_buffer()
        g = self._gview
        if self._wt_ready is not None:           # transposed weights written on the side stream after the last optimizer step
            torch.cuda.current_stream().wait_event(self._wt_ready)
            self._wt_ready = None
        # grads: overwrite unless a previous backward since the last zero_grad already wrote the buffer
        acc = self._grad_valid and all(p.grad is not None for p in self._plist)
        tail = saved["tail"]

        def head_bwd(dlog, f, wname, bname, acc_):
            if not self._pad_head:
                dl = ops.cast_f32_to_bf16(dlog.contiguous())
                ops.gemm_tn(dl, f, g[wname], dbias=g[bname], accumulate=acc_, ws=ws)
                return ops.gemm_nt(dl, wt[wname])                           # [B, D]
            # padded class count: zero columns behind the real ones, products in the padded shape, real rows copied (or added) across
            C, Cp = self.num_classes, self._cls_pad
            dl = torch.zeros((dlog.shape[0], Cp), dtype=bf16, device=dlog.device)
            dl[:, :C].copy_(dlog)                                           # fp32 -> bf16, round to nearest even
            tw = torch.empty((Cp, f.shape[1]), dtype=f32, device=dlog.device)
            tb = torch.empty(Cp, dtype=f32, device=dlog.device)
            ops.gemm_tn(dl, f, tw, dbias=tb, accumulate=False, ws=ws)
            (g[wname].add_(tw[:C]), g[bname].add_(tb[:C])) if acc_ else (g[wname].copy_(tw[:C]), g[bname].copy_(tb[:C]))
            return ops.gemm_nt(dl, wt[wname])                               # wt: the transposed PADDED copy [D, Cp]

        if self.global_pool == "avg":
            df = head_bwd(douts[0], tail["f"], "head.weight", "head.bias", acc)
            dpooled = ops.layernorm_bwd(df, tail["pooled"], tail["mf"], tail["rf"], self.fc_norm.weight, g["fc_norm.weight"],
                                        g["fc_norm.bias"], accumulate=acc, ws=ws)
            dx = ops.pool_avg_bwd(dpooled, B, N, D, npre)
        else:
            if self.distilled:
                if tail["mode"] == "both":
                    d1 = douts[0] if douts[0] is not None else torch.zeros_like(douts[1])
                    d2 = douts[1] if douts[1] is not None else torch.zeros_like(douts[0])
                else:
                    d1 = d2 = douts[0] * 0.5
            else:
                d1 = douts[0]
            df = head_bwd(d1, tail["f"], "head.weight", "head.bias", acc)
            dc = ops.layernorm_bwd(df, tail["c"], tail["mf"], tail["rf"], self.norm.weight, g["norm.weight"], g["norm.bias"],
                                   accumulate=acc, ws=ws)
            dx = torch.empty((M, D), dtype=bf16, device=dc.device)
            ops.scatter_rows(dc, dx, B, N, D, 0, True)
            if self.distilled:
                df2 = head_bwd(d2, tail["f2"], "head_dist.weight", "head_dist.bias", acc)
                dc2 = ops.layernorm_bwd(df2, tail["c2"], tail["mf2"], tail["rf2"], self.norm.weight, g["norm.weight"],
                                        g["norm.bias"], accumulate=True, ws=ws)
                ops.scatter_rows(dc2, dx, B, N, D, 1, False)
        hook = self._grad_ready_hook
        if hook is not None:
            hook(self._head_end, False)

        layers = saved["layers"]

        def zero_branch(names):
            if not acc:
                for n in names:
                    g[n].zero_()

        def ln_bwd(dy, xin, mean, rstd, norm, gname, dres, took, nxt):
            """LayerNorm backward + residual gradient; `took` = the branch this LayerNorm feeds (its dy may be compact),
            `nxt` = the branch whose scaled input gradient this kernel also writes (None: no stochastic depth there)."""
            kw = {}
            if took is not None and took.compact:
                kw["in_inv"] = took.inv
            if nxt is not None:
                kw.update(rowscale=nxt.scale, rows_per_group=N)
                if nxt.compact:
                    if nxt.nk == 0:
                        kw.pop("rowscale")
                    else:
                        kw.update(out_inv=nxt.inv, out_rows=nxt.nk * N)
            if "in_inv" in kw and "rows_per_group" not in kw:
                kw["rows_per_group"] = N
            # nothing inside backward reads a LayerNorm's parameter gradients: with a side stream only the first stage (per-block
            # partial sums) runs here, the fixed-order reduction follows on the side stream with the next weight-gradient GEMM
            r = ops.layernorm_bwd(dy, xin, mean, rstd, norm.weight, g[gname + ".weight"], g[gname + ".bias"], dres=dres, accumulate=acc,
                                  ws=ws, partial_out=pending_ln if defer_ln else None, **kw)
            return r if isinstance(r, tuple) else (r, None)

        def branch_input_grad(dx_full, br, prepared):
            """the gradient that enters a branch: s * dx, compact when the branch is; `prepared` = the copy the LayerNorm backward
            above already wrote"""
            if br is None:
                return dx_full
            if prepared is not None:
                return prepared
            if br.compact:
                return ops.rowscale_gather(dx_full, br.scale, N, br.kept)
            return ops.rowscale(dx_full, br.scale, N)

        # Weight-gradient GEMMs have no consumer inside backward: on a side stream they fill the compute units that the tile
        # rounds of the data-gradient GEMMs leave idle.  Their operands stay referenced until this stream has waited for the
        # side stream's event of the block they belong to (one block later), so the allocator hands their memory on in stream
        # order.  (Tensor.record_stream instead would release a block only once the host SEES the side event complete: with the
        # host a step or more ahead of the GPU every step then allocates fresh memory.)
        side = self._wgrad_stream()
        held, lagging = [], None        # operands of the current block; (event, operands) of the block before
        pending_ln = []                 # LayerNorm parameter-gradient reductions not yet handed to the side stream
        defer_ln = side is not None and os.environ.get("VITK_DEFER_LN_REDUCE", "1") != "0"

        def flush_ln():
            """call with the side stream current and ordered behind the LayerNorm kernels that filled the partial buffers"""
            for item in pending_ln:
                ops.layernorm_bwd_reduce(item)
                held.append(item[0])
            pending_ln.clear()

        Hh, hd, hp = self.num_heads, self.head_dim, self._hd_pad

        def wgrad_gemm(dy, xin, wname, bname, ws_):
            """dW (+)= dy^T xin, dbias (+)= column sums of dy, into the flat gradient buffer.  With zero-padded heads the qkv / proj
            products come out in the padded shape: the real rows / columns are copied (or added) across afterwards."""
            if self._pad_heads and wname.endswith("attn.qkv.weight"):
                tw = torch.empty((3 * Hh * hp, xin.shape[1]), dtype=f32, device=dy.device)
                tb = torch.empty(3 * Hh * hp, dtype=f32, device=dy.device)
                ops.gemm_tn(dy, xin, tw, dbias=tb, accumulate=False, ws=ws_)
                gw, gb = g[wname].view(3 * Hh, hd, -1), g[bname].view(3 * Hh, hd)
                sw, sb = tw.view(3 * Hh, hp, -1)[:, :hd], tb.view(3 * Hh, hp)[:, :hd]
                (gw.add_(sw), gb.add_(sb)) if acc else (gw.copy_(sw), gb.copy_(sb))
            elif self._pad_heads and wname.endswith("attn.proj.weight"):
                tw = torch.empty((dy.shape[1], Hh * hp), dtype=f32, device=dy.device)
                tb = torch.empty(dy.shape[1], dtype=f32, device=dy.device)
                ops.gemm_tn(dy, xin, tw, dbias=tb, accumulate=False, ws=ws_)
                gw, sw = g[wname].view(-1, Hh, hd), tw.view(-1, Hh, hp)[:, :, :hd]
                (gw.add_(sw), g[bname].add_(tb)) if acc else (gw.copy_(sw), g[bname].copy_(tb))
            else:
                ops.gemm_tn(dy, xin, g[wname], dbias=g[bname], accumulate=acc, ws=ws_)

        def wgrad(dy, xin, wname, bname):
            if side is None:
                wgrad_gemm(dy, xin, wname, bname, ws)
                return
            ev = torch.cuda.Event()
            ev.record()
            side.wait_event(ev)
            with torch.cuda.stream(side):
                flush_ln()
                wgrad_gemm(dy, xin, wname, bname, self._ws_side)
            held.append((dy, xin))

        def side_event():
            if side is None:
                return None
            ev = torch.cuda.Event()
            ev.record(side)
            return ev

        def report(upto, final):
            if hook is not None:
                ev = side_event()
                if ev is None:
                    hook(upto, final)
                else:
                    hook(upto, final, after=ev)     # the reduction waits for the side stream; this stream does not

        dout_next = None     # s * dx for the branch about to be processed, when a LayerNorm backward already wrote it
        for i in reversed(range(self.depth)):
            blk = self.blocks[i]
            pre = f"blocks.{i}."
            (x, m1, r1, y1, qkv, o, lse, x2, m2, r2, y2, h, a, s1, s2) = layers[i]
            layers[i] = None
            dx_in = dx
            # ---- MLP branch: x3 = x2 + s2 * fc2(gelu(fc1(ln2(x2))))
            if s2 is not None and s2.compact and s2.nk == 0:
                zero_branch([pre + n for n in ("mlp.fc2.weight", "mlp.fc2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "norm2.weight", "norm2.bias")])
                dx2 = dx
                dout_attn = None
            else:
                dout = branch_input_grad(dx, s2, dout_next)
                wgrad(dout, a, pre + "mlp.fc2.weight", pre + "mlp.fc2.bias")
                dh = ops.gemm_nt(dout, wt[pre + "mlp.fc2.weight"], epilogue=EPI_MUL, aux=h)
                wgrad(dh, y2, pre + "mlp.fc1.weight", pre + "mlp.fc1.bias")
                dy2 = ops.gemm_nt(dh, wt[pre + "mlp.fc1.weight"])
                del dh
                # the LayerNorm backward also writes the stochastic-depth-scaled copy its consumer needs (s1 * dx2 here)
                dx2, dout_attn = ln_bwd(dy2, x2, m2, r2, blk.norm2, pre + "norm2", dx, s2, s1)
            dout_next = None
            # ---- attention branch: x2 = x + s1 * proj(attn(qkv(ln1(x))))
            s2_below = layers[i - 1][14] if i > 0 else None       # the branch processed next
            if s1 is not None and s1.compact and s1.nk == 0:
                zero_branch([pre + n for n in ("attn.proj.weight", "attn.proj.bias", "attn.qkv.weight", "attn.qkv.bias", "norm1.weight", "norm1.bias")])
                dx = dx2
            else:
                dout = branch_input_grad(dx2, s1, dout_attn)
                nb = s1.nk if (s1 is not None and s1.compact) else B
                wgrad(dout, o, pre + "attn.proj.weight", pre + "attn.proj.bias")
                do = ops.gemm_nt(dout, wt[pre + "attn.proj.weight"])
                dqkv = ops.attention_bwd(qkv, o, do, lse, nb, N, H, scale, ws=ws)
                wgrad(dqkv, y1, pre + "attn.qkv.weight", pre + "attn.qkv.bias")
                dy1 = ops.gemm_nt(dqkv, wt[pre + "attn.qkv.weight"])
                del dqkv
                dx, dout_next = ln_bwd(dy1, x, m1, r1, blk.norm1, pre + "norm1", dx2, s1, s2_below)
            if self._bwd_taps is not None:       # tests: the gradients entering / between / leaving the two half-blocks
                self._bwd_taps[i] = dict(dx_in=dx_in, dx2=dx2, dx=dx, x=x, x2=x2)
            report(self._layer_end[i], False)
            if side is not None:
                if lagging is not None:
                    torch.cuda.current_stream().wait_event(lagging[0])
                    lagging[1].clear()
                lagging, held = (side_event(), held), []

        # ---- embedding: x0[b, t] = (t < prefix ? token_t : patches @ Wp^T + bp) + pos[t]
        ops.pos_tokens_bwd(dx, self._gpos2d, self._gtokens, B, N, D, npre, accumulate=acc)
        dxp = ops.slice_tokens(dx, B, N, D, npre)
        ops.gemm_tn(dxp, saved["patches"], g["patch_embed.proj.weight"].view(D, -1), dbias=g["patch_embed.proj.bias"],
                    accumulate=acc, ws=ws)
        for n, p in zip(self._pnames, self._plist):
            if p.grad is None or p.grad.data_ptr() != self._gview[n].data_ptr():
                p.grad = self._gview[n]
        self._grad_valid = True
        if pending_ln:
            ev = torch.cuda.Event()
            ev.record()
            side.wait_event(ev)
            with torch.cuda.stream(side):
                flush_ln()
        report(self._total, True)
        ev = side_event()
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)      # the optimizer step reads every gradient


# ------------------------------------------------------------------------------------------------- registry
# name -> constructor defaults (models/vision_transformer.py:2690-2860, models/my_vit.py:84-165, models/deit.py:306-314)
_MODEL_CFGS = {
    "vit_tiny_patch16_224": dict(embed_dim=192, depth=12, num_heads=3, img_size=224),
    "vit_small_patch16_224": dict(embed_dim=384, depth=12, num_heads=6, img_size=224),
    "vit_base_patch16_224": dict(embed_dim=768, depth=12, num_heads=12, img_size=224),
    "vit_base_patch16_384": dict(embed_dim=768, depth=12, num_heads=12, img_size=384),
    "vit_large_patch16_224": dict(embed_dim=1024, depth=24, num_heads=16, img_size=224),
    "my_vit_mini": dict(embed_dim=144, depth=12, num_heads=3, img_size=224),
    "my_vit_ti": dict(embed_dim=192, depth=12, num_heads=3, img_size=224),
    "my_vit_xs": dict(embed_dim=288, depth=12, num_heads=4, img_size=224),
    "my_vit_s": dict(embed_dim=384, depth=12, num_heads=6, img_size=224),
    "my_vit_b": dict(embed_dim=768, depth=12, num_heads=12, img_size=224),
    "my_vit_l": dict(embed_dim=1024, depth=24, num_heads=16, img_size=224),
    "deit_tiny_distilled_patch16_224": dict(embed_dim=192, depth=12, num_heads=3, img_size=224, distilled=True),
    "deit_small_distilled_patch16_224": dict(embed_dim=384, depth=12, num_heads=6, img_size=224, distilled=True),
    "deit_base_distilled_patch16_224": dict(embed_dim=768, depth=12, num_heads=12, img_size=224, distilled=True),
}


def list_models():
    return sorted(_MODEL_CFGS)


def is_model(name: str) -> bool:
    return name in _MODEL_CFGS


def create_model(model_name: str, pretrained: bool = False, pretrained_cfg=None, pretrained_cfg_overlay=None,
                 checkpoint_path: Optional[str] = None, cache_dir=None, scriptable=None, exportable=None, no_jit=None,
                 **kwargs) -> nn.Module:
    """Same call contract as the reference's models.create_model (models/_factory.py:46-155): None-valued kwargs are
    dropped (:108), a '.tag' suffix on the name is ignored for construction (:130), an unknown name raises
    RuntimeError('Unknown model (name)') (:140).  pretrained=True is refused: there is no network (the reference only
    ever passes pretrained=False, main.py:645, test_kd.py:25)."""
    kwargs = {k: v for k, v in kwargs.items() if v is not None}
    name = model_name.split(".", 1)[0]
    if not is_model(name):
        raise RuntimeError("Unknown model (%s)" % name)
    if pretrained:
        raise RuntimeError("pretrained weights cannot be downloaded in this build (no network); pass pretrained=False "
                           "and load a local checkpoint with checkpoint_path= or load_state_dict")
    cfg = dict(_MODEL_CFGS[name])
    cfg.update(kwargs)
    model = VisionTransformer(patch_size=16, **cfg)
    if checkpoint_path:          # models/_factory.py:151-152 -> load_checkpoint (models/_helpers.py:89-140): strict, EMA weights first
        from .utils import load_checkpoint_state_dict
        model.load_state_dict(load_checkpoint_state_dict(checkpoint_path, use_ema=True))
    return model
