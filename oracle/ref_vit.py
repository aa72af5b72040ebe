"""CPU restatement of the reference ViT (test oracle; see oracle/__init__.py).

Structure follows models/vision_transformer.py: Block :109-178, VisionTransformer.__init__ :452-623 (use_fc_norm rule
:529, dpr=linspace :581), init_weights :634-648 + init_weights_vit_timm :998-1010, _pos_embed :743-780,
forward_features/pool/forward_head :934-990, global_pool_nlc :419-441; the distilled variant follows
models/deit.py:28-119; model sizes follow :2690-2860 and models/my_vit.py:84-165.
Parameter names are timm's state-dict keys, so weights move freely between this oracle and the HIP model.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ref_ops as R

# name -> dict(embed_dim, depth, num_heads, img_size, distilled)
MODEL_CFGS = {
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
    "deit_base_distilled_patch16_224": dict(embed_dim=768, depth=12, num_heads=12, img_size=224, distilled=True),
    "deit_tiny_distilled_patch16_224": dict(embed_dim=192, depth=12, num_heads=3, img_size=224, distilled=True),
}


class _Attn(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _Block(nn.Module):
    def __init__(self, dim, heads, mlp_ratio, drop_path):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attn(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.drop_path = drop_path


class _PatchEmbed(nn.Module):
    def __init__(self, in_chans, dim, patch):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, dim, kernel_size=patch, stride=patch)


class RefViT(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, global_pool="token", embed_dim=768,
                 depth=12, num_heads=12, mlp_ratio=4.0, drop_path_rate=0.0, distilled=False):
        super().__init__()
        assert global_pool in ("avg", "token")
        if distilled:
            assert global_pool == "token"                       # deit.py:38
        self.global_pool, self.distilled = global_pool, distilled
        self.embed_dim, self.depth, self.num_heads, self.num_classes = embed_dim, depth, num_heads, num_classes
        self.patch_size, self.img_size = patch_size, img_size
        self.num_prefix_tokens = 2 if distilled else 1
        self.num_patches = (img_size // patch_size) ** 2
        use_fc_norm = global_pool == "avg"                      # vision_transformer.py:529
        self.patch_embed = _PatchEmbed(in_chans, embed_dim, patch_size)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        if distilled:
            self.dist_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.num_patches + self.num_prefix_tokens, embed_dim))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]   # :581
        self.blocks = nn.ModuleList([_Block(embed_dim, num_heads, mlp_ratio, dpr[i]) for i in range(depth)])
        if use_fc_norm:
            self.fc_norm = nn.LayerNorm(embed_dim, eps=1e-6)    # :616
        else:
            self.norm = nn.LayerNorm(embed_dim, eps=1e-6)       # :603
        self.head = nn.Linear(embed_dim, num_classes)
        if distilled:
            self.head_dist = nn.Linear(embed_dim, num_classes)
            self.distilled_training = False
        self.init_weights()

    def init_weights(self):
        # :634-648, :998-1010, deit.py:49-51; the Conv2d keeps torch's default init
        nn.init.trunc_normal_(self.pos_embed, std=0.02, a=-2.0, b=2.0)
        nn.init.normal_(self.cls_token, std=1e-6)
        if self.distilled:
            nn.init.trunc_normal_(self.dist_token, std=0.02, a=-2.0, b=2.0)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02, a=-2.0, b=2.0)
                nn.init.zeros_(m.bias)

    def no_weight_decay(self):
        return {"pos_embed", "cls_token", "dist_token"}        # :668

    def set_distilled_training(self, enable=True):
        self.distilled_training = enable

    def make_drop_scales(self, B, generator=None):
        """Per-layer per-sample DropPath factors mask/keep (timm drop_path, scale_by_keep=True), [depth][2] of [B] or None."""
        out = []
        for blk in self.blocks:
            pair = []
            for _ in range(2):
                if blk.drop_path > 0.0 and self.training:
                    keep = 1.0 - blk.drop_path
                    mask = torch.empty(B).bernoulli_(keep, generator=generator)
                    pair.append(mask / keep)
                else:
                    pair.append(None)
            out.append(pair)
        return out

    def attn_half(self, blk, x, s1, E, tap=lambda k, v: None):
        """x + drop_path1(attn(norm1(x)))   (models/vision_transformer.py:176); x [B,N,D]"""
        w = lambda t: R.rnd(t, E, fwd=True, bwd=False)
        act = lambda t: R.rnd(t, E, fwd=True, bwd=True)
        y = act(R.layernorm(x, blk.norm1.weight, blk.norm1.bias))
        tap('y1', y)
        qkv = act(y @ w(blk.attn.qkv.weight).t() + blk.attn.qkv.bias)
        tap('qkv', qkv)
        o = R.rnd(R.attention(qkv, self.num_heads, emulate=E), E, fwd=False, bwd=True)
        tap('o', o)
        br = R.rnd(o @ w(blk.attn.proj.weight).t() + blk.attn.proj.bias, E, fwd=False, bwd=True)
        if s1 is not None:
            br = br * s1.to(br.dtype).reshape(-1, 1, 1)
        out = act(x + br)
        tap('x2', out)
        return out

    def mlp_half(self, blk, x, s2, E, tap=lambda k, v: None):
        """x + drop_path2(mlp(norm2(x)))   (models/vision_transformer.py:177)"""
        w = lambda t: R.rnd(t, E, fwd=True, bwd=False)
        act = lambda t: R.rnd(t, E, fwd=True, bwd=True)
        y = act(R.layernorm(x, blk.norm2.weight, blk.norm2.bias))
        tap('y2', y)
        # the pre-activation is never stored (the HIP path keeps gelu'(h) and gelu(h)): no forward rounding; its gradient is
        f1 = R.rnd(y @ w(blk.mlp.fc1.weight).t() + blk.mlp.fc1.bias, E, fwd=False, bwd=True)
        tap('h', f1)
        a = R.rnd(R.gelu(f1, emulate=E), E, fwd=True, bwd=False)
        tap('a', a)
        br = R.rnd(a @ w(blk.mlp.fc2.weight).t() + blk.mlp.fc2.bias, E, fwd=False, bwd=True)
        if s2 is not None:
            br = br * s2.to(br.dtype).reshape(-1, 1, 1)
        out = act(x + br)
        tap('x3', out)
        return out

    def forward(self, x, emulate=False, drop_scales=None, taps=None):
        """x [B,C,H,W] fp32.  drop_scales: output of make_drop_scales (None: drawn here when training with drop path).
        emulate=True rounds to bf16 wherever the HIP kernels store bf16 (see ref_ops)."""
        E = emulate
        B = x.shape[0]
        D, H = self.embed_dim, self.num_heads
        if drop_scales is None:
            drop_scales = self.make_drop_scales(B)
        w = lambda t: R.rnd(t, E, fwd=True, bwd=False)          # bf16 weight copy, fp32 gradient
        act = lambda t: R.rnd(t, E, fwd=True, bwd=True)         # bf16 activation, bf16 gradient
        gonly = lambda t: R.rnd(t, E, fwd=False, bwd=True)      # fp32 value, bf16 gradient

        patches = R.im2col_patches(x, self.patch_size)
        if E:
            patches = R.rb(patches)
        wp = w(self.patch_embed.proj.weight.reshape(D, -1))
        pos = self.pos_embed[0]
        npre = self.num_prefix_tokens
        tok = patches @ wp.t() + self.patch_embed.proj.bias + pos[npre:]
        pre = [self.cls_token[0, 0] + pos[0]]
        if self.distilled:
            pre.append(self.dist_token[0, 0] + pos[1])
        pre = torch.stack(pre, 0).unsqueeze(0).expand(B, -1, -1)
        h = act(torch.cat([pre, tok], dim=1))                   # [B,N,D]
        tap = (lambda k, v: taps.__setitem__(k, v.detach())) if taps is not None else (lambda k, v: None)
        tap('x0', h)

        for li, (blk, (s1, s2)) in enumerate(zip(self.blocks, drop_scales)):
            h = self.attn_half(blk, h, s1, E, lambda k, v: tap(f'{li}.{k}', v))
            h = self.mlp_half(blk, h, s2, E, lambda k, v: tap(f'{li}.{k}', v))

        if self.global_pool == "avg":
            feat = act(h[:, npre:].mean(dim=1))                 # :431-433, prefix tokens excluded
            feat = act(R.layernorm(feat, self.fc_norm.weight, self.fc_norm.bias))
            return gonly(feat @ w(self.head.weight).t() + self.head.bias)
        # 'token': final norm on every token, then take the prefix token(s); LayerNorm is per row, so only those rows matter
        cls = act(R.layernorm(h[:, 0], self.norm.weight, self.norm.bias))
        logits = gonly(cls @ w(self.head.weight).t() + self.head.bias)
        if not self.distilled:
            return logits
        dist = act(R.layernorm(h[:, 1], self.norm.weight, self.norm.bias))
        logits_dist = gonly(dist @ w(self.head_dist.weight).t() + self.head_dist.bias)
        if self.distilled_training and self.training:
            return logits, logits_dist                          # deit.py:114-116
        return (logits + logits_dist) / 2                       # deit.py:119


def create_ref_model(name, num_classes=1000, global_pool=None, drop_path_rate=0.0, depth=None, **kw):
    cfg = dict(MODEL_CFGS[name])
    distilled = cfg.pop("distilled", False)
    if depth is not None:
        cfg["depth"] = depth
    if global_pool is None:
        global_pool = "token"
    cfg.update(kw)
    return RefViT(num_classes=num_classes, global_pool=global_pool, drop_path_rate=drop_path_rate, distilled=distilled, **cfg)
