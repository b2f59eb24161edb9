"""Segmenter (ViT encoder + mask-transformer decoder): the transformer model family attacked by SEA.

Own implementation for PyTorch-ROCm with the reference's state-dict schema (185
tensors for ViT-S/16 with 151 classes; semseg/models/segmenter.py:196-353,
backbones/vit_encoder.py:84-294, heads/segmenter_decoder.py:30-99).  Attention goes through libsea_hip M7 (fp32 MFMA
flash attention, deterministic backward) for fp32 HIP tensors with head dimension 64 and through
``F.scaled_dot_product_attention`` otherwise, instead of materialising softmax(QK^T); LayerNorms with frozen parameters
go through M5.
Forward semantics: pad to a multiple of 16, encode, drop the class token, decode to (B,n_cls,H/16,W/16)
masks, bilinear x16 back to the padded size, crop.
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import convnext_upernet as _cu
from .convnext_upernet import (StochasticDepth, _fp32_bwd, _fp32_fwd, _layer_norm, _linear_bound_word, _linear_frozen,
                               _ln_bound_word, _up)


def _init(m):
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=0.02)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.LayerNorm):
        nn.init.zeros_(m.bias)
        nn.init.ones_(m.weight)


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm (same parameters / state-dict keys) whose forward goes through libsea_hip M5 when the affine
    parameters are frozen and the input is a contiguous fp32 HIP tensor (the attack's case); ATen otherwise."""

    def forward(self, x):
        if len(self.normalized_shape) == 1 and self.weight is not None and self.bias is not None:
            return _layer_norm(x, self.normalized_shape[0], self.weight, self.bias, self.eps)
        return super().forward(x)


USE_HIP_ATTENTION = True


class _AttentionHip(torch.autograd.Function):
    """softmax(q k^T * scale) v on a packed (B,T,3,H,64) fp32 qkv tensor through libsea_hip M7 (fp32 MFMA flash
    attention, deterministic backward); returns (B,T,H*64), the layout the output projection consumes."""

    @staticmethod
    @_fp32_fwd
    def forward(ctx, qkv, scale, train=False):
        from .. import _native as N
        # training: three bf16 terms per operand (the fp32 operands exactly), forward AND backward -- the backward consumes the
        # forward's out / lse; attack (frozen qkv projection): N.attn_terms_fwd(), fp16 x 2 by default
        out, lse = N.attention_qkv(qkv, scale, terms=3 if train else None)
        ctx.save_for_backward(qkv, out, lse)
        ctx.scale, ctx.train = scale, train
        return out

    @staticmethod
    @_fp32_bwd
    def backward(ctx, g):
        from .. import _native as N
        qkv, out, lse = ctx.saved_tensors
        # training: three bf16 terms per operand (the fp32 operands exactly); attack: N.attn_terms_bwd() -- fp16 x 2 by default
        # (22 significant bits per operand like every frozen-weight GEMM of the evaluation, at half the matrix work of three
        # bf16 terms); SEA_ATTN_TERMS_BWD = 3 / 2 / 0 select the other modes
        return N.attention_qkv_backward(qkv, out, lse, g, ctx.scale, terms=3 if ctx.train else None), None, None


class Attention(nn.Module):
    def __init__(self, dim, heads, dropout):
        super().__init__()
        self.heads = heads
        self.scale = (dim // heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3)
        self.attn_drop = nn.Dropout(dropout)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(dropout)

    def forward(self, x, amax=None):
        B, T, D = x.shape
        # frozen weights (the attack's forward): M8 split GEMMs, forward and input gradient; plain F.linear otherwise.
        # ``amax`` (fp16 x 2 mode): bound word of max|x| from the caller; the attention output is a convex combination of
        # the v rows, so the analytic bound of |qkv| bounds the projection's input
        gc = self.__dict__.setdefault("_gemm_cache", ({}, {}))
        a2 = None
        if amax is not None and not (self.qkv.weight.requires_grad or self.qkv.bias.requires_grad):
            a2 = _linear_bound_word(amax, self.qkv.weight, self.qkv.bias, gc[0])
        qkv = _linear_frozen(gc[0], x, self.qkv.weight, self.qkv.bias, amax=amax)
        p = self.attn_drop.p if self.training else 0.0
        if (USE_HIP_ATTENTION and p == 0.0 and qkv.is_cuda and qkv.dtype == torch.float32 and D // self.heads == 64
                and qkv.is_contiguous()):
            y = _AttentionHip.apply(qkv.view(B, T, 3, self.heads, 64), self.scale, self.qkv.weight.requires_grad)
            return self.proj_drop(_linear_frozen(gc[1], y, self.proj.weight, self.proj.bias, amax=a2))
        q, k, v = qkv.reshape(B, T, 3, self.heads, D // self.heads).permute(2, 0, 3, 1, 4)
        y = F.scaled_dot_product_attention(q, k, v, dropout_p=p, scale=self.scale)
        return self.proj_drop(self.proj(y.transpose(1, 2).reshape(B, T, D)))


class FeedForward(nn.Module):
    def __init__(self, dim, hidden, dropout):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)
        self.drop = nn.Dropout(dropout)

    def forward(self, x, amax=None, res=None):
        """``res`` (optional, shape of x): returns res + mlp(x) (the block's residual rides in the second GEMM's split-K
        reduce pass on the fused path)"""
        gc = self.__dict__.setdefault("_gemm_cache", ({}, {}))
        a2 = None
        if amax is not None and not (self.fc1.weight.requires_grad or self.fc1.bias.requires_grad):
            a2 = _linear_bound_word(amax, self.fc1.weight, self.fc1.bias, gc[0])   # |GELU(t)| <= |t| <= bound of the first GEMM
        if ((not self.training or self.drop.p == 0.0) and self.act.approximate == "none"
                and _cu._mlp_fusable(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)):
            r = res if (res is not None and res.is_contiguous()) else None
            y = _cu._FrozenMlp.apply(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, r, gc, amax, a2)
            return y if (res is None or r is not None) else res + y
        h = self.drop(self.act(_linear_frozen(gc[0], x, self.fc1.weight, self.fc1.bias, amax=amax)))
        out = self.drop(_linear_frozen(gc[1], h, self.fc2.weight, self.fc2.bias, amax=a2))
        return out if res is None else res + out


class Block(nn.Module):
    def __init__(self, dim, heads, mlp_dim, dropout, drop_path):
        super().__init__()
        self.norm1 = LayerNorm(dim)
        self.norm2 = LayerNorm(dim)
        self.attn = Attention(dim, heads, dropout)
        self.mlp = FeedForward(dim, mlp_dim, dropout)
        self.drop_path = StochasticDepth(drop_path) if drop_path > 0.0 else nn.Identity()

    def _bound(self, norm, x, slot):
        """fp16 x 2 GEMMs after a frozen LayerNorm take the analytic bound of its output as activation scale (eval mode:
        dropout would rescale the activations between the GEMMs)"""
        from .. import _native as N
        if (self.training or not x.is_cuda or _cu._terms() != 22 or not N.AMAX_FROM_PRODUCERS or norm.weight.requires_grad or norm.bias.requires_grad
                or x.dtype != torch.float32):
            return None
        return _ln_bound_word(norm, self.__dict__.setdefault("_ln_cache", ({}, {}))[slot])

    def forward(self, x):
        x = x + self.drop_path(self.attn(self.norm1(x), self._bound(self.norm1, x, 0)))
        if isinstance(self.drop_path, nn.Identity) or not self.training:
            return self.mlp(self.norm2(x), self._bound(self.norm2, x, 1), res=x)
        return x + self.drop_path(self.mlp(self.norm2(x), self._bound(self.norm2, x, 1)))


class PatchEmbedding(nn.Module):
    def __init__(self, image_size, patch_size, embed_dim, channels):
        super().__init__()
        if image_size[0] % patch_size or image_size[1] % patch_size:
            raise ValueError("image dimensions must be divisible by the patch size")
        self.image_size = image_size
        self.grid_size = (image_size[0] // patch_size, image_size[1] // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.patch_size = patch_size
        self.proj = nn.Conv2d(channels, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, im):
        """The stride-P PxP convolution (vit_encoder.py:135-150) as ONE GEMM on the non-overlapping patches: same
        arithmetic, but a hipBLASLt GEMM instead of a MIOpen convolution: bitwise reproducible, and no MIOpen kernel is
        JIT-compiled per (batch, image size) on a fresh box (140 s each, measured in the GPU test run)."""
        B, C, H, W = im.shape
        P = self.patch_size
        if im.is_cuda and H % P == 0 and W % P == 0 and self.proj.groups == 1:
            patches = im.reshape(B, C, H // P, P, W // P, P).permute(0, 2, 4, 1, 3, 5).reshape(B, (H // P) * (W // P), C * P * P)
            return F.linear(patches, self.proj.weight.reshape(self.proj.weight.shape[0], -1), self.proj.bias)
        return self.proj(im).flatten(2).transpose(1, 2)


def resize_pos_embed(posemb, grid_old, grid_new, num_extra_tokens):
    tok, grid = posemb[:, :num_extra_tokens], posemb[0, num_extra_tokens:]
    if grid_old is None:
        gh = gw = int(math.sqrt(len(grid)))
    else:
        gh, gw = grid_old
    grid = grid.reshape(1, gh, gw, -1).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=grid_new, mode="bilinear")
    grid = grid.permute(0, 2, 3, 1).reshape(1, grid_new[0] * grid_new[1], -1)
    return torch.cat([tok, grid], dim=1)


class VisionTransformer(nn.Module):
    def __init__(self, image_size, patch_size, n_layers, d_model, d_ff, n_heads, n_cls, dropout=0.1,
                 drop_path_rate=0.1, distilled=False, channels=3):
        super().__init__()
        self.patch_embed = PatchEmbedding(image_size, patch_size, d_model, channels)
        self.patch_size, self.n_layers, self.d_model, self.d_ff, self.n_heads = patch_size, n_layers, d_model, d_ff, n_heads
        self.n_cls, self.distilled = n_cls, distilled
        self.dropout = nn.Dropout(dropout)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, d_model))
        extra = 2 if distilled else 1
        if distilled:
            self.dist_token = nn.Parameter(torch.zeros(1, 1, d_model))
        self.pos_embed = nn.Parameter(torch.randn(1, self.patch_embed.num_patches + extra, d_model))
        if distilled:
            self.head_dist = nn.Linear(d_model, n_cls)
        rates = torch.linspace(0, drop_path_rate, n_layers).tolist()
        self.blocks = nn.ModuleList(Block(d_model, n_heads, d_ff, dropout, rates[i]) for i in range(n_layers))
        self.norm = LayerNorm(d_model)
        self.head = nn.Linear(d_model, n_cls)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        if distilled:
            nn.init.trunc_normal_(self.dist_token, std=0.02)
        self.apply(_init)

    def no_weight_decay(self):
        return {"pos_embed", "cls_token", "dist_token"}

    def forward(self, im, pre_neck=False):
        B, _, H, W = im.shape
        x = self.patch_embed(im)
        toks = [self.cls_token.expand(B, -1, -1)]
        if self.distilled:
            toks.append(self.dist_token.expand(B, -1, -1))
        x = torch.cat(toks + [x], dim=1)
        pos = self.pos_embed
        if x.shape[1] != pos.shape[1]:
            pos = resize_pos_embed(pos, self.patch_embed.grid_size, (H // self.patch_size, W // self.patch_size),
                                   1 + int(self.distilled))
        x = self.dropout(x + pos)
        for blk in self.blocks:
            x = blk(x)
        x = self.norm(x)
        if pre_neck:
            return x
        if self.distilled:
            return (self.head(x[:, 0]) + self.head_dist(x[:, 1])) / 2
        return self.head(x[:, 0])


class MaskTransformer(nn.Module):
    def __init__(self, n_cls, patch_size, d_encoder, n_layers, n_heads, d_model, d_ff, drop_path_rate, dropout):
        super().__init__()
        self.d_encoder, self.patch_size, self.n_layers, self.n_cls = d_encoder, patch_size, n_layers, n_cls
        self.d_model, self.d_ff = d_model, d_ff
        self.scale = d_model ** -0.5
        rates = torch.linspace(0, drop_path_rate, n_layers).tolist()
        self.blocks = nn.ModuleList(Block(d_model, n_heads, d_ff, dropout, rates[i]) for i in range(n_layers))
        self.cls_emb = nn.Parameter(torch.randn(1, n_cls, d_model))
        self.proj_dec = nn.Linear(d_encoder, d_model)
        self.proj_patch = nn.Parameter(self.scale * torch.randn(d_model, d_model))
        self.proj_classes = nn.Parameter(self.scale * torch.randn(d_model, d_model))
        self.decoder_norm = LayerNorm(d_model)
        self.mask_norm = LayerNorm(n_cls)
        self.apply(_init)
        nn.init.trunc_normal_(self.cls_emb, std=0.02)

    def no_weight_decay(self):
        return {"cls_emb"}

    def forward(self, x, im_size):
        gs = im_size[0] // self.patch_size
        x = torch.cat((self.proj_dec(x), self.cls_emb.expand(x.size(0), -1, -1)), 1)
        for blk in self.blocks:
            x = blk(x)
        x = self.decoder_norm(x)
        patches, cls_feat = x[:, :-self.n_cls] @ self.proj_patch, x[:, -self.n_cls:] @ self.proj_classes
        patches = patches / patches.norm(dim=-1, keepdim=True)
        cls_feat = cls_feat / cls_feat.norm(dim=-1, keepdim=True)
        masks = self.mask_norm(patches @ cls_feat.transpose(1, 2))           # (B, h*w, n_cls)
        return masks.transpose(1, 2).reshape(x.size(0), self.n_cls, gs, -1)   # b (h w) n -> b n h w


class DecoderLinear(nn.Module):
    def __init__(self, n_cls, patch_size, d_encoder):
        super().__init__()
        self.d_encoder, self.patch_size, self.n_cls = d_encoder, patch_size, n_cls
        self.head = nn.Linear(d_encoder, n_cls)
        self.apply(_init)

    def no_weight_decay(self):
        return set()

    def forward(self, x, im_size):
        gs = im_size[0] // self.patch_size
        return self.head(x).transpose(1, 2).reshape(x.size(0), self.n_cls, gs, -1)


class SegMenter(nn.Module):
    def __init__(self, encoder, decoder, n_cls, backbone):
        super().__init__()
        self.n_cls, self.patch_size, self.encoder, self.decoder, self.backbone = n_cls, 16, encoder, decoder, backbone

    def no_weight_decay(self):
        return {"encoder." + k for k in self.encoder.no_weight_decay()} | {"decoder." + k for k in self.decoder.no_weight_decay()}

    def forward_lowres(self, im):
        """(masks at 1/16 resolution, output size) when no padding is involved, else None (see
        UperNetForSemanticSegmentation.forward_lowres)."""
        H0, W0 = im.shape[2:]
        if H0 % self.patch_size or W0 % self.patch_size:
            return None
        x = self.encoder(im, pre_neck=True)
        x = x[:, 0 if "SAM" in self.backbone else 1 + int(self.encoder.distilled):]
        return self.decoder(x, (H0, W0)).contiguous(), (H0, W0)

    def forward(self, im):
        H0, W0 = im.shape[2:]
        ph, pw = (-H0) % self.patch_size, (-W0) % self.patch_size
        if ph or pw:
            im = F.pad(im, (0, pw, 0, ph), value=0)
        H, W = im.shape[2:]
        x = self.encoder(im, pre_neck=True)
        x = x[:, 0 if "SAM" in self.backbone else 1 + int(self.encoder.distilled):]
        masks = _up(self.decoder(x, (H, W)).contiguous(), (H, W))  # bilinear x16 (libsea_hip M2 on HIP tensors)
        return masks[:, :, :H0, :W0] if (ph or pw) else masks


def create_vit(model_cfg, pretrained=None):
    cfg = dict(model_cfg)
    cfg.pop("backbone", None)
    cfg.pop("normalization", None)
    cfg["n_cls"] = 1000
    cfg["d_ff"] = 4 * cfg["d_model"]
    model = VisionTransformer(**cfg)
    if pretrained:
        ckpt = torch.load(pretrained, map_location="cpu")
        ckpt = {k.replace("model.", "").replace("module.", ""): v for k, v in ckpt.items()}
        for k in ("base_normalize.mean", "base_normalize.std"):
            ckpt.pop(k, None)
        ckpt = {k.replace("base_", ""): v for k, v in ckpt.items()}
        own = model.state_dict()
        if "pos_embed" in ckpt and ckpt["pos_embed"].shape != own["pos_embed"].shape:
            n_extra = 1 + int(model.distilled)
            ckpt["pos_embed"] = resize_pos_embed(ckpt["pos_embed"], None, model.patch_embed.grid_size, n_extra)
        model.load_state_dict({k: v for k, v in ckpt.items() if k in own and v.shape == own[k].shape}, strict=False)
    return model


def create_decoder(encoder, decoder_cfg, backbone):
    cfg = dict(decoder_cfg)
    name = cfg.pop("name")
    cfg["d_encoder"] = 768 if "SAM" in backbone else 384
    cfg["patch_size"] = 16
    if "linear" in name:
        return DecoderLinear(**cfg)
    if name == "mask_transformer":
        dim = cfg["d_encoder"]
        cfg.update(n_heads=dim // 64, d_model=dim, d_ff=4 * dim)
        return MaskTransformer(**cfg)
    raise ValueError(f"Unknown decoder: {name}")


def create_segmenter(model_cfg, pretrained=None, backbone="vit_small_patch16_224"):
    """``create_segmenter(model_cfg, pretrained, backbone)`` (segmenter.py:344-353).  ``pretrained=None``
    skips the ImageNet checkpoint import (the reference always calls torch.load)."""
    cfg = dict(model_cfg)
    dec = dict(cfg.pop("decoder"))
    dec["n_cls"] = cfg["n_cls"]
    encoder = create_vit(cfg, pretrained)
    decoder = create_decoder(encoder, dec, backbone=backbone)
    return SegMenter(encoder, decoder, n_cls=cfg["n_cls"], backbone=backbone)
