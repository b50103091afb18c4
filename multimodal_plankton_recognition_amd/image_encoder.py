"""Image encoder -- drop-in counterpart of /root/reference/src/image_encoder.py:8-29.

The reference delegates the backbone to ``timm.create_model(name, num_classes, pretrained=True,
in_chans)`` (an un-vendored dependency that FETCHES weights).  Here the backbone is built in-repo,
random-initialised with timm's ResNet scheme (kaiming-normal fan_out convs, BN weight 1 / bias 0,
zero-initialised last BN of every residual branch) and keeps timm's ``state_dict`` key names
(``conv1``, ``bn1``, ``layer{1..4}.{i}.{conv1,bn1,conv2,bn2,downsample.{0,1}}``), so a timm
checkpoint loads unchanged.  ``pretrained`` is accepted and ignored (there is no network).
Activations are channels-last bf16 inside; the module boundary is the reference's
(``image`` fp32 [B, in_chans, H, W] in, fp32 [B, num_features + 2] out).
"""
from typing import Dict

import torch
from torch import Tensor, nn

from .layers import BasicBlock, BatchNormParams, PoolTailFn, StemFn, chain_blocks
from .ops import ConvGeom, to_krsc_

_RESNETS = {'resnet10t': None, 'resnet18': (2, 2, 2, 2), 'resnet34': (3, 4, 6, 3), 'resnet14': (1, 2, 2, 1),
            'resnet10': (1, 1, 1, 1)}


class ResNetBackbone(nn.Module):
    """BasicBlock ResNet (timm/torchvision topology): 7x7/2 conv - BN - ReLU - maxpool 3x3/2 - 4 stages."""

    def __init__(self, blocks=(2, 2, 2, 2), in_chans: int = 1, zero_init_last: bool = True):
        super().__init__()
        self.conv1 = nn.Conv2d(in_chans, 64, 7, 2, 3, bias=False)
        self.bn1 = BatchNormParams(64)
        self.geom = ConvGeom(tuple(self.conv1.weight.shape), 2, 3)
        cin = 64
        for li, (reps, ch) in enumerate(zip(blocks, (64, 128, 256, 512)), start=1):
            stride = 1 if li == 1 else 2
            seq = [BasicBlock(2, cin, ch, stride, downsample=(stride != 1 or cin != ch))]
            seq += [BasicBlock(2, ch, ch, 1, downsample=False) for _ in range(1, reps)]
            setattr(self, f'layer{li}', nn.Sequential(*seq))
            cin = ch
        self.num_features = 512
        self.in_chans = in_chans
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        if zero_init_last:
            for m in self.modules():
                if isinstance(m, BasicBlock):
                    nn.init.zeros_(m.bn2.weight)
        to_krsc_(self)          # block filters: [K][R][S][C] memory (state_dict / logical shapes unchanged)
        self._chain = chain_blocks(blk for li in range(1, 5) for blk in getattr(self, f'layer{li}'))

    def forward_features(self, image: Tensor) -> Tensor:
        """fp32 [B, in_chans, H, W] -> channels-last bf16 (fp32 under `precision: 32`) [B, H/32, W/32, 512]."""
        B, C, H, W = image.shape
        x = image.float()
        x = x.reshape(B, H, W, 1) if C == 1 else x.permute(0, 2, 3, 1)
        x = x.contiguous()
        self._chain.clear()
        from . import layers_f32
        if layers_f32.conv_f32():            # `precision: 32`: fp32 maps on the exact-fp32 kernels (parity mode)
            out = layers_f32.stem(self, x)
        else:
            out = StemFn.apply(x, self.conv1.weight, self.bn1.weight, self.bn1.bias, self)
        for li in range(1, 5):
            for blk in getattr(self, f'layer{li}'):
                out = blk(out)
        return out


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _Attn(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)


class _ViTBlock(nn.Module):
    """Parameter container with timm's Block key names (norm1, attn.qkv, attn.proj, norm2, mlp.fc1, mlp.fc2)."""

    def __init__(self, dim, mlp_ratio=4):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attn(dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, dim * mlp_ratio)


class _PatchEmbed(nn.Module):
    def __init__(self, in_chans, dim, patch):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, dim, patch, patch)


class ViTBackbone(nn.Module):
    """timm VisionTransformer (pre-norm blocks, LN eps 1e-6, qkv bias, exact GELU, CLS pooling, learned position
    embedding); timm ``state_dict`` key names.  Arithmetic: ``transformer.py`` (fp32 kernels)."""

    def __init__(self, embed_dim=768, depth=12, num_heads=12, patch=16, img_size=224, in_chans: int = 1):
        super().__init__()
        self.patch, self.num_heads, self.num_features = patch, num_heads, embed_dim
        n_tok = (img_size // patch) ** 2 + 1
        self.patch_embed = _PatchEmbed(in_chans, embed_dim, patch)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.randn(1, n_tok, embed_dim) * .02)
        self.blocks = nn.Sequential(*[_ViTBlock(embed_dim) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        nn.init.normal_(self.cls_token, std=1e-6)
        self.__dict__['_pe_view'] = None
        for m in self.modules():            # timm: trunc_normal(.02) weights, zero biases, LN (1, 0)
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                nn.init.zeros_(m.bias)

    def __deepcopy__(self, memo):
        # the cached [d, C*P*P] view of the patch-embedding filter is a non-leaf tensor (not deep-copyable): the copy
        # rebuilds its own on first use
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k == '_pe_view' else copy.deepcopy(v, memo)
        return new

    def forward_tokens(self, image: Tensor) -> Tensor:
        from . import transformer as TF
        from .layers import linear
        B, C, H, W = image.shape
        P = self.patch
        d = self.num_features
        gh, gw = H // P, W // P
        # conv-as-GEMM patch embedding: [B, gh*gw, C*P*P] x [d, C*P*P]^T
        patches = image.float().reshape(B, C, gh, P, gw, P).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gw, C * P * P)
        if TF.mixed() and (C * P * P) % 8 == 0 and d % 8 == 0:
            from .transformer_mixed import LinearMixedFn
            # ONE persistent [d, C*P*P] view of the conv filter: the bf16 panel cache and the optimizer's repack table
            # hang off the tensor object, a fresh view per step would rebuild both every step
            w4 = self.patch_embed.proj.weight
            w2 = self.__dict__.get('_pe_view')
            want_grad = w4.requires_grad and torch.is_grad_enabled()
            if (w2 is None or w2.data_ptr() != w4.data_ptr() or w2.device != w4.device
                    or (want_grad and w2.grad_fn is None)):
                # (a view made under no_grad -- an eval / export pass before training -- has no grad_fn: used in a later
                # training step it would silently drop the patch embedding's weight gradient; rebuilt here with one)
                w2 = w4.view(d, -1)
                object.__setattr__(self, '_pe_view', w2)     # (not a module attribute: deepcopy / state_dict never see it)
            x = LinearMixedFn.apply(patches.contiguous(), w2, self.patch_embed.proj.bias)
        else:
            x = linear(patches.contiguous(), self.patch_embed.proj.weight.view(d, -1), self.patch_embed.proj.bias)
        x = torch.cat((self.cls_token.expand(B, 1, d), x.view(B, gh * gw, d)), 1).contiguous()
        T = x.shape[1]
        index = torch.arange(T, device=x.device).repeat(B)
        x = TF.EmbeddingAddFn.apply(x.view(B * T, d), self.pos_embed.view(-1, d), index, None).view(B, T, d)
        for blk in self.blocks:
            x = TF.pre_norm_block(blk, x, self.num_heads, 0.0, self.training)
        x = TF.add_layer_norm(x.reshape(B * T, d), None, self.norm.weight, self.norm.bias, self.norm.eps)
        return x.view(B, T, d)

    def forward_pooled(self, image: Tensor) -> Tensor:
        return self.forward_tokens(image)[:, 0].contiguous()          # global_pool='token'


_VITS = {'vit_tiny_patch16_224': (192, 12, 3, 16), 'vit_small_patch16_224': (384, 12, 6, 16),
         'vit_base_patch16_224': (768, 12, 12, 16), 'vit_small_patch32_224': (384, 12, 6, 32)}


def create_backbone(name: str, in_chans: int = 1):
    if name in _RESNETS and _RESNETS[name] is not None:
        return ResNetBackbone(_RESNETS[name], in_chans)
    if name in _VITS:
        dim, depth, heads, patch = _VITS[name]
        return ViTBackbone(dim, depth, heads, patch, 224, in_chans)
    if name == 'efficientnet_b0':
        from .efficientnet import EfficientNetBackbone
        return EfficientNetBackbone(in_chans)
    raise NotImplementedError(
        f"image backbone '{name}': BasicBlock ResNets ({', '.join(k for k, v in _RESNETS.items() if v)}) and ViTs "
        f"({', '.join(_VITS)}) and efficientnet_b0 have native gfx950 kernels so far")


class ImageEncoder(nn.Module):
    """Reference: src/image_encoder.py:8-29 (same arguments, same forward keyword contract)."""

    def __init__(self, name: str, num_classes: int = 0, pretrained: bool = False, dropout: float = 0.1,
                 in_chans: int = 1, metadata: bool = True) -> None:
        super().__init__()
        if num_classes != 0:
            raise NotImplementedError('ImageEncoder: num_classes must be 0 (feature extractor), as in every card')
        self.backbone = create_backbone(name, in_chans)
        self.dim_out = self.backbone.num_features + 2 * metadata
        self.metadata = metadata
        self.p_drop = float(dropout)

    def forward(self, image: Tensor, **kwargs) -> Tensor:
        meta = kwargs['image_shape'].contiguous() if self.metadata else None     # (orig H, W) / tensor H  (:26-27)
        p = self.p_drop if self.training else 0.0
        if isinstance(self.backbone, ViTBackbone):
            from .layers import TailFn
            return TailFn.apply(self.backbone.forward_pooled(image), meta, image.shape[2], p)
        fmap = self.backbone.forward_features(image)
        return PoolTailFn.apply(fmap, meta, 'avg', image.shape[2], p)
