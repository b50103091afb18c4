"""Oracle (test infrastructure): image encoder, fp32 CPU, functional.

PARITY UNPINNED: the backbone arithmetic lives in ``timm`` (un-vendored,
un-pinned dependency; call site /root/reference/src/image_encoder.py:16,24),
which is absent from this image and from /root/reference; the reference holds
no test or golden vector at that boundary.  The topology below is the published
torchvision/timm ResNet-18 (BasicBlock [2,2,2,2], 7x7/2 stem, 3x3/2 max-pool,
global average pool, identity head for num_classes=0) and timm's
VisionTransformer (pre-norm, LN eps 1e-6, qkv bias, exact GELU, CLS pooling)
and timm's EfficientNet-B0 (MBConv stages, SE, SiLU), with timm's state_dict key names.  ImageEncoder's own arithmetic (metadata
concat, image_encoder.py:23-29) IS restated from the reference source.
"""
import math
import torch
import torch.nn.functional as F
from .rounding import r


def _bn2d(sd, name, x, train):
    y = F.batch_norm(x, sd[name + '.running_mean'], sd[name + '.running_var'],
                     sd[name + '.weight'], sd[name + '.bias'], training=train, momentum=0.1, eps=1e-5)
    if train and (name + '.num_batches_tracked') in sd:
        sd[name + '.num_batches_tracked'] += 1
    return y


def _basic_block_2d(sd, p, x, stride, train):
    out = r(F.conv2d(x, r(sd[p + 'conv1.weight']), None, stride, 1))
    out = r(F.relu(_bn2d(sd, p + 'bn1', out, train)))
    out = r(F.conv2d(out, r(sd[p + 'conv2.weight']), None, 1, 1))
    out = _bn2d(sd, p + 'bn2', out, train)
    if (p + 'downsample.0.weight') in sd:
        sc = r(F.conv2d(x, r(sd[p + 'downsample.0.weight']), None, stride, 0))
        sc = r(_bn2d(sd, p + 'downsample.1', sc, train))
    else:
        sc = x
    return r(F.relu(out + sc))


def resnet_features(sd, image, blocks=(2, 2, 2, 2), train=False, prefix='', round_stem_conv=False):
    """ResNet BasicBlock backbone, pooled features [B, 512].  (bf16 emulation: the HIP path's fused stem keeps the conv
    output in registers -- no rounding there -- and rounds the pooled activation only; `round_stem_conv` restores the
    rounding point of the unfused stem kernels, which store that map.)"""
    x = F.conv2d(r(image), r(sd[prefix + 'conv1.weight']), None, 2, 3)
    if round_stem_conv:
        x = r(x)
    x = r(F.relu(_bn2d(sd, prefix + 'bn1', x, train)))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, reps in enumerate(blocks, start=1):
        for bi in range(reps):
            stride = 2 if (li > 1 and bi == 0) else 1
            x = _basic_block_2d(sd, f'{prefix}layer{li}.{bi}.', x, stride, train)
    return x.mean(dim=(2, 3))


def vit_features(sd, image, num_heads, depth, patch=16, prefix=''):
    """timm VisionTransformer forward_features + CLS pooling -> [B, embed_dim]."""
    w = sd[prefix + 'patch_embed.proj.weight']
    x = F.conv2d(image, w, sd[prefix + 'patch_embed.proj.bias'], stride=patch)
    b, d = x.shape[0], x.shape[1]
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat((sd[prefix + 'cls_token'].expand(b, -1, -1), x), 1) + sd[prefix + 'pos_embed']
    hd = d // num_heads
    for i in range(depth):
        p = f'{prefix}blocks.{i}.'
        h = F.layer_norm(x, (d,), sd[p + 'norm1.weight'], sd[p + 'norm1.bias'], 1e-6)
        qkv = F.linear(h, sd[p + 'attn.qkv.weight'], sd[p + 'attn.qkv.bias'])
        t = x.shape[1]
        q, k, v = (z.reshape(b, t, num_heads, hd).transpose(1, 2) for z in qkv.chunk(3, dim=-1))
        a = ((q @ k.transpose(-1, -2)) / math.sqrt(hd)).softmax(dim=-1) @ v
        a = a.transpose(1, 2).reshape(b, t, d)
        x = x + F.linear(a, sd[p + 'attn.proj.weight'], sd[p + 'attn.proj.bias'])
        h = F.layer_norm(x, (d,), sd[p + 'norm2.weight'], sd[p + 'norm2.bias'], 1e-6)
        h = F.linear(F.gelu(F.linear(h, sd[p + 'mlp.fc1.weight'], sd[p + 'mlp.fc1.bias'])),
                     sd[p + 'mlp.fc2.weight'], sd[p + 'mlp.fc2.bias'])
        x = x + h
    x = F.layer_norm(x, (d,), sd[prefix + 'norm.weight'], sd[prefix + 'norm.bias'], 1e-6)
    return x[:, 0]


EFFICIENTNET_B0 = ((1, 3, 1, 1, 16), (2, 3, 2, 6, 24), (2, 5, 2, 6, 40), (3, 3, 2, 6, 80), (3, 5, 1, 6, 112),
                   (4, 5, 2, 6, 192), (1, 3, 1, 6, 320))        # (repeats, kernel, stride, expansion, out channels)


def _se(sd, p, x):
    """timm SqueezeExcite: x * sigmoid(conv_expand(silu(conv_reduce(mean_hw x))))."""
    s = x.mean((2, 3), keepdim=True)
    s = F.silu(F.conv2d(s, sd[p + 'conv_reduce.weight'], sd[p + 'conv_reduce.bias']))
    s = F.conv2d(s, sd[p + 'conv_expand.weight'], sd[p + 'conv_expand.bias'])
    return x * torch.sigmoid(s)


def efficientnet_features(sd, image, arch=EFFICIENTNET_B0, train=False, prefix=''):
    """timm EfficientNet (efficientnet_b0 family: symmetric padding, BN eps 1e-5, SiLU, SE on 1/4 of the block input
    width, no drop-path), pooled features [B, 1280].  Topology from the published definition; parity unpinned."""
    x = r(F.conv2d(r(image), sd[prefix + 'conv_stem.weight'], None, 2, 1))
    x = r(F.silu(r(_bn2d(sd, prefix + 'bn1', x, train))))
    for si, (reps, k, stride, exp, cout) in enumerate(arch):
        for bi in range(reps):
            p = f'{prefix}blocks.{si}.{bi}.'
            s = stride if bi == 0 else 1
            inp = x
            if exp != 1:
                x = r(F.conv2d(x, r(sd[p + 'conv_pw.weight'])))
                x = r(F.silu(r(_bn2d(sd, p + 'bn1', x, train))))
                dw_bn = 'bn2'
            else:
                dw_bn = 'bn1'
            w = sd[p + 'conv_dw.weight']
            x = r(F.conv2d(x, w, None, s, k // 2, groups=w.shape[0]))
            x = r(F.silu(r(_bn2d(sd, p + dw_bn, x, train))))
            x = r(_se(sd, p + 'se.', x))
            last, last_bn = ('conv_pw', 'bn2') if exp == 1 else ('conv_pwl', 'bn3')
            x = r(F.conv2d(x, r(sd[p + last + '.weight'])))
            x = _bn2d(sd, p + last_bn, x, train)
            x = r(x + inp) if (s == 1 and inp.shape[1] == x.shape[1]) else r(x)
    x = r(F.conv2d(x, r(sd[prefix + 'conv_head.weight'])))
    x = r(F.silu(r(_bn2d(sd, prefix + 'bn2', x, train))))
    return x.mean(dim=(2, 3))


def image_encoder_forward(sd, image, image_shape, arch='resnet18', train=False, metadata=True,
                          prefix='backbone.', **arch_kw):
    """ImageEncoder.forward, image_encoder.py:23-29: features ++ (orig H, W) / tensor H."""
    if arch.startswith('resnet'):
        x = resnet_features(sd, image, arch_kw.get('blocks', (2, 2, 2, 2)), train, prefix)
    elif arch.startswith('efficientnet'):
        x = efficientnet_features(sd, image, arch_kw.get('arch', EFFICIENTNET_B0), train, prefix)
    else:
        x = vit_features(sd, image, arch_kw['num_heads'], arch_kw['depth'], arch_kw.get('patch', 16), prefix)
    if metadata:
        x = torch.cat((x, image_shape.to(image.dtype) / image.shape[2]), 1)
    return x
