"""Writes tests/golden/timm_topology.json: state_dict key -> shape tables of the two BASELINE image backbones as timm
publishes them (`timm.create_model(name, num_classes=0, in_chans=1)`, the call at /root/reference/src/image_encoder.py:16),
written out from the PUBLISHED architecture definitions -- not from this repo's modules:

  * resnet18  (He et al. 2015, torchvision / timm `resnet18`): 7x7/2 stem conv 64, BN, 3x3/2 max pool, four stages of two
    BasicBlocks (64, 128, 256, 512; 3x3 convs; 1x1/2 conv + BN shortcut at the entry of stages 2-4), global average pool,
    fc.  Published size with 3 input channels and the 1000-class fc: 11,689,512 parameters.
  * vit_base_patch16_224 (Dosovitskiy et al. 2020, timm): 16x16 patch embedding conv (bias), class token, 197 learned
    positions, 12 pre-norm blocks of width 768 (fused qkv with bias, proj, MLP 3072), final LayerNorm, head.  Published
    size with 3 input channels and the 1000-class head: 86,567,656 parameters.

timm itself is not installed here (and must not be fetched): the tables are the pin the oracle and the in-repo backbones
are checked against; the script also re-derives the two published totals from the tables as a self-check.
    python tests/golden/make_timm_topology.py
"""
import json
import os


def bn(prefix, c):
    return {prefix + '.weight': [c], prefix + '.bias': [c], prefix + '.running_mean': [c], prefix + '.running_var': [c],
            prefix + '.num_batches_tracked': []}


def resnet18(in_chans, num_classes):
    t = {'conv1.weight': [64, in_chans, 7, 7]}
    t.update(bn('bn1', 64))
    cin = 64
    for li, ch in enumerate((64, 128, 256, 512), start=1):
        for bi in range(2):
            p = f'layer{li}.{bi}.'
            t[p + 'conv1.weight'] = [ch, cin if bi == 0 else ch, 3, 3]
            t.update(bn(p + 'bn1', ch))
            t[p + 'conv2.weight'] = [ch, ch, 3, 3]
            t.update(bn(p + 'bn2', ch))
            if bi == 0 and li > 1:
                t[p + 'downsample.0.weight'] = [ch, cin, 1, 1]
                t.update(bn(p + 'downsample.1', ch))
        cin = ch
    if num_classes:
        t['fc.weight'], t['fc.bias'] = [num_classes, 512], [num_classes]
    return t


def vit_base(in_chans, num_classes, dim=768, depth=12, mlp=3072, patch=16, tokens=197):
    t = {'cls_token': [1, 1, dim], 'pos_embed': [1, tokens, dim],
         'patch_embed.proj.weight': [dim, in_chans, patch, patch], 'patch_embed.proj.bias': [dim]}
    for i in range(depth):
        p = f'blocks.{i}.'
        t[p + 'norm1.weight'], t[p + 'norm1.bias'] = [dim], [dim]
        t[p + 'attn.qkv.weight'], t[p + 'attn.qkv.bias'] = [3 * dim, dim], [3 * dim]
        t[p + 'attn.proj.weight'], t[p + 'attn.proj.bias'] = [dim, dim], [dim]
        t[p + 'norm2.weight'], t[p + 'norm2.bias'] = [dim], [dim]
        t[p + 'mlp.fc1.weight'], t[p + 'mlp.fc1.bias'] = [mlp, dim], [mlp]
        t[p + 'mlp.fc2.weight'], t[p + 'mlp.fc2.bias'] = [dim, mlp], [dim]
    t['norm.weight'], t['norm.bias'] = [dim], [dim]
    if num_classes:
        t['head.weight'], t['head.bias'] = [num_classes, dim], [num_classes]
    return t


def n_params(table):
    n = 0
    for k, shape in table.items():
        if k.endswith(('running_mean', 'running_var', 'num_batches_tracked')):
            continue
        m = 1
        for s in shape:
            m *= s
        n += m
    return n


if __name__ == '__main__':
    assert n_params(resnet18(3, 1000)) == 11_689_512          # torchvision / timm resnet18, as published
    assert n_params(vit_base(3, 1000)) == 86_567_656          # timm vit_base_patch16_224, as published
    out = {'resnet18': {'call': "timm.create_model('resnet18', num_classes=0, in_chans=1)", 'keys': resnet18(1, 0),
                        'num_params': n_params(resnet18(1, 0)), 'published_3ch_1000cls_params': 11_689_512,
                        'num_features': 512},
           'vit_base_patch16_224': {'call': "timm.create_model('vit_base_patch16_224', num_classes=0, in_chans=1)",
                                    'keys': vit_base(1, 0), 'num_params': n_params(vit_base(1, 0)),
                                    'published_3ch_1000cls_params': 86_567_656, 'num_features': 768}}
    assert out['resnet18']['num_params'] == 11_170_240
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'timm_topology.json')
    json.dump(out, open(path, 'w'), indent=0, sort_keys=True)
    print({k: (v['num_params'], len(v['keys'])) for k, v in out.items()})
