"""EfficientNet-B0 backbone: topology against timm's published parameter count / key names (CPU), the new
kernels against torch on identical bf16-rounded operands, a reduced-depth network against the oracle (GPU)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

DEV = 'cuda'
BF = torch.bfloat16


def rel(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-20))


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def test_efficientnet_b0_topology_matches_timm_published_numbers():
    from multimodal_plankton_recognition_amd.efficientnet import EfficientNetBackbone
    from multimodal_plankton_recognition_amd.image_encoder import ImageEncoder
    m = EfficientNetBackbone(in_chans=3)
    assert sum(p.numel() for p in m.parameters()) == 4_007_548          # timm efficientnet_b0, num_classes=0
    sd = m.state_dict()
    for k, shape in {'conv_stem.weight': (32, 3, 3, 3), 'blocks.0.0.conv_dw.weight': (32, 1, 3, 3),
                     'blocks.0.0.se.conv_reduce.weight': (8, 32, 1, 1), 'blocks.0.0.conv_pw.weight': (16, 32, 1, 1),
                     'blocks.1.0.conv_pw.weight': (96, 16, 1, 1), 'blocks.1.0.se.conv_reduce.bias': (4,),
                     'blocks.2.0.conv_dw.weight': (144, 1, 5, 5), 'blocks.5.3.conv_pwl.weight': (192, 1152, 1, 1),
                     'blocks.6.0.bn3.running_var': (320,), 'conv_head.weight': (1280, 320, 1, 1), 'bn2.bias': (1280,)}.items():
        assert tuple(sd[k].shape) == shape, k
    enc = ImageEncoder('efficientnet_b0')                                  # the reference cards' default backbone
    assert enc.dim_out == 1282 and enc.backbone.conv_stem.weight.shape[1] == 1


def test_oracle_efficientnet_equals_a_torch_nn_restatement():
    """The functional oracle against an independent nn.Module composition of one MBConv stage (same weights)."""
    from oracle.image_encoder import efficientnet_features
    from multimodal_plankton_recognition_amd.efficientnet import EfficientNetBackbone
    arch = ((1, 3, 1, 1, 16), (2, 5, 2, 6, 24))
    torch.manual_seed(0)
    m = EfficientNetBackbone(in_chans=1, arch=arch, num_features=64)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = rnd(2, 1, 32, 32, seed=1)
    out = efficientnet_features(sd, x, arch, train=False)
    # hand-rolled: stem
    bn = lambda t, p: F.batch_norm(t, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                                   False, 0.1, 1e-5)
    h = F.silu(bn(F.conv2d(x, sd['conv_stem.weight'], None, 2, 1), 'bn1'))
    p = 'blocks.0.0.'
    h = F.silu(bn(F.conv2d(h, sd[p + 'conv_dw.weight'], None, 1, 1, groups=32), p + 'bn1'))
    s = F.adaptive_avg_pool2d(h, 1)
    s = torch.sigmoid(F.conv2d(F.silu(F.conv2d(s, sd[p + 'se.conv_reduce.weight'], sd[p + 'se.conv_reduce.bias'])),
                               sd[p + 'se.conv_expand.weight'], sd[p + 'se.conv_expand.bias']))
    h = bn(F.conv2d(h * s, sd[p + 'conv_pw.weight']), p + 'bn2')
    for i, stride in ((0, 2), (1, 1)):
        p = f'blocks.1.{i}.'
        inp = h
        t = F.silu(bn(F.conv2d(h, sd[p + 'conv_pw.weight']), p + 'bn1'))
        t = F.silu(bn(F.conv2d(t, sd[p + 'conv_dw.weight'], None, stride, 2, groups=t.shape[1]), p + 'bn2'))
        s = F.adaptive_avg_pool2d(t, 1)
        s = torch.sigmoid(F.conv2d(F.silu(F.conv2d(s, sd[p + 'se.conv_reduce.weight'], sd[p + 'se.conv_reduce.bias'])),
                                   sd[p + 'se.conv_expand.weight'], sd[p + 'se.conv_expand.bias']))
        t = bn(F.conv2d(t * s, sd[p + 'conv_pwl.weight']), p + 'bn3')
        h = t + inp if stride == 1 else t
    h = F.silu(bn(F.conv2d(h, sd['conv_head.weight']), 'bn2')).mean((2, 3))
    np.testing.assert_allclose(out.numpy(), h.numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('B,H,W,C,k,stride', [(2, 17, 13, 16, 3, 1), (3, 16, 16, 24, 5, 2), (1, 9, 20, 96, 3, 2),
                                              (2, 7, 7, 40, 5, 1),
                                              # the register-tiled kernels at EfficientNet-B0's widths: partial strips (7, 14 and
                                              # odd widths), channel-group counts that are not powers of two (18, 84, 144)
                                              (2, 14, 14, 480, 3, 1), (3, 7, 7, 1152, 5, 1), (2, 28, 28, 144, 5, 2),
                                              (2, 15, 17, 672, 5, 2), (2, 56, 56, 32, 3, 1), (5, 14, 14, 672, 5, 1),
                                              (2, 29, 31, 240, 3, 2)])
def test_depthwise_conv_kernels(B, H, W, C, k, stride):
    from multimodal_plankton_recognition_amd import efficientnet as E
    from multimodal_plankton_recognition_amd.ops import ConvGeom
    x = rnd(B, C, H, W, seed=1).to(BF).float()
    w = rnd(C, 1, k, k, seed=2, scale=0.3)
    g = ConvGeom((C, 1, k, k), stride, k // 2)
    xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.conv2d(xg, wg, None, stride, k // 2, groups=C)
    dy = rnd(*ref.shape, seed=3).to(BF).float()
    ref.backward(dy)
    xd = x.permute(0, 2, 3, 1).contiguous().to(BF).to(DEV)
    wd = w.to(DEV)
    y = E.dwconv_fwd(xd, wd, g)
    assert rel(y.permute(0, 3, 1, 2), ref) < 4e-3
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(BF).to(DEV)
    dx = E.dwconv_dgrad(dyd, wd, g, xd.shape)
    assert rel(dx.permute(0, 3, 1, 2), xg.grad) < 4e-3
    dw = E.dwconv_wgrad(xd, dyd, g, wd)
    np.testing.assert_allclose(dw.cpu().numpy(), wg.grad.numpy(), rtol=2e-4, atol=2e-4 * float(wg.grad.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize('B,C,rd', [(3, 48, 12), (70, 1152, 48), (5, 24, 6), (3, 48, 72)])
def test_squeeze_excite_and_silu_functions(B, C, rd):
    """rd <= 64: the bottleneck in one launch forward / two backward (csrc/se_mlp.hip); wider: the GEMM path."""
    from multimodal_plankton_recognition_amd import efficientnet as E
    H, W = 6, 5
    x = rnd(B, C, H, W, seed=1).to(BF).float()
    ps = [rnd(rd, C, 1, 1, seed=2, scale=0.2), rnd(rd, seed=3, scale=0.2), rnd(C, rd, 1, 1, seed=4, scale=0.2), rnd(C, seed=5, scale=0.2)]
    xr = x.clone().requires_grad_(True)
    pr = [p.clone().requires_grad_(True) for p in ps]
    s = xr.mean((2, 3), keepdim=True)
    s = torch.sigmoid(F.conv2d(F.silu(F.conv2d(s, pr[0], pr[1])), pr[2], pr[3]))
    ref = F.silu(xr * s)
    dy = rnd(*ref.shape, seed=6).to(BF).float()
    ref.backward(dy)
    xd = x.permute(0, 2, 3, 1).contiguous().to(BF).to(DEV).requires_grad_(True)
    pd = [p.to(DEV).requires_grad_(True) for p in ps]
    out = E.SiLUFn.apply(E.SqueezeExciteFn.apply(xd, *pd))
    assert rel(out.permute(0, 3, 1, 2), ref) < 8e-3
    out.backward(dy.permute(0, 2, 3, 1).contiguous().to(BF).to(DEV))
    assert rel(xd.grad.permute(0, 3, 1, 2), xr.grad) < 2e-2
    for a, b in zip(pd, pr):
        assert rel(a.grad, b.grad) < 2e-2


@pytest.mark.gpu
def test_reduced_efficientnet_against_oracle():
    from multimodal_plankton_recognition_amd.efficientnet import EfficientNetBackbone
    from oracle.image_encoder import efficientnet_features
    arch = ((1, 3, 1, 1, 16), (2, 3, 2, 6, 24), (2, 5, 2, 6, 40))
    torch.manual_seed(0)
    m = EfficientNetBackbone(in_chans=1, arch=arch, num_features=128)
    with torch.no_grad():
        for k, v in m.named_buffers():
            if k.endswith('running_mean'):
                v.copy_(rnd(*v.shape, seed=len(k), scale=0.1))
            elif k.endswith('running_var'):
                v.copy_(1 + 0.3 * torch.rand(v.shape, generator=torch.Generator().manual_seed(len(k))))
        for k, v in m.named_parameters():
            if v.dim() == 1:
                v.add_(rnd(*v.shape, seed=len(k), scale=0.1))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    image = rnd(8, 1, 64, 64, seed=1, scale=0.5)
    wsum = rnd(8, 128, seed=2)
    m.to(DEV)
    # eval: running statistics
    m.eval()
    ref = efficientnet_features({k: v.clone() for k, v in sd.items()}, image, arch, train=False)
    got = m.forward_features(image.to(DEV)).float().mean((1, 2))
    assert rel(got, ref) < 2e-2, rel(got, ref)
    # train: batch statistics, gradients, running-buffer update
    m.train()
    params = {k: v.clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd.items()}
    ref = efficientnet_features(params, image, arch, train=True)
    (ref * wsum).sum().backward()
    out = m.forward_features(image.to(DEV)).float().mean((1, 2))
    assert rel(out, ref) < 5e-2, rel(out, ref)
    (out * wsum.to(DEV)).sum().backward()
    cos, bad = [], []
    for k, v in m.named_parameters():
        a, b = v.grad.detach().float().cpu().reshape(-1), params[k].grad.reshape(-1)
        if float(b.norm()) < 1e-4:
            # a shift in front of a linear map + train-mode BatchNorm (the bias of a projection BN without activation)
            # has a mathematically zero gradient: the oracle shows fp32 noise there, the bf16 path bf16 noise
            assert float(a.norm()) < 5e-2, (k, float(a.norm()))
            continue
        c = float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-30))
        cos.append(c)
        if c < 0.9:
            bad.append((k, round(c, 3), float(a.norm()), float(b.norm())))
    assert np.median(cos) > 0.95 and not bad, (np.median(cos), bad)
    new = m.state_dict()
    assert rel(new['blocks.1.0.bn2.running_var'], params['blocks.1.0.bn2.running_var']) < 2e-2
    assert int(new['bn1.num_batches_tracked']) == 1


@pytest.mark.gpu
def test_efficientnet_b0_in_multimodel_trains():
    from multimodal_plankton_recognition_amd.model import MultiModel
    torch.manual_seed(0)
    model = MultiModel(dim_embed=64, image_encoder_args=dict(name='efficientnet_b0', dropout=0.1),
                       profile_encoder_args=dict(dim_in=6, blocks=[1, 1, 1, 1], base_channels=8, dropout=0.1),
                       coordination_args=dict(method='clip'), optim_args=dict(lr=1e-2, momentum=0.9, nesterov=True)).to(DEV).train()
    opt = model.configure_optimizers()
    g = torch.Generator().manual_seed(1)
    batch = dict(image=torch.randn(8, 1, 96, 96, generator=g).to(DEV), profile=(torch.rand(8, 96, 6, generator=g) * 2 - 1).to(DEV),
                 image_shape=torch.randint(32, 400, (8, 2), generator=g).to(DEV),
                 profile_len=torch.randint(8, 1024, (8, 1), generator=g).to(DEV), buckets=1)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = model.training_step(batch, 0)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[0] != losses[-1], losses
    w = model.image_encoder.backbone.blocks[3][0].conv_dw.weight
    assert w.grad is not None and float(w.grad.abs().sum()) > 0
