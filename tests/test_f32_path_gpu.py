"""fp32 PARITY mode of the conv stacks (`precision: 32`, csrc/conv_f32.hip + layers_f32.py) against
(a) torch fp32 on the CPU, op by op, (b) the golden fixtures generated from the reference's own ProfileCNN
(src/profile_encoder.py:111-240) and the composed two-step fixture, (c) the CPU oracle for whole ResNet-18 / MultiModel.

Tolerances: forward 1e-4, parameter gradients 1e-3 (relative L2), whole ResNet-18 train-mode gradients 1e-2;
two runs of one build are bit-identical (no atomics anywhere on this path)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'
T = torch.from_numpy


def rel_l2(got, ref):
    got = got.detach().float().cpu()
    ref = torch.as_tensor(ref).detach().float()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-12))


@pytest.fixture
def f32_mode():
    from multimodal_plankton_recognition_amd import layers_f32
    old = layers_f32.set_conv_precision('32')
    yield
    layers_f32._PRECISION[0] = old


def _load(mod, g, prefix='sd.'):
    mod.load_state_dict({k[len(prefix):]: T(v.copy()) for k, v in g.items() if k.startswith(prefix)}, strict=True)


# geometry: (dims, B, H, W, C, K, R, stride, pad, krsc)
CONV_CASES = [
    (2, 3, 12, 10, 16, 24, 3, 1, 1, True),       # body 3x3 / 1
    (2, 2, 13, 11, 64, 128, 3, 2, 1, True),      # downsampling 3x3 / 2 on odd sizes
    (2, 2, 14, 14, 64, 128, 1, 2, 0, True),      # projection shortcut 1x1 / 2
    (2, 2, 32, 32, 1, 64, 7, 2, 3, False),       # ResNet stem, one channel, OIHW memory
    (1, 4, 1, 50, 6, 16, 3, 2, 1, False),        # ProfileCNN conv1 (Conv1d k3 / 2 on 6 channels)
    (1, 3, 1, 37, 32, 32, 3, 1, 1, True),        # 1-D body conv
    (2, 1, 5, 5, 8, 8, 3, 1, 1, False),          # tiny: one partial tile
    (2, 5, 20, 20, 72, 200, 3, 1, 1, True),      # sizes that are no multiple of the 64 x 64 tile or the chunk of 32
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_f32_matches_torch(case):
    from multimodal_plankton_recognition_amd.layers_f32 import ConvF32Fn
    from multimodal_plankton_recognition_amd.ops import ConvGeom, grad_target
    dims, B, H, W, C, K, R, stride, pad, krsc = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    if dims == 2:
        x = torch.randn(B, C, H, W, generator=g)
        w = torch.randn(K, C, R, R, generator=g) / (C * R * R) ** 0.5
    else:
        x = torch.randn(B, C, W, generator=g)
        w = torch.randn(K, C, R, generator=g) / (C * R) ** 0.5
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    conv = F.conv2d if dims == 2 else F.conv1d
    y_ref = conv(xr, wr, stride=stride, padding=pad)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)

    wd = w.to(DEV)
    if krsc:     # channels-last filter memory with the logical OIHW shape (what ops.to_krsc_ gives the block filters)
        wd = wd.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2) if dims == 2 else wd.permute(0, 2, 1).contiguous().permute(0, 2, 1)
    wd.requires_grad_(True)
    xd = (x.permute(0, 2, 3, 1) if dims == 2 else x.permute(0, 2, 1)).contiguous().to(DEV).requires_grad_(True)
    geom = ConvGeom(tuple(w.shape), stride, pad)
    y = ConvF32Fn.apply(xd, wd, geom)
    gyd = (gy.permute(0, 2, 3, 1) if dims == 2 else gy.permute(0, 2, 1)).contiguous().to(DEV)
    y.backward(gyd)
    back = (lambda t: t.permute(0, 3, 1, 2)) if dims == 2 else (lambda t: t.permute(0, 2, 1))
    assert rel_l2(back(y), y_ref) < 2e-6
    assert rel_l2(back(xd.grad), xr.grad) < 2e-6
    assert rel_l2(wd.grad, wr.grad) < 5e-6
    assert grad_target(wd) is None


@pytest.mark.parametrize('train', [True, False])
@pytest.mark.parametrize('shape', [(6, 9, 7, 24), (4, 33, 64), (2, 3, 3, 130)])
def test_bn_act_f32_matches_torch(train, shape):
    from multimodal_plankton_recognition_amd.layers import BatchNormParams
    from multimodal_plankton_recognition_amd.layers_f32 import BNActF32Fn
    C = shape[-1]
    g = torch.Generator().manual_seed(C)
    x = torch.randn(shape, generator=g) * 1.7 + 0.4
    res = torch.randn(shape, generator=g)
    gy = torch.randn(shape, generator=g)
    ref = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        ref.weight.copy_(torch.rand(C, generator=g) + 0.5)
        ref.bias.copy_(torch.randn(C, generator=g) * 0.2)
        ref.running_mean.copy_(torch.randn(C, generator=g) * 0.1)
        ref.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    bn = BatchNormParams(C)
    bn.load_state_dict(ref.state_dict())
    bn.to(DEV)
    ref.train(train)
    bn.train(train)
    xr, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    y_ref = torch.relu(ref(xr.reshape(-1, C)).reshape(shape) + rr)
    y_ref.backward(gy)
    xd, rd = x.to(DEV).requires_grad_(True), res.to(DEV).requires_grad_(True)
    y = BNActF32Fn.apply(xd, bn.weight, bn.bias, rd, True, bn)
    y.backward(gy.to(DEV))
    assert rel_l2(y, y_ref) < 2e-6
    assert rel_l2(xd.grad, xr.grad) < 2e-5
    assert torch.equal(rd.grad.cpu() != 0, rr.grad != 0) and rel_l2(rd.grad, rr.grad) < 1e-6
    assert rel_l2(bn.weight.grad, ref.weight.grad) < 2e-5 and rel_l2(bn.bias.grad, ref.bias.grad) < 2e-5
    sd = bn.state_dict()
    assert rel_l2(sd['running_mean'], ref.running_mean) < 1e-6 and rel_l2(sd['running_var'], ref.running_var) < 1e-6
    assert int(sd['num_batches_tracked']) == int(ref.num_batches_tracked)


@pytest.mark.parametrize('shape', [(3, 17, 13, 8), (2, 16, 16, 64), (4, 31, 16)])
def test_pools_f32_match_torch(shape):
    from multimodal_plankton_recognition_amd import layers_f32 as L
    g = torch.Generator().manual_seed(5)
    x = torch.randn(shape, generator=g)
    x = (x * 4).round() / 4          # ties, so that the first-maximum rule matters
    dims = len(shape) - 2
    to_cf = (lambda t: t.permute(0, 3, 1, 2)) if dims == 2 else (lambda t: t.permute(0, 2, 1))
    xr = to_cf(x).contiguous().requires_grad_(True)
    y_ref = (F.max_pool2d if dims == 2 else F.max_pool1d)(xr, 3, 2, 1)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    xd = x.to(DEV).requires_grad_(True)
    y = L.MaxPoolF32Fn.apply(xd, 3, 2, 1)
    to_cl = (lambda t: t.permute(0, 2, 3, 1)) if dims == 2 else (lambda t: t.permute(0, 2, 1))
    y.backward(to_cl(gy).contiguous().to(DEV))
    assert torch.equal(to_cf(y.cpu()), y_ref.detach())
    assert rel_l2(to_cf(xd.grad), xr.grad) < 1e-6
    for mode in ('avg', 'max'):
        feat, idx = L.global_pool_fwd(x.to(DEV), mode)
        flat = x.reshape(shape[0], -1, shape[-1])
        ref = flat.mean(1) if mode == 'avg' else flat.max(1).values
        assert rel_l2(feat, ref) < 1e-6
        d = torch.randn(feat.shape, generator=g)
        dx = L.global_pool_bwd(d.to(DEV), idx, x.shape, mode)
        fr = flat.clone().requires_grad_(True)
        (fr.mean(1) if mode == 'avg' else fr.max(1).values).backward(d)
        assert rel_l2(dx.reshape(flat.shape), fr.grad) < 1e-6


# ------------------------------------------------------------------------------------------------ reference fixtures
@pytest.mark.parametrize('tag', ['b8_2222', 'b16_1111'])
def test_profile_cnn_f32_matches_reference_fixtures(golden, f32_mode, tag):
    """Train-mode forward <= 1e-4, every parameter gradient <= 1e-3, running statistics <= 1e-5 against the fixture written
    by the reference's own ProfileCNN (tests/golden/make_golden.py; src/profile_encoder.py:111-240)."""
    from multimodal_plankton_recognition_amd.profile_encoder import ProfileCNN
    g = golden('profile_cnn_' + tag)
    m = ProfileCNN(dim_in=6, blocks=[int(b) for b in g['blocks']], base_channels=int(g['base']), dropout=0.0)
    _load(m, g)
    m.to(DEV)
    x, plen, wsum = T(g['profile']).to(DEV), T(g['profile_len']).to(DEV), T(g['wsum']).to(DEV)
    m.eval()
    with torch.no_grad():
        fm = m.forward_features(x)
        assert fm.dtype == torch.float32
        assert rel_l2(fm.transpose(1, 2), g['eval.features']) < 1e-5
        assert rel_l2(m(profile=x, profile_len=plen), g['eval.out']) < 1e-5
    m.train()
    y = m(profile=x, profile_len=plen)
    assert rel_l2(y, g['train.out']) < 1e-4
    (y * wsum).sum().backward()
    worst = max((rel_l2(v.grad, g['train.grad.' + k]), k) for k, v in m.named_parameters())
    assert worst[0] < 1e-3, worst
    sd = m.state_dict()
    for k in g:
        if k.startswith('train.after.'):
            name = k[len('train.after.'):]
            if name.endswith('num_batches_tracked'):
                assert int(sd[name]) == int(g[k])
            else:
                assert rel_l2(sd[name], g[k]) < 1e-5, name


def test_composed_step_f32_matches_reference_fixture(golden, f32_mode):
    """ProfileCNN -> projection || image features -> projection -> CLIP(buckets=2) -> 2x fused SGD: post-step parameters
    <= 1e-3 (the bf16 path holds 0.2 here)."""
    from multimodal_plankton_recognition_amd.profile_encoder import ProfileCNN
    from multimodal_plankton_recognition_amd.coordination import CLIPLoss
    from multimodal_plankton_recognition_amd.model import _BiasFreeLinear
    from multimodal_plankton_recognition_amd.ops import FusedSGD
    g = golden('composed_step')
    enc = ProfileCNN(dim_in=6, blocks=[1, 1, 1, 1], base_channels=8, dropout=0.0)
    pproj = _BiasFreeLinear(enc.dim_out, 32, bias=False)
    iproj = _BiasFreeLinear(18, 32, bias=False)
    loss_mod = CLIPLoss()
    mods = {'profile_encoder.': enc, 'profile_projection.': pproj, 'image_projection.': iproj, 'loss.': loss_mod}
    for pre, m in mods.items():
        _load(m, g, 'sd0.' + pre)
        m.to(DEV).train()
    params = [p for m in mods.values() for p in m.parameters()]
    opt = FusedSGD(params, lr=5e-2, momentum=0.9, weight_decay=1e-3, nesterov=True)
    for step in range(2):
        opt.zero_grad()
        feat = enc(profile=T(g[f'step{step}.profile']).to(DEV), profile_len=T(g[f'step{step}.profile_len']).to(DEV))
        loss = loss_mod(iproj(T(g[f'step{step}.image_feat']).to(DEV)), pproj(feat), 2)
        loss.backward()
        opt.step()
        assert abs(loss.item() - float(g[f'step{step}.loss'])) < 1e-4 * abs(float(g[f'step{step}.loss']))
    for pre, m in mods.items():
        for k, v in m.state_dict().items():
            if 'num_batches' in k:
                assert int(v) == int(g['sd2.' + pre + k])
            else:
                assert rel_l2(v, g['sd2.' + pre + k]) < 1e-3, pre + k


# ------------------------------------------------------------------------------------------------ oracle
def _resnet_case():
    from multimodal_plankton_recognition_amd.image_encoder import ImageEncoder
    torch.manual_seed(0)
    enc = ImageEncoder('resnet18', dropout=0.0)
    with torch.no_grad():       # timm zero-inits the last BN of each block: perturb so the branch matters
        for n_, p_ in enc.named_parameters():
            if n_.endswith('bn2.weight'):
                p_.fill_(0.5)
    g = torch.Generator().manual_seed(1)
    image = (torch.randn(8, 1, 96, 96, generator=g) * 0.3).clamp(-1, 1)
    shape = torch.randint(32, 400, (8, 2), generator=g)
    wsum = torch.randn(8, 514, generator=g)
    return enc, image, shape, wsum


def test_resnet18_f32_train_mode_matches_oracle(f32_mode):
    """Whole ResNet-18, train-mode BatchNorm: forward <= 1e-4, EVERY parameter gradient <= 1e-2 (measured ~1e-4),
    running statistics <= 1e-5 against the fp32 oracle; then eval mode on the updated statistics."""
    from oracle.image_encoder import image_encoder_forward
    enc, image, shape, wsum = _resnet_case()
    osd = {k: v.detach().cpu().clone() for k, v in enc.state_dict().items()}
    params = {k: v.requires_grad_(True) for k, v in osd.items() if v.is_floating_point() and 'running' not in k}
    ref = image_encoder_forward(osd, image, shape, arch='resnet18', train=True)
    (ref * wsum).sum().backward()
    enc.to(DEV).train()
    out = enc(image=image.to(DEV), image_shape=shape.to(DEV))
    (out * wsum.to(DEV)).sum().backward()
    assert rel_l2(out, ref) < 1e-4
    errs = sorted(((rel_l2(v.grad, params[k].grad), k) for k, v in enc.named_parameters()), reverse=True)
    print('ResNet-18 fp32 path, worst parameter-gradient errors:', errs[:3])
    assert errs[0][0] < 1e-2, errs[:3]
    new = enc.state_dict()
    for k in new:
        if 'running' in k:
            assert rel_l2(new[k], osd[k]) < 1e-5, k
    enc.eval()
    with torch.no_grad():
        out_e = enc(image=image.to(DEV), image_shape=shape.to(DEV))
    ref_e = image_encoder_forward({k: v.detach() for k, v in osd.items()}, image, shape, arch='resnet18', train=False)
    assert rel_l2(out_e, ref_e) < 1e-4


def test_f32_path_is_bitwise_reproducible(f32_mode):
    enc, image, shape, wsum = _resnet_case()
    enc.to(DEV).train()
    runs = []
    for _ in range(2):
        enc.zero_grad(set_to_none=True)
        out = enc(image=image.to(DEV), image_shape=shape.to(DEV))
        (out * wsum.to(DEV)).sum().backward()
        runs.append([out.detach().clone()] + [p.grad.clone() for p in enc.parameters()])
    assert all(torch.equal(a, b) for a, b in zip(*runs))


def test_multimodel_f32_steps_follow_oracle(f32_mode):
    """Three optimisation steps of MultiModel (ResNet-18 + ProfileCNN + CLIP + fused SGD) in fp32 mode: per-step loss
    <= 1e-4 and post-step parameters <= 1e-3 against the oracle's steps from the same state."""
    from multimodal_plankton_recognition_amd.model import MultiModel
    from oracle import model as OM
    import top1_task as TT
    cfg = dict(TT.CFG, optim_args=dict(TT.CFG['optim_args'], lr=5e-3))
    torch.manual_seed(0)
    model = MultiModel(dim_embed=TT.DIM_EMBED, **cfg)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    batches, _, _ = TT.make_task(3, 16, 0.4, n_test=12)
    model.to(DEV).train()
    opt = model.configure_optimizers()
    bufs = {}
    for b in batches:
        ref_loss, _ = OM.train_step(sd, b, cfg, bufs)
        opt.zero_grad()
        loss = model.training_step({k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in b.items()}, 0)
        loss.backward()
        opt.step()
        assert abs(loss.item() - ref_loss.item()) < 1e-4 * abs(ref_loss.item()), (loss.item(), ref_loss.item())
    new = model.state_dict()
    errs = sorted(((rel_l2(new[k], v), k) for k, v in sd.items() if v.is_floating_point()), reverse=True)
    print('MultiModel fp32 path after 3 steps, worst parameter errors:', errs[:3])
    assert errs[0][0] < 1e-3, errs[:3]
