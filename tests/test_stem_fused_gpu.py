"""Fused / recomputed ResNet stem (csrc/stem_fused.hip: conv 7x7/2 -> BN -> ReLU -> maxpool 3x3/2 without the
full-resolution map) against the oracle with bf16-storage emulation, against the unfused kernels, and in eval mode
(running statistics) forward + backward.  timm ResNet conv1/bn1/act1/maxpool behind src/image_encoder.py:16,24."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rel_l2(got, ref):
    got = got.detach().float().cpu()
    ref = torch.as_tensor(ref).detach().float().cpu()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-12))


def _backbone(seed=1):
    from multimodal_plankton_recognition_amd.image_encoder import ResNetBackbone
    torch.manual_seed(seed)
    m = ResNetBackbone((1, 1, 1, 1), 1)
    with torch.no_grad():
        m.bn1.weight.copy_(torch.rand_like(m.bn1.weight) + 0.5)
        m.bn1.bias.copy_(torch.rand_like(m.bn1.bias) - 0.5)
        m.bn1.running_mean.copy_(torch.randn_like(m.bn1.running_mean) * 0.1)
        m.bn1.running_var.copy_(torch.rand_like(m.bn1.running_var) + 0.5)
    return m


def _oracle(sd, x, train):
    """x [B,1,H,W] fp32 -> pooled [B,64,H/4,W/4] with the HIP path's rounding points (autograd-able in sd): bf16
    operands, fp32 conv output (it never leaves the registers), bf16 activation."""
    from oracle.rounding import emulate_bf16, r
    from oracle.profile_encoder import _bn
    with emulate_bf16():
        o = F.conv2d(r(x), r(sd['conv1.weight']), None, 2, 3)
        return r(F.max_pool2d(F.relu(_bn(sd, 'bn1', o, train)), 3, 2, 1))    # (rounding commutes with max)


SHAPES = [(3, 64, 64), (2, 96, 128), (2, 224, 224), (5, 32, 96)]


@pytest.mark.parametrize('B,H,W', SHAPES)
@pytest.mark.parametrize('train', [True, False])
def test_fused_stem_vs_emulated_oracle(B, H, W, train):
    from multimodal_plankton_recognition_amd import ops
    from multimodal_plankton_recognition_amd.layers import StemFn
    m = _backbone()
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = ((torch.randn(B, 1, H, W, generator=g) * 0.0938 + 0.6136).clamp(0, 1) * 2 - 1)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items() if k.startswith(('conv1', 'bn1'))}
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and 'running' not in k}
    ref = _oracle(sd, x, train)
    wsum = torch.randn(ref.shape, generator=g).to(torch.bfloat16).float()
    (ref * wsum).sum().backward()
    m.to(DEV).train(train)
    xin = x.reshape(B, H, W, 1).to(DEV)
    assert ops.stemf_ok(xin, m.geom)
    out = StemFn.apply(xin, m.conv1.weight, m.bn1.weight, m.bn1.bias, m)
    assert out.shape == (B, H // 4, W // 4, 64) and out.dtype == torch.bfloat16
    assert rel_l2(out.permute(0, 3, 1, 2), ref) < 1e-3
    (out.float() * wsum.permute(0, 2, 3, 1).to(DEV)).sum().backward()
    for k in ('conv1.weight', 'bn1.weight', 'bn1.bias'):
        got = dict(m.named_parameters())[k].grad
        assert rel_l2(got, params[k].grad) < 1e-2, (k, rel_l2(got, params[k].grad))
    if train:
        new = m.state_dict()
        for k in ('bn1.running_mean', 'bn1.running_var'):
            assert rel_l2(new[k], sd[k]) < 1e-4, k


@pytest.mark.parametrize('B,H,W', [(4, 64, 96), (2, 224, 224)])
def test_fused_stem_vs_unfused_kernels(B, H, W):
    """Same arithmetic as the space-to-depth conv + fused BN/pool kernels, except that those round the conv output to
    bf16 on its way through HBM: activations within a bf16 ulp."""
    from multimodal_plankton_recognition_amd import ops
    from multimodal_plankton_recognition_amd.layers import StemFn
    g = torch.Generator().manual_seed(7)
    x = ((torch.randn(B, H, W, 1, generator=g) * 0.0938 + 0.6136).clamp(0, 1) * 2 - 1).to(DEV)
    dout = torch.randn(B, H // 4, W // 4, 64, generator=g).to(torch.bfloat16).to(DEV)
    res = {}
    for fused in (True, False):
        m = _backbone().to(DEV).train()
        ops.STEM_FUSED = fused
        try:
            out = StemFn.apply(x, m.conv1.weight, m.bn1.weight, m.bn1.bias, m)
            out.backward(dout)
        finally:
            ops.STEM_FUSED = True
        res[fused] = (out.detach().float(), m.conv1.weight.grad, m.bn1.weight.grad, m.bn1.bias.grad,
                      m.bn1.running_mean.clone(), m.bn1.running_var.clone())
    a, b = res[True], res[False]
    assert rel_l2(a[0], b[0]) < 4e-3
    assert float((a[0] - b[0]).abs().max()) <= float(b[0].abs().max()) * 2 ** -6
    # (gradients: the two paths pick different winners wherever the differently rounded activations tie or flip)
    for i in (1, 2, 3):
        assert rel_l2(a[i], b[i]) < 0.15, i
    for i in (4, 5):
        assert rel_l2(a[i], b[i]) < 1e-4, i


def test_fused_stem_argmax_codes():
    """idx = the maximum of the fp32 activations, first in torch's (kh, kw) scan order among equals (15 where the pooled
    activation is 0), checked against a torch restatement (recomputed in fp32 from the same bf16 operands)."""
    from multimodal_plankton_recognition_amd import ops
    m = _backbone().to(DEV).train()
    B, H, W = 2, 64, 64
    g = torch.Generator().manual_seed(3)
    x = ((torch.randn(B, H, W, 1, generator=g) * 0.3).clamp(-1, 1)).to(DEV)
    pooled, st, (xb, wp, idx) = ops.stemf_forward(x, m.conv1.weight, m.bn1, True, True)
    xr = x.permute(0, 3, 1, 2).to(torch.bfloat16).float()
    wr = m.conv1.weight.detach().to(torch.bfloat16).float()
    y = F.conv2d(xr, wr, None, 2, 3)
    a = F.relu(y * st.scale.view(1, -1, 1, 1) + st.shift.view(1, -1, 1, 1))
    ref, ridx = F.max_pool2d(a, 3, 2, 1, return_indices=True)          # arg-max on the fp32 activations, as the reference
    ref = ref.to(torch.bfloat16).float()
    got = pooled.float().permute(0, 3, 1, 2)
    same = got == ref
    assert float(same.float().mean()) > 0.99                     # (fp32 summation order: isolated 1-ulp flips)
    Q = W // 2
    code = idx.permute(0, 3, 1, 2).long()
    assert bool(((code == 15) == (got == 0)).all())
    ph = torch.arange(H // 4, device=DEV).view(1, 1, -1, 1) * 2 - 1 + code // 3
    pw = torch.arange(W // 4, device=DEV).view(1, 1, 1, -1) * 2 - 1 + code % 3
    sel = same & (got > 0)
    assert float(((ph * Q + pw) == ridx)[sel].float().mean()) > 0.999
    assert int(sel.sum()) > 1000
