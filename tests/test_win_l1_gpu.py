"""64 -> (<= 64)-channel 3x3 / stride 1 convolutions on the filter-in-registers kernel (csrc/conv_win.hip: conv_win_l1_kernel,
ResNet layer1 behind /root/reference/src/image_encoder.py:24): forward (+ BatchNorm partial sums) and the plain data gradient
against fp32 torch on the same bf16-rounded operands and -- bit for bit -- against the shifted-window kernels it replaces
(same products, same summation order), on ragged rasters, partial last tiles and an output width that is not 64."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def want_of(y, K):
    yq = y.double().reshape(-1, K)
    return torch.stack([yq.sum(0), (yq * yq).sum(0)])


@pytest.mark.parametrize('B,H,W,K', [(64, 56, 56, 64), (37, 19, 23, 64), (9, 56, 40, 64), (340, 7, 7, 64), (16, 33, 12, 40),
                                     (3, 56, 56, 64), (1, 8, 9, 8)])
def test_l1_kernel_matches_window_kernels_and_fp32(B, H, W, K):
    from multimodal_plankton_recognition_amd import ops, _native as N
    C = 64
    g = ops.ConvGeom((K, C, 3, 3), 1, 1)
    gen = torch.Generator().manual_seed(B * 131 + W)
    w = (torch.randn(K, C, 3, 3, generator=gen) * 0.05).to(DEV)
    wf, wd = ops.packed_weights(w, g)
    x = torch.randn(B, H, W, C, generator=gen).to(DEV).to(torch.bfloat16)
    dy = torch.randn(B, H, W, K, generator=gen).to(DEV).to(torch.bfloat16)
    base = N.query('mpr_conv_set_window_variant', 5)
    res = {}
    try:
        for name, var in (('l1', 5), ('old', 5 | 256), ('old16', 5 | 256 | 1024)):
            N.query('mpr_conv_set_window_variant', var)
            y, st = ops.conv_fwd(x, wf, g, True)
            dx = ops.conv_dgrad(dy, wd, g, tuple(x.shape))
            res[name] = (y, st.double().sum(0), dx)
    finally:
        N.query('mpr_conv_set_window_variant', base)
    wq = w.to(torch.bfloat16).float()
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), wq, padding=1).permute(0, 2, 3, 1)
    refd = torch.nn.functional.conv_transpose2d(dy.float().permute(0, 3, 1, 2), wq, padding=1).permute(0, 2, 3, 1)
    y1, s1, d1 = res['l1']
    y0, s0, d0 = res['old']
    assert torch.equal(y1, y0) and torch.equal(d1, d0)
    # conv_win_kernel on v_mfma_f32_16x16x32_bf16 (variant bit 10, the default): the same 32-term groups per instruction pair
    y6, s6, d6 = res['old16']
    assert torch.equal(y6, y0) and torch.equal(d6, d0)
    assert (s6 - want_of(y1, K)).abs().max().item() <= 1e-4 * want_of(y1, K).abs().max().item()
    assert (y1.float() - ref).abs().max().item() <= 8e-3 * ref.abs().max().item()          # one bf16 ulp of the largest value
    assert (d1.float() - refd).abs().max().item() <= 8e-3 * refd.abs().max().item()
    # BatchNorm partial sums of the ROUNDED output (what the apply pass normalises)
    want = want_of(y1, K)
    assert (s1 - want).abs().max().item() <= 1e-4 * want.abs().max().item()
    assert (s0 - want).abs().max().item() <= 1e-4 * want.abs().max().item()


@pytest.mark.parametrize('B,H,W', [(150, 19, 23), (30, 56, 40), (1400, 7, 7), (11, 56, 56)])
def test_l1_fused_data_gradient_epilogues_match_conv_win_kernel(B, H, W):
    """Skip add of the unrounded tile, ReLU mask + BatchNorm-backward sums (both mask modes): the gradient bit for bit, the sums
    to summation order, against conv_win_kernel's ADD / BNB epilogues (mpr_conv_set_window_variant bit 9)."""
    from multimodal_plankton_recognition_amd import ops, _native as N
    C = K = 64
    g = ops.ConvGeom((K, C, 3, 3), 1, 1)
    gen = torch.Generator().manual_seed(B + W)
    w = (torch.randn(K, C, 3, 3, generator=gen) * 0.05).to(DEV)
    _, wd = ops.packed_weights(w, g)
    dy = torch.randn(B, H, W, K, generator=gen).to(DEV).to(torch.bfloat16)
    bn_x = torch.randn(B, H, W, C, generator=gen).to(DEV).to(torch.bfloat16)
    res = torch.randn(B, H, W, C, generator=gen).to(DEV).to(torch.bfloat16)
    gamma = (torch.rand(C, generator=gen) + 0.5).to(DEV)
    beta = (torch.randn(C, generator=gen) * 0.3).to(DEV)
    xf = bn_x.float().reshape(-1, C)
    mean, var = xf.mean(0), xf.var(0, unbiased=False)

    class St:
        pass
    st = St()
    st.mean, st.invstd = mean.contiguous(), (var + 1e-5).rsqrt().contiguous()
    st.scale = (gamma * st.invstd).contiguous()
    st.shift = (beta - mean * st.scale).contiguous()
    mask_y = torch.relu(xf * st.scale + st.shift + res.float().reshape(-1, C)).to(torch.bfloat16).reshape(B, H, W, C)
    cases = [lambda: (ops.conv_dgrad(dy, wd, g, (B, H, W, C), add=res), None),
             lambda: ops.conv_dgrad_bn(dy, wd, g, (B, H, W, C), bn_x, st, 2),
             lambda: ops.conv_dgrad_bn(dy, wd, g, (B, H, W, C), bn_x, st, 1, mask_y=mask_y)]
    base = N.query('mpr_conv_set_window_variant', 5)
    arena, ops.SLICE_ARENA = ops.SLICE_ARENA, False
    try:
        for fn in cases:
            out = []
            for var in (5, 5 | 512, 5 | 512 | 1024):
                N.query('mpr_conv_set_window_variant', var)
                r = fn()
                assert r is not None
                out.append((r[0].clone(), None if r[1] is None else r[1].double().sum(0)))
            (d1, s1), (d0, s0), (d6, s6) = out
            assert torch.equal(d1, d0) and torch.equal(d6, d0)
            if s1 is not None:
                assert (s1 - s0).abs().max().item() <= 1e-3 * s0.abs().max().item()
                assert (s6 - s0).abs().max().item() <= 1e-3 * s0.abs().max().item()
    finally:
        N.query('mpr_conv_set_window_variant', base)
        ops.SLICE_ARENA = arena


@pytest.mark.parametrize('B,H,W,C', [(24, 28, 28, 128), (90, 14, 14, 256), (340, 7, 7, 512), (40, 19, 23, 128), (32, 28, 20, 192)])
def test_window_kernel_mfma_shapes_agree_bit_for_bit(B, H, W, C):
    """conv_win_kernel on v_mfma_f32_16x16x32_bf16 (variant bit 10, the default) against its 32x32x16 form on the wide tiles
    (N > 64: 8 waves): forward, plain data gradient and the fused data-gradient epilogues -- outputs bit for bit (an instruction
    pair sums the same 32-term group), BatchNorm sums to summation order."""
    from multimodal_plankton_recognition_amd import ops, _native as N
    K = C
    g = ops.ConvGeom((K, C, 3, 3), 1, 1)
    gen = torch.Generator().manual_seed(B * 7 + C)
    w = (torch.randn(K, C, 3, 3, generator=gen) * 0.05).to(DEV)
    wf, wd = ops.packed_weights(w, g)
    x = torch.randn(B, H, W, C, generator=gen).to(DEV).to(torch.bfloat16)
    dy = torch.randn(B, H, W, K, generator=gen).to(DEV).to(torch.bfloat16)
    res = torch.randn(B, H, W, C, generator=gen).to(DEV).to(torch.bfloat16)
    xf = x.float().reshape(-1, C)

    class St:
        pass
    st = St()
    st.mean = xf.mean(0).contiguous()
    st.invstd = (xf.var(0, unbiased=False) + 1e-5).rsqrt().contiguous()
    st.scale = (st.invstd * 1.1).contiguous()
    st.shift = (0.1 - st.mean * st.scale).contiguous()
    mask_y = torch.relu(xf * st.scale + st.shift + res.float().reshape(-1, C)).to(torch.bfloat16).reshape(B, H, W, C)
    cases = [lambda: ops.conv_fwd(x, wf, g, True), lambda: (ops.conv_dgrad(dy, wd, g, (B, H, W, C)), None),
             lambda: (ops.conv_dgrad(dy, wd, g, (B, H, W, C), add=res), None),
             lambda: ops.conv_dgrad_bn(dy, wd, g, (B, H, W, C), x, st, 2),
             lambda: ops.conv_dgrad_bn(dy, wd, g, (B, H, W, C), x, st, 1, mask_y=mask_y, add=res)]
    base = N.query('mpr_conv_set_window_variant', 5)
    arena, ops.SLICE_ARENA = ops.SLICE_ARENA, False
    try:
        for fn in cases:
            out = []
            for var in (5 | 512, 5 | 512 | 1024):
                N.query('mpr_conv_set_window_variant', var)
                r = fn()
                assert r is not None
                out.append((r[0].clone(), None if r[1] is None else r[1].double().sum(0)))
            (a, sa), (b, sb) = out
            assert torch.equal(a, b)
            if sa is not None:
                assert (sa - sb).abs().max().item() <= 1e-3 * sb.abs().max().item()
    finally:
        N.query('mpr_conv_set_window_variant', base)
        ops.SLICE_ARENA = arena
