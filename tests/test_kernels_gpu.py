"""GPU parity of every C-ABI kernel against a plain torch fp32 CPU computation of the same op, fed the
same bf16-rounded operands (so the only differences are accumulation order and the final bf16
rounding).  Tolerances: bf16 outputs 2^-8 relative (one bf16 ulp) + small absolute; fp32 outputs 1e-4.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def _ops():
    from multimodal_plankton_recognition_amd import ops
    return ops


def bf(x):
    return x.to(torch.bfloat16)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def assert_close_bf16(got, ref, what, rtol=1.0 / 128, atol=2e-2):
    got = got.float().cpu()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    assert not bad.any(), f'{what}: {int(bad.sum())}/{bad.numel()} off, max err {err.max():.4g} (ref max {ref.abs().max():.4g})'


def nhwc(x):      # NCHW -> NHWC contiguous
    return x.permute(0, 2, 3, 1).contiguous()


CONV_CASES = [
    # B, H, W, C, K, R, stride, pad
    (2, 12, 12, 64, 64, 3, 1, 1),
    (3, 13, 9, 64, 128, 3, 2, 1),
    (2, 8, 8, 128, 128, 3, 1, 1),
    (2, 10, 10, 64, 128, 1, 2, 0),
    (5, 7, 7, 256, 512, 3, 2, 1),
    (4, 14, 14, 32, 32, 3, 1, 1),
    (2, 9, 11, 32, 64, 3, 2, 1),
    (1, 20, 20, 8, 16, 3, 1, 1),
    # even extents at stride 2: the data gradient runs as 4 parity classes (ragged class tiles, >1 tile per class)
    (3, 12, 16, 64, 128, 3, 2, 1),
    (2, 14, 14, 256, 512, 3, 2, 1),
    (9, 20, 12, 64, 64, 3, 2, 1),
    # 3x3 / stride 1 / pad 1 with 64-channel blocks: the shifted-window kernel (several tiles, widest halo, ragged N
    # tile, several channel blocks, images smaller than the halo)
    (2, 56, 56, 64, 64, 3, 1, 1),
    (3, 9, 7, 64, 192, 3, 1, 1),
    (5, 7, 7, 256, 128, 3, 1, 1),
    (40, 4, 3, 128, 64, 3, 1, 1),
    (1, 2, 2, 64, 8, 3, 1, 1),
]


@pytest.fixture(params=['regs', 'dma', 'win'])
def igemm_path(request):
    """Run the conv tests through every implicit-GEMM kernel: register-staged / LDS-DMA ring / LDS-DMA ring with the
    shifted-window kernel taking the 3x3 stride-1 pad-1 cases."""
    from multimodal_plankton_recognition_amd import _native
    thr = 1 << 30 if request.param == 'regs' else 0
    old_win = _native.query('mpr_conv_set_window', 1 if request.param == 'win' else 0)
    request.addfinalizer(lambda: _native.query('mpr_conv_set_window', old_win))
    old_mw = _native.query('mpr_conv_set_window_fwd_min_width', 0)      # the window kernel on narrow maps too
    request.addfinalizer(lambda: _native.query('mpr_conv_set_window_fwd_min_width', old_mw))
    old_ww = _native.query('mpr_conv_set_wgrad_window', 1 if request.param == 'win' else 0)
    request.addfinalizer(lambda: _native.query('mpr_conv_set_wgrad_window', old_ww))
    old = _native.query('mpr_conv_set_dma_min_rows', thr)
    old_w = _native.query('mpr_conv_set_wgrad_dma_min_pixels', thr)
    old_p = _native.query('mpr_conv_set_dgrad_parity', 2)      # parity classes for 1x1 filters too
    yield request.param
    _native.query('mpr_conv_set_dgrad_parity', old_p)
    _native.query('mpr_conv_set_dma_min_rows', old)
    _native.query('mpr_conv_set_wgrad_dma_min_pixels', old_w)


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv2d_fwd_dgrad_wgrad(case, igemm_path):
    ops = _ops()
    B, H, W, C, K, R, stride, pad = case
    x = bf(rnd(B, C, H, W, seed=1)).float()
    w = bf(rnd(K, C, R, R, seed=2, scale=(C * R * R) ** -0.5)).float()
    g = ops.ConvGeom((K, C, R, R), stride, pad)
    wt = w.to(DEV)
    wf, wd = ops.packed_weights(wt, g)
    xd = bf(nhwc(x)).to(DEV)
    y, stats = ops.conv_fwd(xd, wf, g, True)
    ref = F.conv2d(x, w, None, stride, pad)
    assert_close_bf16(y, nhwc(ref), 'conv fwd')
    # fused BN partial sums == sums of the stored (rounded) output
    yr = y.float().cpu().reshape(-1, K)
    s = stats.cpu().sum(0)
    np.testing.assert_allclose(s[0].numpy(), yr.sum(0).numpy(), rtol=2e-3, atol=2e-2)
    np.testing.assert_allclose(s[1].numpy(), (yr * yr).sum(0).numpy(), rtol=2e-3, atol=2e-2)
    # dgrad (+ fused residual) and wgrad
    dy = bf(rnd(*ref.shape, seed=3)).float()
    xg = x.clone().requires_grad_(True)
    wg = w.clone().requires_grad_(True)
    F.conv2d(xg, wg, None, stride, pad).backward(dy)
    dyd = bf(nhwc(dy)).to(DEV)
    dx = ops.conv_dgrad(dyd, wd, g, xd.shape)
    assert_close_bf16(dx, nhwc(xg.grad), 'conv dgrad')
    add = bf(rnd(B, H, W, C, seed=4))
    dx2 = ops.conv_dgrad(dyd, wd, g, xd.shape, add=add.to(DEV))
    assert_close_bf16(dx2, nhwc(xg.grad) + add.float(), 'conv dgrad + add')
    dw = ops.conv_wgrad(xd, dyd, g, (K, C, R, R))
    np.testing.assert_allclose(dw.cpu().numpy(), wg.grad.numpy(), rtol=2e-3, atol=2e-3 * float(wg.grad.abs().max()))


@pytest.mark.parametrize('case', [(3, 40, 32, 32, 3, 1, 1), (2, 33, 32, 64, 3, 2, 1), (4, 17, 64, 128, 1, 2, 0),
                                  (2, 24, 8, 8, 3, 1, 1)])
def test_conv1d_fwd_dgrad_wgrad(case, igemm_path):
    ops = _ops()
    B, L, C, K, S, stride, pad = case
    x = bf(rnd(B, C, L, seed=5)).float()
    w = bf(rnd(K, C, S, seed=6, scale=(C * S) ** -0.5)).float()
    g = ops.ConvGeom((K, C, S), stride, pad)
    wf, wd = ops.packed_weights(w.to(DEV), g)
    xd = bf(x.transpose(1, 2).contiguous()).to(DEV)          # [B, L, C]
    y, _ = ops.conv_fwd(xd, wf, g, False)
    ref = F.conv1d(x, w, None, stride, pad)
    assert_close_bf16(y, ref.transpose(1, 2), 'conv1d fwd')
    dy = bf(rnd(*ref.shape, seed=7)).float()
    xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    F.conv1d(xg, wg, None, stride, pad).backward(dy)
    dyd = bf(dy.transpose(1, 2).contiguous()).to(DEV)
    assert_close_bf16(ops.conv_dgrad(dyd, wd, g, xd.shape), xg.grad.transpose(1, 2), 'conv1d dgrad')
    dw = ops.conv_wgrad(xd, dyd, g, (K, C, S))
    np.testing.assert_allclose(dw.cpu().numpy(), wg.grad.numpy(), rtol=2e-3, atol=2e-3 * float(wg.grad.abs().max()))


def test_conv_large_rows_many_tiles(igemm_path):
    """More than one wave of workgroups, M not a multiple of the tile, ResNet layer1 shape."""
    ops = _ops()
    B, H, W, C, K = 6, 56, 56, 64, 64
    x = bf(rnd(B, C, H, W, seed=8)).float()
    w = bf(rnd(K, C, 3, 3, seed=9, scale=0.05)).float()
    g = ops.ConvGeom((K, C, 3, 3), 1, 1)
    wf, wd = ops.packed_weights(w.to(DEV), g)
    xd = bf(nhwc(x)).to(DEV)
    y, _ = ops.conv_fwd(xd, wf, g, True)
    assert_close_bf16(y, nhwc(F.conv2d(x, w, None, 1, 1)), 'conv fwd layer1')
    dy = bf(rnd(B, K, H, W, seed=10)).float()
    xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    F.conv2d(xg, wg, None, 1, 1).backward(dy)
    dw = ops.conv_wgrad(xd, bf(nhwc(dy)).to(DEV), g, (K, C, 3, 3))
    np.testing.assert_allclose(dw.cpu().numpy(), wg.grad.numpy(), rtol=3e-3, atol=3e-3 * float(wg.grad.abs().max()))


@pytest.mark.parametrize('dims', [2, 1, 3])
def test_stem_fwd_and_wgrad(dims):
    ops = _ops()
    if dims in (2, 3):
        # (3: EfficientNet's stem -- one channel, 3 x 3, stride 2: the all-taps-per-thread weight gradient of stem.hip)
        B, H, W, Cin, K, R, st, pad = (3, 30, 26, 1, 64, 7, 2, 3) if dims == 2 else (5, 37, 30, 1, 32, 3, 2, 1)
        x = rnd(B, Cin, H, W, seed=11)
        w = rnd(K, Cin, R, R, seed=12, scale=0.1)
        g = ops.ConvGeom((K, Cin, R, R), st, pad)
        ref = F.conv2d(x, w, None, st, pad)
        xin = x.reshape(B, H, W, 1).to(DEV)
        to_cl = nhwc
    else:
        B, L, Cin, K, S, st, pad = 5, 50, 6, 32, 3, 2, 1
        x = rnd(B, Cin, L, seed=13)
        w = rnd(K, Cin, S, seed=14, scale=0.2)
        g = ops.ConvGeom((K, Cin, S), st, pad)
        ref = F.conv1d(x, w, None, st, pad)
        xin = x.transpose(1, 2).contiguous().to(DEV)
        to_cl = lambda t: t.transpose(1, 2).contiguous()
    y, stats = ops.stem_fwd(xin, w.to(DEV), g, True)
    assert_close_bf16(y, to_cl(ref), 'stem fwd', atol=1e-2)
    yr = y.float().cpu().reshape(-1, K)
    np.testing.assert_allclose(stats.cpu().sum(0)[0].numpy(), yr.sum(0).numpy(), rtol=1e-3, atol=1e-2)
    dy = bf(rnd(*ref.shape, seed=15)).float()
    wg = w.clone().requires_grad_(True)
    (F.conv2d(x, wg, None, st, pad) if dims != 1 else F.conv1d(x, wg, None, st, pad)).backward(dy)
    dw = ops.stem_wgrad(xin, bf(to_cl(dy)).to(DEV), g, tuple(w.shape))
    np.testing.assert_allclose(dw.cpu().numpy(), wg.grad.numpy(), rtol=1e-3, atol=1e-3 * float(wg.grad.abs().max()))


class _BN:
    def __init__(self, C, seed):
        g = torch.Generator().manual_seed(seed)
        self.weight = (torch.rand(C, generator=g) + 0.5).to(DEV)
        self.bias = (torch.rand(C, generator=g) - 0.5).to(DEV)
        self.running_mean = (torch.rand(C, generator=g) - 0.5).to(DEV)
        self.running_var = (torch.rand(C, generator=g) + 0.5).to(DEV)
        self.momentum, self.eps = 0.1, 1e-5


@pytest.mark.parametrize('C,rows,res', [(64, 1000, False), (32, 777, True), (512, 130, True), (8, 5000, False),
                                        (24, 301, True)])
def test_batchnorm_train_fwd_bwd(C, rows, res):
    ops = _ops()
    x = bf(rnd(rows, C, seed=20) * 1.5 + 0.3)
    r = bf(rnd(rows, C, seed=21)) if res else None
    bn = _BN(C, 22)
    rm0, rv0 = bn.running_mean.clone().cpu(), bn.running_var.clone().cpu()
    xd = x.to(DEV).reshape(1, rows, C)
    st = ops.bn_coefs(None, rows, bn, True, xd)
    y = ops.bn_apply(xd, st, r.to(DEV).reshape(1, rows, C) if res else None, True)
    # reference (fp32 on the bf16-rounded input)
    xr = x.float().requires_grad_(True)
    gamma = bn.weight.cpu().clone().requires_grad_(True)
    beta = bn.bias.cpu().clone().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    z = F.batch_norm(xr, rm, rv, gamma, beta, True, 0.1, 1e-5)
    if res:
        z = z + r.float()
    ref = F.relu(z)
    assert_close_bf16(y, ref.detach().reshape(1, rows, C), 'bn apply')
    np.testing.assert_allclose(bn.running_mean.cpu().numpy(), rm.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), rv.numpy(), rtol=1e-4, atol=1e-5)
    dy = bf(rnd(rows, C, seed=23))
    # use the GPU's own (rounded) output as the relu mask in the reference to avoid sign flips at 0
    mask = (y.float().cpu().reshape(rows, C) > 0).float()
    (z * mask).backward(dy.float())
    dx, dgamma, dbeta, dz = ops.bn_bwd(dy.to(DEV).reshape(1, rows, C), y, xd, bn.weight, st, ops.MASK_Y, want_dz=True)
    assert_close_bf16(dx, xr.grad.reshape(1, rows, C), 'bn bwd dx', atol=3e-2)
    np.testing.assert_allclose(dgamma.cpu().numpy(), gamma.grad.numpy(), rtol=2e-3, atol=5e-2)
    np.testing.assert_allclose(dbeta.cpu().numpy(), beta.grad.numpy(), rtol=2e-3, atol=5e-2)
    assert_close_bf16(dz, (dy.float() * mask).reshape(1, rows, C), 'bn bwd dz')


@pytest.mark.parametrize('C,rows', [(128, 3000), (512, 130), (8, 5000), (24, 301)])
@pytest.mark.parametrize('mode', ['pending', 'finalized', 'eval'])
def test_projection_shortcut_output_in_one_pass(C, rows, mode):
    """act(BN(x) + bf16(BN_r(xr))) (mpr_bn_apply_dual) == the two passes it replaces, bit for bit: output, saved
    statistics and running statistics -- with the statistics finalized inside the kernel (what a training block does),
    finalized before it, and in eval mode."""
    ops = _ops()
    x = bf(rnd(rows, C, seed=30) * 1.5 + 0.3).to(DEV).reshape(1, rows, C)
    xr = bf(rnd(rows, C, seed=31) * 0.7 - 0.2).to(DEV).reshape(1, rows, C)
    train, defer = mode != 'eval', mode == 'pending'

    def run(dual):
        bn, bnr = _BN(C, 32), _BN(C, 33)
        st = ops.bn_coefs(None, rows, bn, train, x, defer=defer, want_bwd=True)
        str_ = ops.bn_coefs(None, rows, bnr, train, xr, defer=defer, want_bwd=True)
        assert (st.pending is not None) == defer
        if dual:
            y = ops.bn_apply_dual(x, st, xr, str_, True)
            assert y is not None and st.pending is None and str_.pending is None
        else:
            y = ops.bn_apply(x, st, ops.bn_apply(xr, str_, None, False), True)
        return y, [st.scale, st.shift, st.mean, st.invstd, str_.scale, str_.shift, str_.mean, str_.invstd,
                   bn.running_mean, bn.running_var, bnr.running_mean, bnr.running_var]
    y1, s1 = run(True)
    y0, s0 = run(False)
    assert torch.equal(y1, y0)
    for a, b in zip(s1, s0):
        assert torch.equal(a, b)
    ref = F.relu(F.batch_norm(x.float().cpu()[0], None, None, _BN(C, 32).weight.cpu(), _BN(C, 32).bias.cpu(), True)
                 + F.batch_norm(xr.float().cpu()[0], None, None, _BN(C, 33).weight.cpu(), _BN(C, 33).bias.cpu(), True)) \
        if train else None
    if ref is not None:
        assert_close_bf16(y1, ref.reshape(1, rows, C), 'dual bn apply', atol=3e-2)


def test_bn_relu_maxpool_2d_and_1d():
    ops = _ops()
    for dims in (2, 1):
        C = 64
        shape = (3, 17, 15, C) if dims == 2 else (4, 29, C)
        x = bf(rnd(*shape, seed=30))
        bn = _BN(C, 31)
        xd = x.to(DEV)
        rows = x.numel() // C
        st = ops.bn_coefs(None, rows, bn, True, xd)
        y, idx = ops.bn_relu_maxpool_fwd(xd, st)
        a = F.relu(x.float() * st.scale.cpu() + st.shift.cpu()).to(torch.bfloat16).float()
        if dims == 2:
            ref, ridx = F.max_pool2d(a.permute(0, 3, 1, 2), 3, 2, 1, return_indices=True)
            ref = ref.permute(0, 2, 3, 1)
        else:
            ref, ridx = F.max_pool1d(a.transpose(1, 2), 3, 2, 1, return_indices=True)
            ref = ref.transpose(1, 2)
        assert torch.equal(y.float().cpu(), ref.contiguous()), 'maxpool values must be exact'
        dy = bf(rnd(*ref.shape, seed=32))
        da = ops.maxpool_bwd(dy.to(DEV), idx, xd.shape)
        ag = a.clone().requires_grad_(True)
        if dims == 2:
            F.max_pool2d(ag.permute(0, 3, 1, 2), 3, 2, 1).backward(dy.float().permute(0, 3, 1, 2))
        else:
            F.max_pool1d(ag.transpose(1, 2), 3, 2, 1).backward(dy.float().transpose(1, 2))
        # ties (post-ReLU zeros) may pick a different tap; compare where the activation is positive
        pos = a > 0
        assert_close_bf16(da.float().cpu()[pos], ag.grad[pos], 'maxpool bwd', atol=1e-2)


def test_global_pools():
    ops = _ops()
    x = bf(rnd(5, 7, 7, 512, seed=40))
    y, _ = ops.global_pool_fwd(x.to(DEV), 'avg')
    np.testing.assert_allclose(y.cpu().numpy(), x.float().mean((1, 2)).numpy(), rtol=1e-5, atol=1e-6)
    dy = rnd(5, 512, seed=41)
    dx = ops.global_pool_bwd(dy.to(DEV), None, x.shape, 'avg')
    assert_close_bf16(dx, (dy / 49)[:, None, None, :].expand(5, 7, 7, 512), 'avgpool bwd', atol=1e-4)
    x1 = bf(rnd(6, 7, 256, seed=42))
    y1, idx = ops.global_pool_fwd(x1.to(DEV), 'max')
    ref, ridx = x1.float().max(1)
    assert torch.equal(y1.cpu(), ref)
    dx1 = ops.global_pool_bwd(rnd(6, 256, seed=43).to(DEV), idx, x1.shape, 'max')
    expect = torch.zeros(6, 7, 256).scatter_(1, ridx[:, None, :], rnd(6, 256, seed=43)[:, None, :])
    assert_close_bf16(dx1, expect, 'maxpool1d bwd', atol=1e-4)


@pytest.mark.parametrize('M,N,K,ta,tb', [(70, 50, 33, False, True), (128, 512, 514, False, True),
                                         (257, 96, 64, True, False), (64, 64, 16, False, False), (5, 3, 7, True, True)])
def test_gemm_f32_exact(M, N, K, ta, tb):
    ops = _ops()
    a = rnd(*((K, M) if ta else (M, K)), seed=50)
    b = rnd(*((N, K) if tb else (K, N)), seed=51)
    bias = rnd(N, seed=52)
    out = ops.gemm(a.to(DEV), b.to(DEV), ta, tb, bias=bias.to(DEV))
    ref = (a.T if ta else a).double() @ (b.T if tb else b).double() + bias.double()
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=2e-5, atol=2e-4)   # fp32 chain over K<=514
    ab = rnd(3, M, K, seed=53)
    bb = rnd(3, N, K, seed=54)
    outb = ops.gemm(ab.to(DEV), bb.to(DEV), False, True)
    np.testing.assert_allclose(outb.cpu().numpy(), (ab.double() @ bb.double().transpose(1, 2)).numpy(), rtol=2e-5, atol=2e-4)


def test_tail_and_softmax_ce():
    ops = _ops()
    feat = rnd(9, 20, seed=60)
    meta = torch.randint(1, 500, (9, 2))
    out, mask = ops.tail_fwd(feat.to(DEV), meta.to(DEV), 224, 0.0, 0)
    np.testing.assert_allclose(out.cpu().numpy(), torch.cat((feat, meta.float() / 224), 1).numpy(), rtol=1e-6)
    out, mask = ops.tail_fwd(feat.to(DEV), meta.to(DEV), 224, 0.25, 1234)
    m = mask.cpu().bool()
    ref = torch.cat((feat, meta.float() / 224), 1)
    np.testing.assert_allclose(out.cpu().numpy(), torch.where(m, ref / 0.75, torch.zeros_like(ref)).numpy(), rtol=1e-6)
    big, bm = ops.tail_fwd(rnd(512, 512, seed=61).to(DEV), None, 1, 0.1, 77)
    assert abs(bm.float().mean().item() - 0.9) < 0.01
    dfeat = ops.tail_bwd(torch.ones(9, 22, device=DEV), mask, 0.25, 20)
    np.testing.assert_allclose(dfeat.cpu().numpy(), (m[:, :20].float() / 0.75).numpy(), rtol=1e-6)
    logits = rnd(33, 50, seed=62) * 3
    labels = torch.randint(0, 50, (33,))
    loss, argmax, dl = ops.softmax_ce(logits.to(DEV), labels.to(DEV), want_grad=True)
    lr = logits.clone().requires_grad_(True)
    ref = F.cross_entropy(lr, labels)
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-5
    assert torch.equal(argmax.cpu(), logits.argmax(1))          # class indices: bit-exact
    np.testing.assert_allclose(dl.cpu().numpy(), lr.grad.numpy(), rtol=1e-4, atol=1e-7)


def test_fused_sgd_matches_torch():
    ops = _ops()
    shapes = [(64, 1, 7, 7), (64,), (128, 64, 3, 3), (), (513,), (3, 5)]
    ps = [rnd(*s, seed=70 + i) if s else torch.tensor(1.0) for i, s in enumerate(shapes)]
    ref = [p.clone().requires_grad_(True) for p in ps]
    mine = [p.clone().to(DEV).requires_grad_(True) for p in ps]
    kw = dict(lr=5e-3, momentum=0.9, weight_decay=1e-3, nesterov=True)
    o_ref = torch.optim.SGD(ref, **kw)
    o_mine = ops.FusedSGD(mine, **kw)
    for step in range(3):
        for i, (a, b) in enumerate(zip(ref, mine)):
            g = rnd(*a.shape, seed=100 + 10 * step + i) if a.dim() else torch.tensor(0.3 * (step + 1))
            a.grad = g.clone()
            b.grad = g.clone().to(DEV)
        o_ref.step()
        o_mine.step()
    for a, b in zip(ref, mine):
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().numpy(), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('shape', [(64, 64, 3, 3), (24, 40, 3, 3), (128, 64, 1, 1), (72, 8, 4, 4)])
def test_multi_tensor_weight_pack_equals_single_pack(shape):
    """The one-launch repack of every filter panel (tiled transpose for the data-gradient panel) must reproduce the
    single-filter pack bit for bit, for OIHW and for channels-last ([K][R][S][C]) weight memory."""
    ops = _ops()
    from multimodal_plankton_recognition_amd import _native as N
    K, C, R, S = shape
    g = ops.ConvGeom(shape, 1, R // 2)
    for krsc in (False, True):
        w = rnd(*shape, seed=5).to(DEV)
        if krsc:
            w = w.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        wf, wd = ops.packed_weights(w, g)
        # the optimizer changes the filter in place; the repack refreshes the BODY of the existing panels (their zero
        # padding is written once, by the first pack)
        with torch.no_grad():
            w.mul_(1.5).add_(0.25)
        ops.pack_registry.repack_all()
        wf_ref, wd_ref = torch.empty_like(wf), torch.empty_like(wd)
        N.call('mpr_conv_pack_weights_strided', w, *ops._kcrs_strides(w), wf_ref, wd_ref, K, C, R, S)
        assert torch.equal(wf, wf_ref) and torch.equal(wd, wd_ref), (shape, krsc)


def test_wide_linear_takes_the_256x256_tile_and_matches_torch():
    """Transformer-sized 1x1 'convolution' (M >= 8192 rows, N >= 1536 outputs): the 16-wave 256 x 256 tile, with a ragged
    M tail and a ragged N tail, forward and data gradient against torch on identical bf16 operands."""
    ops = _ops()
    B, L, C, K = 40, 205, 64, 1544
    x = bf(rnd(B, L, C, seed=21)).float()
    w = bf(rnd(K, C, seed=22, scale=C ** -0.5)).float()
    g = ops.ConvGeom((K, C))
    wf, wd = ops.packed_weights(w.to(DEV), g)
    y, _ = ops.conv_fwd(bf(x).to(DEV), wf, g, False)
    assert_close_bf16(y, x @ w.t(), 'wide linear fwd')
    dy = bf(rnd(B, L, K, seed=23)).float()
    g2 = ops.ConvGeom((C, K))                      # data gradient of a K -> C linear == wide-output GEMM dy W
    w2 = bf(rnd(C, K, seed=24, scale=K ** -0.5)).float()
    _, wd2 = ops.packed_weights(w2.to(DEV), g2)
    dx = ops.conv_dgrad(bf(rnd(B, L, C, seed=25)).to(DEV), wd2, g2, (B, L, K))
    assert_close_bf16(dx, bf(rnd(B, L, C, seed=25)).float() @ w2, 'wide linear dgrad')
