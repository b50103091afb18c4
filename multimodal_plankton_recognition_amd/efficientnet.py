"""EfficientNet-B0 image backbone on the gfx950 kernels -- the ``timm.create_model('efficientnet_b0', ...)`` the
reference's cards name (/root/reference/model_cards/example_multi.yaml:9, scripts/train_multi.sh:9-14) behind
/root/reference/src/image_encoder.py:16,24.  timm is an un-vendored dependency: the topology is restated from its
published definition (stem 3x3/2 -> 32; MBConv stages [1x16 k3, 2x24 k3/2, 2x40 k5/2, 3x80 k3/2, 3x112 k5, 4x192 k5/2,
1x320 k3], expansion 6 (1 in the first), squeeze-excite at 1/4 of the block INPUT width, SiLU, BatchNorm eps 1e-5; head
1x1 -> 1280) with timm's ``state_dict`` key names (``conv_stem``, ``bn1``, ``blocks.S.I.{conv_pw,bn1,conv_dw,bn2,
se.conv_reduce,se.conv_expand,conv_pwl,bn3}``, ``conv_head``, ``bn2``), so a timm checkpoint loads unchanged.

Kernels: 1x1 expansions / projections / head = bf16 MFMA implicit-GEMM convs with fused BatchNorm partial sums
(``ops.conv_*``); depthwise convs and the SE gate = ``csrc/dwconv.hip``; BatchNorm = ``csrc/batchnorm.hip``; SiLU = the bf16
elementwise passes of ``csrc/transformer_bf16.hip``; the SE bottleneck (two tiny FCs on pooled [B, C]) = exact-fp32 GEMM.
Small autograd Functions chained by autograd (conv+BN, SiLU, SE): this backbone is built for coverage, not yet tuned.
"""
import math

import torch
from torch import Tensor, nn

from . import _native as N
from . import ops
from .layers import BatchNormParams, _bn_coefs, _rows
from .ops import ConvGeom, BF16, F32

_SILU, _SIGMOID = 3, 4
MASK_SILU = 3          # batchnorm.hip: dz = dy * silu'(x * scale + shift)


# ---------------------------------------------------------------------------------------------- kernel wrappers
def _dw_args(x, g):
    B, H, W, C = x.shape
    return B, H, W, C, g.R, g.S, g.sh, g.sw, g.ph, g.pw


def dwconv_fwd(x, w, g):
    B, H, W, C = x.shape
    P, Q = g.out_hw(H, W)
    y = torch.empty(B, P, Q, C, dtype=BF16, device=x.device)
    N.call('mpr_dwconv_fwd', x, w.detach(), torch.empty(w.numel(), dtype=F32, device=x.device), y, *_dw_args(x, g))
    return y


def dwconv_dgrad(dy, w, g, x_shape):
    dx = torch.empty(x_shape, dtype=BF16, device=dy.device)
    N.call('mpr_dwconv_dgrad', dy, w.detach(), torch.empty(w.numel(), dtype=F32, device=dy.device), dx, *_dw_args(dx, g))
    return dx


def dwconv_wgrad(x, dy, g, weight):
    B, H, W, C = x.shape
    P, Q = dy.shape[1], dy.shape[2]
    ws = torch.empty(N.query('mpr_dwconv_wgrad_workspace_floats', B, P, Q, C, g.R, g.S), dtype=F32, device=x.device)
    tgt = ops.grad_target(weight)
    if tgt is not None and tgt.is_contiguous():
        N.call('mpr_dwconv_wgrad', x, dy, tgt, ws, 1, *_dw_args(x, g))
        return None
    dw = torch.empty(weight.shape, dtype=F32, device=x.device)
    N.call('mpr_dwconv_wgrad', x, dy, dw, ws, 0, *_dw_args(x, g))
    return dw


class ConvBnFn(torch.autograd.Function):
    """conv (kind: 'pw' implicit-GEMM | 'dw' depthwise | 'stem' few-input-channel direct) -> BatchNorm (+ residual)
    -> optional SiLU fused into the BatchNorm passes (forward: inside the apply kernel; backward: the derivative is
    recomputed from the conv output inside both BatchNorm-backward passes).  Train mode: batch statistics, running buffers
    updated."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, residual, mod, kind, geom, silu=False):
        train = mod.training
        if kind == 'pw':
            wf, wd = ops.packed_weights(w, geom, train)
            xc, stats = ops.conv_fwd(x, wf, geom, train)
        elif kind == 'dw':
            wd = None
            xc, stats = dwconv_fwd(x, w, geom), None
        else:
            wd = None
            xc, stats = ops.stem_fwd(x, w, geom, train)
        st = _bn_coefs(stats, _rows(xc), mod, train, xc)
        out = ops.bn_apply(xc, st, residual, 2 if silu else 0)
        ctx.silu = silu
        ctx.train, ctx.kind, ctx.geom, ctx.st, ctx.wd = train, kind, geom, st, wd
        ctx.has_res = residual is not None
        if train:
            ctx.save_for_backward(x, xc, w, gamma, beta)
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.train:
            raise RuntimeError('backward through an eval-mode BatchNorm conv is not implemented')
        x, xc, w, gamma, beta = ctx.saved_tensors
        g, kind = ctx.geom, ctx.kind
        dout = dout.contiguous()
        dxc, dgamma, dbeta, _ = ops.bn_bwd(dout, None, xc, gamma, ctx.st, MASK_SILU if ctx.silu else ops.MASK_NONE, beta=beta)
        dx = None
        if kind == 'pw':
            dw = ops.conv_wgrad(x, dxc, g, w)
            if ctx.needs_input_grad[0]:
                dx = ops.conv_dgrad(dxc, ctx.wd, g, x.shape)
        elif kind == 'dw':
            dw = dwconv_wgrad(x, dxc, g, w)
            dx = dwconv_dgrad(dxc, w, g, x.shape)
        else:
            dw = ops.stem_wgrad(x, dxc, g, w)
        return dx, dw, dgamma, dbeta, (dout if ctx.has_res else None), None, None, None, None


class SiLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        C = x.shape[-1]
        y = torch.empty_like(x)
        N.call('mpr_tf_bias_act_fwd', x, None, _SILU, 0.0, 0, y, x.numel() // C, C)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        C = x.shape[-1]
        dx = torch.empty_like(x)
        N.call('mpr_tf_ew_bwd', 1, dy.contiguous(), x, None, _SILU, 0.0, 0, dx, None, None, x.numel() // C, C)
        return dx


def _act32(x, bias, act):
    y = torch.empty_like(x)
    N.call('mpr_bias_act_fwd', x, bias.detach(), act, 0.0, 0, y, None, x.numel(), x.shape[-1])
    return y


def _act32_bwd(dy, x, bias, act):
    dx = torch.empty_like(x)
    N.call('mpr_bias_act_bwd', dy, x, bias.detach(), act, 0.0, None, dx, x.numel(), x.shape[-1])
    return dx


class SqueezeExciteFn(torch.autograd.Function):
    """y = x * sigmoid(W2 silu(W1 mean_hw(x) + b1) + b2)   (timm SqueezeExcite: 1x1 convs on the pooled map)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        B, H, W, C = x.shape
        rd = w1.shape[0]
        pooled = torch.empty(B, C, dtype=F32, device=x.device)
        N.call('mpr_se_pool', x, pooled, B, H * W, C)                                 # squeeze: mean over pixels
        ctx.fused = rd <= N.query('mpr_se_mlp_max_rd')
        if ctx.fused:                                                                  # the bottleneck in one launch
            z1 = torch.empty(B, rd, dtype=F32, device=x.device)
            r = torch.empty_like(z1)
            gate = torch.empty(B, C, dtype=F32, device=x.device)
            z2 = gate
            N.call('mpr_se_mlp_fwd', pooled, w1.detach(), b1.detach(), w2.detach(), b2.detach(), z1, r, gate, B, C, rd)
        else:
            z1 = ops.gemm(pooled, w1.detach().view(rd, C), trans_b=True)              # pre-activation (bias added in act)
            r = _act32(z1, b1, _SILU)
            z2 = ops.gemm(r, w2.detach().view(C, rd), trans_b=True)
            gate = _act32(z2, b2, _SIGMOID)
        y = torch.empty_like(x)
        N.call('mpr_se_scale', x, gate, y, B, H * W, C)
        ctx.save_for_backward(x, pooled, z1, r, z2, gate, w1, b1, w2, b2)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, pooled, z1, r, z2, gate, w1, b1, w2, b2 = ctx.saved_tensors
        B, H, W, C = x.shape
        rd = w1.shape[0]
        dev = x.device
        dy = dy.contiguous()
        dgate = torch.empty(B, C, dtype=F32, device=dev)
        N.call('mpr_se_dgate', x, dy, dgate, B, H * W, C)
        dx = torch.empty_like(x)
        if ctx.fused:
            # one buffer for the four parameter gradients (accumulated by atomics: zeroed in one launch) and the scratch
            n1, n2 = rd * C, C * rd
            buf = torch.zeros(n1 + rd + n2 + C, dtype=F32, device=dev)
            dw1, db1, dw2, db2 = buf[:n1], buf[n1:n1 + rd], buf[n1 + rd:n1 + rd + n2], buf[n1 + rd + n2:]
            scr = torch.empty(2 * B * C + B * rd, dtype=F32, device=dev)
            dz2, dpooled, dz1 = scr[:B * C], scr[B * C:2 * B * C], scr[2 * B * C:]
            N.call('mpr_se_mlp_bwd', dgate, gate, z1, r, pooled, w1.detach(), w2.detach(), dz2, dz1, dpooled, dw1, db1, dw2, db2,
                   1.0 / (H * W), B, C, rd)
            # direct path dy * gate and the gate path through the mean (a per-image constant) in one pass
            N.call('mpr_se_scale_add', dy, gate, dpooled, dx, B, H * W, C)
            return dx, dw1.view(w1.shape), db1, dw2.view(w2.shape), db2
        N.call('mpr_se_scale', dy, gate, dx, B, H * W, C)                             # direct path: dy * gate
        dz2 = _act32_bwd(dgate, z2, b2, _SIGMOID)
        ones = torch.ones(B, 1, dtype=F32, device=dev)
        dw2 = ops.gemm(dz2, r, trans_a=True).view(w2.shape)
        db2 = ops.gemm(ones, dz2, trans_a=True).reshape(-1)
        dr = ops.gemm(dz2, w2.detach().view(C, rd))
        dz1 = _act32_bwd(dr, z1, b1, _SILU)
        dw1 = ops.gemm(dz1, pooled, trans_a=True).view(w1.shape)
        db1 = ops.gemm(ones, dz1, trans_a=True).reshape(-1)
        dpooled = ops.gemm(dz1, w1.detach().view(rd, C))
        dx_pool = ops.global_pool_bwd(dpooled, None, x.shape, 'avg')                  # gate path through the pooling
        dx = _add_bf16(dx, dx_pool)
        return dx, dw1, db1, dw2, db2


def _add_bf16(a, b):
    """a + b on bf16 maps: BatchNorm-apply with unit scale / zero shift and `b` as the residual (no new kernel)."""
    C = a.shape[-1]
    key = (C, a.device)
    cached = _add_bf16.cache.get(key)
    if cached is None:
        cached = _add_bf16.cache[key] = (torch.ones(C, dtype=F32, device=a.device), torch.zeros(C, dtype=F32, device=a.device))
    y = torch.empty_like(a)
    N.call('mpr_bn_apply', a, cached[0], cached[1], b, 0, y, a.numel() // C, C)
    return y


_add_bf16.cache = {}


# ---------------------------------------------------------------------------------------------- modules (timm names)
class _SE(nn.Module):
    def __init__(self, chs, rd):
        super().__init__()
        self.conv_reduce = nn.Conv2d(chs, rd, 1)
        self.conv_expand = nn.Conv2d(rd, chs, 1)

    def forward(self, x):
        return SqueezeExciteFn.apply(x, self.conv_reduce.weight, self.conv_reduce.bias, self.conv_expand.weight,
                                     self.conv_expand.bias)


def _conv_bn(x, conv, bn, kind, geom, residual=None, silu=False):
    return ConvBnFn.apply(x, conv.weight, bn.weight, bn.bias, residual, bn, kind, geom, silu)


class DepthwiseSeparableConv(nn.Module):
    """timm DepthwiseSeparableConv: dw kxk - BN - SiLU - SE - pw 1x1 - BN (+ skip)."""

    def __init__(self, cin, cout, k, stride, rd):
        super().__init__()
        self.conv_dw = nn.Conv2d(cin, cin, k, stride, k // 2, groups=cin, bias=False)
        self.bn1 = BatchNormParams(cin)
        self.se = _SE(cin, rd)
        self.conv_pw = nn.Conv2d(cin, cout, 1, bias=False)
        self.bn2 = BatchNormParams(cout)
        self.has_skip = stride == 1 and cin == cout
        self.g_dw = ConvGeom((cin, 1, k, k), stride, k // 2)
        self.g_pw = ConvGeom((cout, cin, 1, 1), 1, 0)

    def forward(self, x):
        h = _conv_bn(x, self.conv_dw, self.bn1, 'dw', self.g_dw, silu=True)
        h = self.se(h)
        return _conv_bn(h, self.conv_pw, self.bn2, 'pw', self.g_pw, x if self.has_skip else None)


class InvertedResidual(nn.Module):
    """timm InvertedResidual (MBConv): pw expand - BN - SiLU - dw kxk - BN - SiLU - SE - pw project - BN (+ skip)."""

    def __init__(self, cin, cout, k, stride, exp, rd):
        super().__init__()
        mid = cin * exp
        self.conv_pw = nn.Conv2d(cin, mid, 1, bias=False)
        self.bn1 = BatchNormParams(mid)
        self.conv_dw = nn.Conv2d(mid, mid, k, stride, k // 2, groups=mid, bias=False)
        self.bn2 = BatchNormParams(mid)
        self.se = _SE(mid, rd)
        self.conv_pwl = nn.Conv2d(mid, cout, 1, bias=False)
        self.bn3 = BatchNormParams(cout)
        self.has_skip = stride == 1 and cin == cout
        self.g_pw = ConvGeom((mid, cin, 1, 1), 1, 0)
        self.g_dw = ConvGeom((mid, 1, k, k), stride, k // 2)
        self.g_pwl = ConvGeom((cout, mid, 1, 1), 1, 0)

    def forward(self, x):
        h = _conv_bn(x, self.conv_pw, self.bn1, 'pw', self.g_pw, silu=True)
        h = _conv_bn(h, self.conv_dw, self.bn2, 'dw', self.g_dw, silu=True)
        h = self.se(h)
        return _conv_bn(h, self.conv_pwl, self.bn3, 'pw', self.g_pwl, x if self.has_skip else None)


# (repeats, kernel, stride, expansion, out channels) per stage -- efficientnet_b0
_B0 = ((1, 3, 1, 1, 16), (2, 3, 2, 6, 24), (2, 5, 2, 6, 40), (3, 3, 2, 6, 80), (3, 5, 1, 6, 112), (4, 5, 2, 6, 192),
       (1, 3, 1, 6, 320))


class EfficientNetBackbone(nn.Module):
    def __init__(self, in_chans: int = 1, arch=_B0, stem_chs: int = 32, num_features: int = 1280):
        super().__init__()
        self.conv_stem = nn.Conv2d(in_chans, stem_chs, 3, 2, 1, bias=False)
        self.bn1 = BatchNormParams(stem_chs)
        self.g_stem = ConvGeom((stem_chs, in_chans, 3, 3), 2, 1)
        stages, cin = [], stem_chs
        for reps, k, stride, exp, cout in arch:
            blocks = []
            for i in range(reps):
                s = stride if i == 0 else 1
                rd = max(1, int(round(cin * 0.25)))
                blocks.append(DepthwiseSeparableConv(cin, cout, k, s, rd) if exp == 1
                              else InvertedResidual(cin, cout, k, s, exp, rd))
                cin = cout
            stages.append(nn.Sequential(*blocks))
        self.blocks = nn.Sequential(*stages)
        self.conv_head = nn.Conv2d(cin, num_features, 1, bias=False)
        self.bn2 = BatchNormParams(num_features)
        self.g_head = ConvGeom((num_features, cin, 1, 1), 1, 0)
        self.num_features, self.in_chans = num_features, in_chans
        for m in self.modules():                      # timm efficientnet_init_weights (goog): normal(0, sqrt(2 / fan_out))
            if isinstance(m, nn.Conv2d):
                fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
                nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / fan_out))
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward_features(self, image: Tensor) -> Tensor:
        """fp32 [B, in_chans, H, W] -> channels-last bf16 [B, H/32, W/32, 1280]."""
        B, C, H, W = image.shape
        x = image.float()
        x = (x.reshape(B, H, W, 1) if C == 1 else x.permute(0, 2, 3, 1)).contiguous()
        h = _conv_bn(x, self.conv_stem, self.bn1, 'stem', self.g_stem, silu=True)
        for stage in self.blocks:
            for blk in stage:
                h = blk(h)
        return _conv_bn(h, self.conv_head, self.bn2, 'pw', self.g_head, silu=True)
