"""fp32 PARITY mode of the conv stacks (`precision: 32`): the same modules, parameters and state_dict, but fp32 feature
maps and the kernels of csrc/conv_f32.hip -- exact-fp32 MFMA products, double statistics, no atomics, everything on the
caller's stream (bitwise reproducible from run to run).  One autograd Function per reference op, nothing fused: this mode
exists to be compared with the reference's fp32 CPU arithmetic at 1e-4 / 1e-3 (tests/test_f32_path_gpu.py), the bf16
Functions of layers.py are the throughput path.

Reference ops: nn.Conv1d/2d, nn.BatchNorm1d/2d (train and eval), nn.ReLU, nn.MaxPool1d/2d(3, 2, 1), the residual add of
timm's BasicBlock / _BasicBlock (src/profile_encoder.py:111-148), global average / max pooling (src/image_encoder.py:24,
src/profile_encoder.py:232-236).
"""
import torch

from . import _native as N
from . import ops
from .ops import F32, _geom, _kcrs_strides, _like_spatial

_PRECISION = ['bf16']


def set_conv_precision(precision):
    """Trainer / card `precision` -> arithmetic of the conv stacks: '32' / '32-true' / 'fp32' select this module's fp32
    kernels; None and the mixed modes keep the bf16-storage kernels (what the reference's cards train with:
    model_cards/example_multi.yaml:37 `16-mixed`).  Returns the previous mode."""
    old = _PRECISION[0]
    name = str(precision).lower() if precision is not None else ''
    _PRECISION[0] = 'fp32' if name in ('32', '32-true', 'fp32', 'float32', '64', '64-true') else 'bf16'
    return old


def conv_f32():
    return _PRECISION[0] == 'fp32'


# ------------------------------------------------------------------------------------------------ convolution
class ConvF32Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, g):
        B, H, W, C = _geom(x)
        if C != g.C:
            raise N.NativeLibraryError(f'conv (fp32): input has {C} channels, weight expects {g.C}')
        P, Q = g.out_hw(H, W)
        y = torch.empty(_like_spatial(x, B, P, Q, g.K), dtype=F32, device=x.device)
        N.call('mpr_f32_conv_fwd', x, weight.detach(), *_kcrs_strides(weight), y, B, H, W, C, g.K, *g.tail)
        ctx.save_for_backward(x, weight)
        ctx.g = g
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        g = ctx.g
        dy = dy.contiguous()
        B, H, W, C = _geom(x)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            N.call('mpr_f32_conv_dgrad', dy, weight.detach(), *_kcrs_strides(weight), dx, None, B, H, W, C, g.K, *g.tail)
        dw = None
        if ctx.needs_input_grad[1]:
            tgt = ops.grad_target(weight)
            out = tgt if tgt is not None else torch.empty_like(weight)          # (keeps the filter's strides)
            if _kcrs_strides(out) != _kcrs_strides(weight):
                raise N.NativeLibraryError('conv (fp32): gradient memory does not have the filter\'s layout')
            need = N.query('mpr_f32_conv_wgrad_scratch_floats', B, H, W, C, g.K, *g.tail)
            scratch = torch.empty(need, dtype=F32, device=x.device) if need else None
            N.call('mpr_f32_conv_wgrad', x, dy, out, *_kcrs_strides(weight), int(tgt is not None), scratch, need, B, H, W, C,
                   g.K, *g.tail)
            dw = None if tgt is not None else out
        return dx, dw, None


# ------------------------------------------------------------------------------------------------ BatchNorm (+ add, ReLU)
class BNActF32Fn(torch.autograd.Function):
    """y = act(BN(x) (+ residual)); train mode uses and updates batch statistics exactly as nn.BatchNorm (biased variance
    for the normalisation, unbiased for running_var, momentum / eps of the module), eval mode the running statistics."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, relu, bn):
        C = x.shape[-1]
        rows = x.numel() // C
        dev = x.device
        scale, shift = torch.empty(C, dtype=F32, device=dev), torch.empty(C, dtype=F32, device=dev)
        train = bn.training
        if train:
            nparts = N.query('mpr_f32_bn_parts', rows, C)
            part = torch.empty(nparts, 2, C, dtype=torch.float64, device=dev)
            mean, invstd = torch.empty(C, dtype=F32, device=dev), torch.empty(C, dtype=F32, device=dev)
            N.call('mpr_f32_bn_stats', x, part, rows, C)
            N.call('mpr_f32_bn_finalize', part, nparts, rows, gamma.detach(), beta.detach(), bn.running_mean, bn.running_var,
                   float(bn.momentum), float(bn.eps), scale, shift, mean, invstd, C)
            bn.count_batch()
        else:
            N.call('mpr_bn_eval_coefs', gamma.detach(), beta.detach(), bn.running_mean, bn.running_var, float(bn.eps), scale,
                   shift, C)
            mean = bn.running_mean
            invstd = torch.empty(C, dtype=F32, device=dev)
            N.call('mpr_bn_eval_invstd', bn.running_var, float(bn.eps), invstd, C)
        y = torch.empty_like(x)
        N.call('mpr_f32_bn_apply', x, scale, shift, residual, int(relu), y, rows, C)
        ctx.save_for_backward(x, y if relu else None, gamma, beta, mean, invstd)
        ctx.cfg = (train, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, beta, mean, invstd = ctx.saved_tensors
        train, has_res = ctx.cfg
        dy = dy.contiguous()
        C = x.shape[-1]
        rows = x.numel() // C
        dev = x.device
        nparts = N.query('mpr_f32_bn_parts', rows, C)
        part = torch.empty(nparts, 2, C, dtype=torch.float64, device=dev)
        N.call('mpr_f32_bn_bwd_reduce', dy, y, x, mean, invstd, part, rows, C)
        tg, tb = ops.grad_target(gamma), ops.grad_target(beta)
        own = tg is not None and tb is not None
        dgamma = tg if own else torch.empty(C, dtype=F32, device=dev)
        dbeta = tb if own else torch.empty(C, dtype=F32, device=dev)
        coef = torch.empty(3, C, dtype=F32, device=dev)
        N.call('mpr_f32_bn_bwd_finalize', part, nparts, rows if train else 0, gamma.detach(), mean, invstd, dgamma, dbeta,
               int(own), coef, C)
        dx = torch.empty_like(x)
        # the masked gradient is also the residual branch's gradient
        dz = torch.empty_like(x) if (has_res and y is not None) else None
        N.call('mpr_f32_bn_bwd_apply', dy, y, x, coef, dx, dz, rows, C)
        dres = (dz if dz is not None else dy) if has_res else None
        return dx, (None if own else dgamma), (None if own else dbeta), dres, None, None


# ------------------------------------------------------------------------------------------------ pooling
class MaxPoolF32Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, s, p):
        B, H, W, C, RH, RW, SH, SW, PH, PW = ops._pool_geom(x, k, s, p)
        P, Q = (H + 2 * PH - RH) // SH + 1, (W + 2 * PW - RW) // SW + 1
        y = torch.empty(_like_spatial(x, B, P, Q, C), dtype=F32, device=x.device)
        idx = torch.empty(y.shape, dtype=torch.int32, device=x.device)
        N.call('mpr_f32_maxpool_fwd', x, y, idx, B, H, W, C, RH, RW, SH, SW, PH, PW)
        ctx.save_for_backward(idx)
        ctx.cfg = (x.shape, (B, H, W, C, RH, RW, SH, SW, PH, PW))
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        shape, geo = ctx.cfg
        dx = torch.empty(shape, dtype=F32, device=dy.device)
        N.call('mpr_f32_maxpool_bwd', dy.contiguous(), idx, dx, *geo)
        return dx, None, None, None


def global_pool_fwd(x, mode):
    B, C = x.shape[0], x.shape[-1]
    L = x.numel() // (B * C)
    y = torch.empty(B, C, dtype=F32, device=x.device)
    idx = torch.empty(B, C, dtype=torch.int32, device=x.device) if mode != 'avg' else None
    N.call('mpr_f32_global_pool_fwd', x, y, idx, B, L, C, 0 if mode == 'avg' else 1)
    return y, idx


def global_pool_bwd(dy, idx, x_shape, mode):
    dx = torch.empty(x_shape, dtype=F32, device=dy.device)
    B, C = x_shape[0], x_shape[-1]
    L = dx.numel() // (B * C)
    N.call('mpr_f32_global_pool_bwd', dy, idx, dx, B, L, C, 0 if mode == 'avg' else 1)
    return dx


# ------------------------------------------------------------------------------------------------ module-level forwards
def stem(mod, x):
    """conv1 -> bn1 -> ReLU -> MaxPool(3, 2, 1) of a ResNetBackbone / ProfileCNN (x: fp32 channels-last)."""
    y = ConvF32Fn.apply(x, mod.conv1.weight, mod.geom)
    y = BNActF32Fn.apply(y, mod.bn1.weight, mod.bn1.bias, None, True, mod.bn1)
    return MaxPoolF32Fn.apply(y, 3, 2, 1)


def basic_block(blk, x):
    """timm BasicBlock / _BasicBlock.forward (src/profile_encoder.py:132-148)."""
    out = ConvF32Fn.apply(x, blk.conv1.weight, blk.geom1)
    out = BNActF32Fn.apply(out, blk.bn1.weight, blk.bn1.bias, None, True, blk.bn1)
    out = ConvF32Fn.apply(out, blk.conv2.weight, blk.geom2)
    identity = x
    if blk.downsample is not None:
        identity = ConvF32Fn.apply(x, blk.downsample[0].weight, blk.geomd)
        identity = BNActF32Fn.apply(identity, blk.downsample[1].weight, blk.downsample[1].bias, None, False, blk.downsample[1])
    return BNActF32Fn.apply(out, blk.bn2.weight, blk.bn2.bias, identity, True, blk.bn2)
