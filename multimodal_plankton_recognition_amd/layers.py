"""Fused building blocks of both ResNets (image: 2-D, profile: 1-D) as autograd Functions.

Each Function's forward/backward is a hand-ordered sequence of C-ABI kernel launches
(``ops.py``); autograd only chains stem -> blocks -> pool/tail -> projection -> loss, so no torch
arithmetic kernel runs between them (the skip-connection gradient add is fused into the dgrad
epilogue).  Parameter containers are ordinary ``torch.nn`` modules so that ``state_dict()`` keys,
shapes and default initialisation match the reference modules
(src/profile_encoder.py:111-148,151-240; timm ResNet BasicBlock).
"""
import random

import torch
from torch import nn

from . import ops
from .ops import ConvGeom, MASK_NONE, MASK_RECOMPUTE, MASK_Y

_seed_rng = None


def next_seed():
    """Dropout seeds: a Python stream derived from torch's global seed (no device sync)."""
    global _seed_rng
    if _seed_rng is None:
        _seed_rng = random.Random(torch.initial_seed())
    return _seed_rng.getrandbits(32)


class BatchNormParams(nn.Module):
    """Parameter/buffer container with nn.BatchNorm{1,2}d's state_dict layout and defaults
    (momentum 0.1, eps 1e-5).  ``num_batches_tracked`` is counted on the host and flushed into the
    buffer whenever the state_dict is taken (keeps 40 one-element kernels out of every step)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))
        self._pending = 0
        self._register_state_dict_hook(BatchNormParams._flush_hook)

    @staticmethod
    def _flush_hook(module, state_dict, prefix, local_metadata):
        if module._pending:
            module.num_batches_tracked += module._pending
            module._pending = 0
            state_dict[prefix + 'num_batches_tracked'] = module.num_batches_tracked

    def count_batch(self):
        self._pending += 1


def _bn_coefs(stats, count, bn, train, x=None, defer=False, want_bwd=False):
    st = ops.bn_coefs(stats, count, bn, train, x, defer=defer, want_bwd=want_bwd)
    if train:
        bn.count_batch()
    return st


def _rows(t):
    return t.numel() // t.shape[-1]


# ================================================================================================ stem
class StemFn(torch.autograd.Function):
    """conv(k, stride 2, few input channels) -> BN -> ReLU -> MaxPool(3, 2, 1), one Function.
    timm ResNet conv1/bn1/act1/maxpool; ProfileCNN.forward_features src/profile_encoder.py:215-220."""

    @staticmethod
    def forward(ctx, x, conv_w, bn_w, bn_b, mod):
        g = mod.geom
        train = mod.training
        ctx.train = train
        ctx.fused = ops.stemf_ok(x, g)
        if ctx.fused:
            # ResNet stem, 1 channel, 64 filters: fused + recomputed (csrc/stem_fused.hip) -- the full-resolution conv
            # output is never stored; eval-mode BatchNorm (running statistics) has a backward pass here as well
            want_bwd = any(ctx.needs_input_grad[1:4])
            pooled, st, saved = ops.stemf_forward(x, conv_w, mod.bn1, train, want_bwd)
            if train:
                mod.bn1.count_batch()
            if want_bwd:
                ctx.save_for_backward(*saved, conv_w, bn_w, bn_b)
                ctx.st, ctx.bn = st, mod.bn1
            return pooled
        # ResNet stem (1 channel, 7x7/2, pad 3, even image): space-to-depth -> MFMA implicit GEMM
        s2d = (x.dim() == 4 and g.C == 1 and g.R == 7 and g.S == 7 and g.sh == 2 and g.ph == 3
               and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0 and g.K % 8 == 0)
        if s2d:
            x, g2, wf2 = ops.stem_s2d_operands(x, conv_w)
            y, stats = ops.conv_fwd(x, wf2, g2, train)
            g = g2
        else:
            y, stats = ops.stem_fwd(x, conv_w, g, train)
        st = _bn_coefs(stats, _rows(y), mod.bn1, train)
        pooled, idx = ops.bn_relu_maxpool_fwd(y, st)
        if train:
            ctx.save_for_backward(x, y, idx, conv_w, bn_w, bn_b)
            ctx.st, ctx.g, ctx.s2d = st, g, s2d
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        if ctx.fused:
            xb, wp, idx, conv_w, bn_w, bn_b = ctx.saved_tensors
            dw, dgamma, dbeta = ops.stemf_backward(dpooled.contiguous(), xb, wp, idx, conv_w, bn_w, bn_b, ctx.st,
                                                   ctx.bn, ctx.train)
            return None, dw, dgamma, dbeta, None
        if not ctx.train:
            raise RuntimeError('backward through an eval-mode BatchNorm stem is not implemented')
        x, y, idx, conv_w, bn_w, bn_b = ctx.saved_tensors
        # parameter gradients go straight into the optimizer's flat buffer when it owns one (ops.grad_target):
        # the corresponding return values are then None
        dx, dgamma, dbeta = ops.pool_bn_bwd(dpooled.contiguous(), idx, y, bn_w, ctx.st, beta=bn_b)
        if ctx.s2d:
            dw = ops.stem_s2d_wgrad(x, dx, ctx.g, conv_w)
        else:
            dw = ops.stem_wgrad(x, dx, ctx.g, conv_w)
        return None, dw, dgamma, dbeta, None


# ================================================================================================ basic block
class BasicBlockFn(torch.autograd.Function):
    """conv3-BN-ReLU-conv3-BN (+ 1x1-conv-BN shortcut) + add + ReLU.
    src/profile_encoder.py:132-148 (1-D) and timm BasicBlock (2-D) -- same kernels, H == 1 for 1-D."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, w2, g2, b2, wd, gd, bd, mod):
        train = mod.training
        c1, c2, cd = mod.geom1, mod.geom2, mod.geomd
        # a backward pass will follow when anything differentiable comes in -- also in eval mode, where BatchNorm is the
        # affine map of its running statistics (ops.bn_coefs keeps mean / invstd for it when gradients are enabled)
        need_bwd = any(ctx.needs_input_grad)
        wf1, wd1 = ops.packed_weights(w1, c1, need_bwd)
        wf2, wd2 = ops.packed_weights(w2, c2, need_bwd)
        x1, s1 = ops.conv_fwd(x, wf1, c1, train)
        st1 = _bn_coefs(s1, _rows(x1), mod.bn1, train, x1, defer=True, want_bwd=need_bwd)   # finalized inside the bn_apply that follows
        a1 = ops.bn_apply(x1, st1, None, True)
        x2, s2 = ops.conv_fwd(a1, wf2, c2, train)
        st2 = _bn_coefs(s2, _rows(x2), mod.bn2, train, x2, defer=True, want_bwd=need_bwd)
        if wd is not None:
            wfd, wdd = ops.packed_weights(wd, cd, need_bwd)
            xd, sd = ops.conv_fwd(x, wfd, cd, train)
            std = _bn_coefs(sd, _rows(xd), mod.downsample[1], train, xd, defer=True, want_bwd=need_bwd)
            # both BatchNorms, the add and the ReLU in one pass: the normalised shortcut map is never stored
            out = ops.bn_apply_dual(x2, st2, xd, std, True)
            if out is None:
                out = ops.bn_apply(x2, st2, ops.bn_apply(xd, std, None, False), True)
        else:
            xd = std = wdd = None
            out = ops.bn_apply(x2, st2, x, True)
        if need_bwd:
            ctx.save_for_backward(x, x1, a1, x2, out, xd, w1, w2, wd, g1, g2, gd, b1, b2, bd)
            ctx.misc = (st1, st2, std, c1, c2, cd, wd1, wd2, wdd)
            ctx.chain = mod.chain
            if mod.chain is not None and mod.feeds_block:
                # `out` feeds exactly one consumer, the next BasicBlock: its backward may fuse this block's bn2 reduction
                # into the data gradient that produces d(out) (ops.conv_dgrad_bn)
                mod.chain.note_bn2(out, x2, st2)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, x1, a1, x2, out, xd, w1, w2, wd, g1, g2, gd, b1, b2, bd = ctx.saved_tensors
        st1, st2, std, c1, c2, cd, wd1, wd2, wdd = ctx.misc
        dout = dout.contiguous()
        chain = ctx.chain
        if chain is not None:
            chain.take_note(out)          # (a note nobody used must not outlive this pass)
        # out = relu(bn2(x2) + identity):  dz = dout * (out > 0) feeds bn2 AND the shortcut
        sums2 = chain.take_sums(dout) if chain is not None else None
        if sums2 is not None:
            # the consumer's data gradient already masked d(out) and left bn2's backward sums: one pass instead of two
            dz = dout
            dx2, dg2, db2 = ops.bn_bwd_from_sums(dz, x2, g2, st2, sums2, beta=b2)
        else:
            dx2, dg2, db2, dz = ops.bn_bwd(dout, out, x2, g2, st2, MASK_Y, want_dz=True, beta=b2)
        dw2 = ops.conv_wgrad(a1, dx2, c2, w2)
        # a1 = bf16(relu(x1*scale+shift)) has no residual: its ReLU mask is recomputed from x1 (bit-identical to
        # a1 > 0) -- inside the data gradient's epilogue together with bn1's backward sums where the kernel allows it
        fused1 = ops.conv_dgrad_bn(dx2, wd2, c2, a1.shape, x1, st1, 2)
        if fused1 is not None:
            dx1, dg1, db1 = ops.bn_bwd_from_sums(fused1[0], x1, g1, st1, fused1[1], beta=b1)
        else:
            da1 = ops.conv_dgrad(dx2, wd2, c2, a1.shape)
            dx1, dg1, db1, _ = ops.bn_bwd(da1, None, x1, g1, st1, MASK_RECOMPUTE, beta=b1)
        dw1 = ops.conv_wgrad(x, dx1, c1, w1)
        if wd is not None:
            dxd, dgd, dbd, _ = ops.bn_bwd(dz, None, xd, gd, std, MASK_NONE, beta=bd)
            dwd = ops.conv_wgrad(x, dxd, cd, wd)
            # the shortcut conv's data gradient lives on the even pixels only: half-resolution GEMM, added by the
            # parity-class kernel of conv1's stride-2 data gradient; x is the previous block's output: if that block left
            # a note, its ReLU mask and bn2's backward sums are fused into the same epilogue
            note = chain.take_note(x) if chain is not None else None
            res = ops.conv_dgrad_shortcut(dx1, wd1, c1, x.shape, dxd, wdd, cd, note=note, mask_y=x)
            if res is not None:
                if res[1] is not None:
                    chain.offer_sums(res[0], res[1])
                return res[0], dw1, dg1, db1, dw2, dg2, db2, dwd, dgd, dbd, None
            skip = ops.conv_dgrad(dxd, wdd, cd, x.shape)
        else:
            dgd = dbd = dwd = None
            skip = dz
            note = chain.take_note(x) if chain is not None else None
        # d(x) = conv1's data gradient + the skip-connection gradient (fused in the epilogue).  x is the previous block's
        # output: if that block left a note, its ReLU mask and bn2 sums are fused in as well
        fused0 = ops.conv_dgrad_bn(dx1, wd1, c1, x.shape, note[0], note[1], 1, mask_y=x, add=skip) if note else None
        if fused0 is not None:
            dx = fused0[0]
            chain.offer_sums(dx, fused0[1])
        else:
            dx = ops.conv_dgrad(dx1, wd1, c1, x.shape, add=skip)
        return dx, dw1, dg1, db1, dw2, dg2, db2, dwd, dgd, dbd, None


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, dims, in_channels, out_channels, stride, downsample):
        super().__init__()
        conv = nn.Conv2d if dims == 2 else nn.Conv1d
        self.conv1 = conv(in_channels, out_channels, 3, stride, 1, bias=False)
        self.bn1 = BatchNormParams(out_channels)
        self.conv2 = conv(out_channels, out_channels, 3, 1, 1, bias=False)
        self.bn2 = BatchNormParams(out_channels)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(conv(in_channels, out_channels, 1, stride, 0, bias=False),
                                            BatchNormParams(out_channels))
        self.stride = stride
        # set by the backbone (layers.chain_blocks): its blocks share a hand-off table, and every block but the last feeds
        # the next BasicBlock and nothing else
        self.chain, self.feeds_block = None, False
        self.geom1 = ConvGeom(tuple(self.conv1.weight.shape), stride, 1)
        self.geom2 = ConvGeom(tuple(self.conv2.weight.shape), 1, 1)
        self.geomd = ConvGeom(tuple(self.downsample[0].weight.shape), stride, 0) if downsample else None

    def forward(self, x):
        ds = self.downsample
        if x.dtype == torch.float32:          # fp32 parity mode (`precision: 32`): unfused exact-fp32 kernels, layers_f32.py
            from . import layers_f32
            return layers_f32.basic_block(self, x)
        return BasicBlockFn.apply(x, self.conv1.weight, self.bn1.weight, self.bn1.bias, self.conv2.weight,
                                  self.bn2.weight, self.bn2.bias,
                                  ds[0].weight if ds is not None else None,
                                  ds[1].weight if ds is not None else None,
                                  ds[1].bias if ds is not None else None, self)


def chain_blocks(blocks):
    """Declare `blocks` a strict chain (each block's output is consumed by the next block and by nothing else): their
    backward passes may then hand BatchNorm-backward sums to each other (ops.BlockChain).  Returns the shared table; the
    owner clears it at the start of every forward pass."""
    chain = ops.BlockChain()
    blocks = list(blocks)
    for i, blk in enumerate(blocks):
        blk.chain, blk.feeds_block = chain, i + 1 < len(blocks)
    return chain


# ================================================================================================ pool + tail
class PoolTailFn(torch.autograd.Function):
    """global pool (avg | max) -> concat(metadata / denom) -> dropout.
    src/image_encoder.py:24-29 (timm global avg pool) and src/profile_encoder.py:232-240."""

    @staticmethod
    def forward(ctx, fmap, meta, mode, denom, p_drop):
        f32 = fmap.dtype == torch.float32     # fp32 parity mode: fp32 feature map (layers_f32.py)
        if f32:
            from . import layers_f32
            feat, idx = layers_f32.global_pool_fwd(fmap, mode)
        else:
            feat, idx = ops.global_pool_fwd(fmap, mode)
        out, mask = ops.tail_fwd(feat, meta, denom, p_drop, next_seed() if p_drop > 0 else 0)
        ctx.save_for_backward(idx, mask)
        ctx.cfg = (mode, p_drop, feat.shape[1], fmap.shape, f32)
        return out

    @staticmethod
    def backward(ctx, dout):
        idx, mask = ctx.saved_tensors
        mode, p_drop, Fd, shape, f32 = ctx.cfg
        dfeat = ops.tail_bwd(dout.contiguous(), mask, p_drop, Fd)
        if f32:
            from . import layers_f32
            return layers_f32.global_pool_bwd(dfeat, idx, shape, mode), None, None, None, None
        return ops.global_pool_bwd(dfeat, idx, shape, mode), None, None, None, None


class TailFn(torch.autograd.Function):
    """concat(metadata / denom) -> dropout on already-pooled fp32 features (transformer / LSTM encoders)."""

    @staticmethod
    def forward(ctx, feat, meta, denom, p_drop):
        out, mask = ops.tail_fwd(feat.contiguous(), meta, denom, p_drop, next_seed() if p_drop > 0 else 0)
        ctx.save_for_backward(mask)
        ctx.cfg = (p_drop, feat.shape[1])
        return out

    @staticmethod
    def backward(ctx, dout):
        (mask,) = ctx.saved_tensors
        p_drop, Fd = ctx.cfg
        return ops.tail_bwd(dout.contiguous(), mask, p_drop, Fd), None, None, None


# ================================================================================================ linear (exact fp32)
_ones_cache = {}


def _ones(n, device):
    key = (n, str(device))
    if key not in _ones_cache:
        _ones_cache[key] = torch.ones(n, 1, dtype=torch.float32, device=device)
    return _ones_cache[key]


class LinearFn(torch.autograd.Function):
    """y = x W^T (+ b) on the exact-fp32 MFMA GEMM.  nn.Linear at src/model.py:31-32,40-41,164,316."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return ops.gemm(x, weight, trans_b=True, bias=bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = ops.gemm(dy, weight) if ctx.needs_input_grad[0] else None
        dw = ops.linear_wgrad(dy, x, weight)
        db = ops.gemm(_ones(dy.shape[0], dy.device), dy, trans_a=True).reshape(-1) if ctx.has_bias else None
        return dx, dw, db


def linear(x, weight, bias=None):
    return LinearFn.apply(x, weight, bias)
