"""Mixed-precision ('bf16-mixed' / '16-mixed') transformer blocks on the gfx950 kernels.

The reference trains under Lightning's ``precision: 16-mixed`` (model_cards/example_multi.yaml:37): autocast runs every
``nn.Linear`` and the attention products in half precision and keeps LayerNorm / softmax / the residual stream in
fp32.  Here the same split is explicit: the residual stream is fp32 ``[B*T, d]``; LayerNorm outputs, qkv, attention
outputs and MLP activations (and their gradients) are bf16, the four linears of a block are 1x1 implicit-GEMM
convolutions on the bf16 MFMA kernels (``ops.conv_fwd / conv_dgrad / conv_wgrad``), attention is one fused kernel per
direction (``mpr_attn_fwd / mpr_attn_bwd``: scores never reach HBM) and everything in between is fused elementwise /
row-wise HIP (``csrc/transformer_bf16.hip``).  One autograd Function per encoder block: its backward is written out by
hand, parameter gradients go straight into the optimizer's flat gradient buffer when there is one.

Blocks: timm's pre-norm ViT block (behind /root/reference/src/image_encoder.py:16,24) and torch's post-norm
``nn.TransformerEncoderLayer`` (/root/reference/src/profile_encoder.py:22-30,57-68).
"""
import math

import torch

from . import _native as N
from . import ops
from .layers import next_seed

F32, BF16 = torch.float32, torch.bfloat16
_ACT = {'none': 0, 'gelu': 1, 'relu': 2}


def supported(T, head_dim):
    return bool(N.query('mpr_attn_supported', int(T), int(head_dim)))


def _check(T, d, heads):
    if d % heads or not supported(T, d // heads) or d % 8 or d > 1024:
        raise NotImplementedError(
            f'bf16-mixed transformer path: needs head size 64 with T <= 256 tokens or head size 32 with T <= 288, and d_model <= 1024 '
            f'(got T={T}, d_model={d}, heads={heads}); run this model with precision 32 (exact-fp32 kernels)')


# ---------------------------------------------------------------------------------------------- kernel wrappers
def add_ln(x, r=None, rbias=None, p=0.0, seed=0, gamma=None, beta=None, eps=1e-5, want_s=False, want32=False,
           want16=False):
    """s = x + drop(r + rbias) (x None: zeros); y = LN(s).  -> (s | None, y32 | None, y16 | None, mean | None, rstd | None)."""
    rows, D = (x if x is not None else r).shape
    dev = (x if x is not None else r).device
    s = torch.empty(rows, D, dtype=F32, device=dev) if want_s else None
    y32 = torch.empty(rows, D, dtype=F32, device=dev) if want32 else None
    y16 = torch.empty(rows, D, dtype=BF16, device=dev) if want16 else None
    mean = rstd = None
    if gamma is not None:
        mean = torch.empty(rows, dtype=F32, device=dev)
        rstd = torch.empty(rows, dtype=F32, device=dev)
    N.call('mpr_tf_add_ln_fwd', x, r, rbias.detach() if rbias is not None else None, float(p), int(seed),
           gamma.detach() if gamma is not None else None, beta.detach() if beta is not None else None, float(eps), s, y32,
           y16, mean, rstd, rows, D)
    return s, y32, y16, mean, rstd


def _param_grad(param):
    """-> (buffer the kernels accumulate into, value handed back to autograd, accumulate flag)."""
    tgt = ops.grad_target(param)
    if tgt is not None:
        return tgt, None, 1
    z = torch.zeros_like(param)
    return z, z, 0


def ln_bwd(dy16, dy32, s, gamma, beta, mean, rstd, dskip):
    rows, D = s.shape
    ds = torch.empty_like(s)
    dg, dg_ret, acc = _param_grad(gamma)
    db, db_ret, acc_b = _param_grad(beta)
    if acc != acc_b:            # one parameter owned by the optimizer, the other not: keep the kernel's flag simple
        raise N.NativeLibraryError('LayerNorm weight and bias must both (or neither) belong to the fused optimizer')
    ws = torch.empty(N.query('mpr_tf_ln_bwd_workspace_floats', rows, D), dtype=F32, device=s.device)
    N.call('mpr_tf_ln_bwd', dy16, dy32, s, gamma.detach(), mean, rstd, dskip, ds, dg, db, ws, acc, rows, D)
    return ds, dg_ret, db_ret


def bias_act(x16, bias, act, p, seed):
    rows, D = x16.shape
    y = torch.empty_like(x16)
    N.call('mpr_tf_bias_act_fwd', x16, bias.detach() if bias is not None else None, act, float(p), int(seed), y, rows, D)
    return y


def ew_bwd(mode, dy, rows, D, x16=None, bias=None, act=0, p=0.0, seed=0):
    """mode 0: bias gradient of a bf16 tensor; 1: activation backward (bf16); 2: fp32 -> bf16 branch gradient.
    -> (dx16 | None, value handed to autograd for the bias gradient)."""
    if mode == 0 and bias is None:
        return None, None
    dx = torch.empty(rows, D, dtype=BF16, device=dy.device) if mode else None
    db = db_ret = ws = None
    if bias is not None:
        db, db_ret, _ = _param_grad(bias)
        ws = torch.empty(N.query('mpr_tf_ew_bwd_workspace_floats', rows, D), dtype=F32, device=dy.device)
    N.call('mpr_tf_ew_bwd', mode, dy, x16, bias.detach() if (bias is not None and mode == 1) else None, act, float(p),
           int(seed), dx, db, ws, rows, D)
    return dx, db_ret


def cast_bf16(x):
    y = torch.empty(x.shape, dtype=BF16, device=x.device)
    N.call('mpr_tf_cast', x.contiguous(), y, x.numel(), 1)
    return y


class _Lin:
    """One nn.Linear weight as a 1x1 convolution over [B, T, C] bf16 tokens."""

    def __init__(self, weight, B, T):
        self.w = weight
        self.g = ops.ConvGeom(tuple(weight.shape))
        self.B, self.T = B, T

    def fwd(self, x16):
        wf, _ = ops.packed_weights(self.w, self.g)
        y, _ = ops.conv_fwd(x16.view(self.B, self.T, self.g.C), wf, self.g, False)
        return y.view(self.B * self.T, self.g.K)

    def bwd(self, x16, dy16, need_dx=True):
        """-> (dx16 | None, weight gradient for autograd (None when it went into the optimizer's buffer))."""
        dy3 = dy16.view(self.B, self.T, self.g.K)
        x3 = x16.view(self.B, self.T, self.g.C)
        dx = None
        if need_dx:
            _, wd = ops.packed_weights(self.w, self.g)
            dx = ops.conv_dgrad(dy3, wd, self.g, x3.shape).view(self.B * self.T, self.g.C)
        dw = ops.conv_wgrad(x3, dy3, self.g, self.w)
        return dx, dw


def attn_fwd(qkv16, bias, mask8, B, T, heads, p, seed):
    d = qkv16.shape[1] // 3
    hd = d // heads
    out = torch.empty(B * T, d, dtype=BF16, device=qkv16.device)
    lse = torch.empty(B * heads, T, dtype=F32, device=qkv16.device)
    N.call('mpr_attn_fwd', qkv16, bias.detach() if bias is not None else None, mask8, out, lse, B, T, heads, hd,
           1.0 / math.sqrt(hd), float(p), int(seed))
    return out, lse


def attn_bwd(qkv16, bias, mask8, out16, dout16, lse, B, T, heads, p, seed):
    d = qkv16.shape[1] // 3
    hd = d // heads
    dqkv = torch.empty_like(qkv16)
    delta = torch.empty_like(lse)
    N.call('mpr_attn_bwd', qkv16, bias.detach() if bias is not None else None, mask8, out16, dout16, lse, delta, dqkv, B, T,
           heads, hd, 1.0 / math.sqrt(hd), float(p), int(seed))
    return dqkv


class LinearMixedFn(torch.autograd.Function):
    """fp32 in -> bf16 GEMM -> fp32 out (+ bias): a linear at the boundary of the mixed path whose input needs no
    gradient (the ViT patch embedding: its input is the image)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        rows = x.shape[0]
        x16 = cast_bf16(x)
        y16 = _Lin(weight, 1, rows).fwd(x16)
        y, _, _, _, _ = add_ln(None, y16, bias, want_s=True)
        ctx.save_for_backward(x16, weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x16, weight, bias = ctx.saved_tensors
        rows = x16.shape[0]
        dy16, db = ew_bwd(2, dy.contiguous(), rows, weight.shape[0], bias=bias)
        _, dw = _Lin(weight, 1, rows).bwd(x16, dy16, need_dx=False)
        return None, dw, db


# ---------------------------------------------------------------------------------------------- pre-norm (timm ViT)
class PreNormBlockFn(torch.autograd.Function):
    """x += proj(attn(LN1(x)));  x += fc2(gelu(fc1(LN2(x)))).  x: fp32 [B*T, d]."""

    @staticmethod
    def forward(ctx, x, B, T, heads, p, eps, n1w, n1b, qkv_w, qkv_b, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b):
        x = x.contiguous()
        seeds = [next_seed() if p > 0 else 0 for _ in range(3)]
        L = [_Lin(w, B, T) for w in (qkv_w, proj_w, fc1_w, fc2_w)]
        _, _, h16, mean1, rstd1 = add_ln(x, gamma=n1w, beta=n1b, eps=eps, want16=True)
        qkv16 = L[0].fwd(h16)
        a16, lse = attn_fwd(qkv16, qkv_b, None, B, T, heads, 0.0, 0)
        o16 = L[1].fwd(a16)
        x1, _, h2, mean2, rstd2 = add_ln(x, o16, proj_b, p, seeds[0], n2w, n2b, eps, want_s=True, want16=True)
        u16 = L[2].fwd(h2)
        g16 = bias_act(u16, fc1_b, _ACT['gelu'], p, seeds[1])
        f16 = L[3].fwd(g16)
        x2, _, _, _, _ = add_ln(x1, f16, fc2_b, p, seeds[2], want_s=True)
        ctx.save_for_backward(x, mean1, rstd1, h16, qkv16, a16, lse, x1, mean2, rstd2, h2, u16, g16, n1w, n1b, qkv_w, qkv_b,
                              proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b)
        ctx.cfg = (B, T, heads, p, seeds)
        return x2

    @staticmethod
    def backward(ctx, dx2):
        (x, mean1, rstd1, h16, qkv16, a16, lse, x1, mean2, rstd2, h2, u16, g16, n1w, n1b, qkv_w, qkv_b, proj_w, proj_b, n2w,
         n2b, fc1_w, fc1_b, fc2_w, fc2_b) = ctx.saved_tensors
        B, T, heads, p, seeds = ctx.cfg
        rows, d = x.shape
        dx2 = dx2.contiguous()
        L = [_Lin(w, B, T) for w in (qkv_w, proj_w, fc1_w, fc2_w)]
        df16, d_fc2_b = ew_bwd(2, dx2, rows, d, bias=fc2_b, p=p, seed=seeds[2])
        dg16, d_fc2_w = L[3].bwd(g16, df16)
        du16, d_fc1_b = ew_bwd(1, dg16, rows, u16.shape[1], x16=u16, bias=fc1_b, act=_ACT['gelu'], p=p, seed=seeds[1])
        dh2, d_fc1_w = L[2].bwd(h2, du16)
        dx1, d_n2w, d_n2b = ln_bwd(dh2, None, x1, n2w, n2b, mean2, rstd2, dx2)
        do16, d_proj_b = ew_bwd(2, dx1, rows, d, bias=proj_b, p=p, seed=seeds[0])
        da16, d_proj_w = L[1].bwd(a16, do16)
        dqkv16 = attn_bwd(qkv16, qkv_b, None, a16, da16, lse, B, T, heads, 0.0, 0)
        _, d_qkv_b = ew_bwd(0, dqkv16, rows, 3 * d, bias=qkv_b)
        dh16, d_qkv_w = L[0].bwd(h16, dqkv16)
        dx, d_n1w, d_n1b = ln_bwd(dh16, None, x, n1w, n1b, mean1, rstd1, dx1)
        return (dx, None, None, None, None, None, d_n1w, d_n1b, d_qkv_w, d_qkv_b, d_proj_w, d_proj_b, d_n2w, d_n2b, d_fc1_w,
                d_fc1_b, d_fc2_w, d_fc2_b)


def pre_norm_block(blk, x, heads, p_drop, training):
    B, T, d = x.shape
    _check(T, d, heads)
    p = p_drop if training else 0.0
    a, m = blk.attn, blk.mlp
    y = PreNormBlockFn.apply(x.reshape(B * T, d), B, T, heads, p, blk.norm1.eps, blk.norm1.weight, blk.norm1.bias,
                             a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias, blk.norm2.weight, blk.norm2.bias,
                             m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias)
    return y.view(B, T, d)


# ---------------------------------------------------------------------------------------------- post-norm (torch)
class PostNormLayerFn(torch.autograd.Function):
    """x = LN1(x + drop(out_proj(attn(x))));  x = LN2(x + drop(linear2(drop(act(linear1(x)))))).
    Takes the stream as fp32 and as its bf16 copy, returns both (the bf16 copy is the next layer's GEMM operand)."""

    @staticmethod
    def forward(ctx, x, x16, mask8, B, T, heads, p, act, in_w, in_b, out_w, out_b, n1w, n1b, eps1, l1w, l1b, l2w, l2b, n2w,
                n2b, eps2):
        x = x.contiguous()
        seeds = [next_seed() if p > 0 else 0 for _ in range(4)]
        L = [_Lin(w, B, T) for w in (in_w, out_w, l1w, l2w)]
        qkv16 = L[0].fwd(x16)
        a16, lse = attn_fwd(qkv16, in_b, mask8, B, T, heads, p, seeds[0])
        o16 = L[1].fwd(a16)
        s1, x2, x2_16, mean1, rstd1 = add_ln(x, o16, out_b, p, seeds[1], n1w, n1b, eps1, want_s=True, want32=True, want16=True)
        u16 = L[2].fwd(x2_16)
        g16 = bias_act(u16, l1b, act, p, seeds[2])
        f16 = L[3].fwd(g16)
        s2, x3, x3_16, mean2, rstd2 = add_ln(x2, f16, l2b, p, seeds[3], n2w, n2b, eps2, want_s=True, want32=True, want16=True)
        ctx.save_for_backward(x16, mask8, qkv16, a16, lse, s1, mean1, rstd1, x2_16, u16, g16, s2, mean2, rstd2, in_w, in_b,
                              out_w, out_b, n1w, n1b, l1w, l1b, l2w, l2b, n2w, n2b)
        ctx.cfg = (B, T, heads, p, act, seeds)
        ctx.mark_non_differentiable(x3_16)
        return x3, x3_16

    @staticmethod
    def backward(ctx, dy, _unused):
        (x16, mask8, qkv16, a16, lse, s1, mean1, rstd1, x2_16, u16, g16, s2, mean2, rstd2, in_w, in_b, out_w, out_b, n1w,
         n1b, l1w, l1b, l2w, l2b, n2w, n2b) = ctx.saved_tensors
        B, T, heads, p, act, seeds = ctx.cfg
        rows, d = s1.shape
        dy = dy.contiguous()
        L = [_Lin(w, B, T) for w in (in_w, out_w, l1w, l2w)]
        ds2, d_n2w, d_n2b = ln_bwd(None, dy, s2, n2w, n2b, mean2, rstd2, None)
        df16, d_l2b = ew_bwd(2, ds2, rows, d, bias=l2b, p=p, seed=seeds[3])
        dg16, d_l2w = L[3].bwd(g16, df16)
        du16, d_l1b = ew_bwd(1, dg16, rows, u16.shape[1], x16=u16, bias=l1b, act=act, p=p, seed=seeds[2])
        dx2_16, d_l1w = L[2].bwd(x2_16, du16)
        ds1, d_n1w, d_n1b = ln_bwd(dx2_16, ds2, s1, n1w, n1b, mean1, rstd1, None)
        do16, d_out_b = ew_bwd(2, ds1, rows, d, bias=out_b, p=p, seed=seeds[1])
        da16, d_out_w = L[1].bwd(a16, do16)
        dqkv16 = attn_bwd(qkv16, in_b, mask8, a16, da16, lse, B, T, heads, p, seeds[0])
        _, d_in_b = ew_bwd(0, dqkv16, rows, 3 * d, bias=in_b)
        dx16, d_in_w = L[0].bwd(x16, dqkv16)
        dx, _, _, _, _ = add_ln(ds1, dx16, want_s=True)          # fp32 residual path + bf16 GEMM path
        return (dx, None, None, None, None, None, None, None, d_in_w, d_in_b, d_out_w, d_out_b, d_n1w, d_n1b, None, d_l1w,
                d_l1b, d_l2w, d_l2b, d_n2w, d_n2b, None)


def post_norm_layer(layer, x, x16, key_padding_mask, p_drop, training):
    """x: fp32 [B, T, d]; x16: its bf16 copy [B*T, d] (None: made here).  -> (x_next fp32 [B, T, d], bf16 copy)."""
    B, T, d = x.shape
    sa = layer.self_attn
    _check(T, d, sa.num_heads)
    if layer.linear1.weight.shape[0] % 8:
        raise NotImplementedError('bf16-mixed transformer path: dim_feedforward must be a multiple of 8')
    p = p_drop if training else 0.0
    x2d = x.reshape(B * T, d)
    if x16 is None:
        x16 = cast_bf16(x2d)
    mask8 = key_padding_mask.contiguous().view(torch.uint8) if key_padding_mask is not None else None
    name = getattr(layer.activation, '__name__', str(layer.activation))
    act = _ACT['gelu'] if 'gelu' in name else _ACT['relu']
    y, y16 = PostNormLayerFn.apply(x2d, x16, mask8, B, T, sa.num_heads, p, act, sa.in_proj_weight, sa.in_proj_bias,
                                   sa.out_proj.weight, sa.out_proj.bias, layer.norm1.weight, layer.norm1.bias,
                                   layer.norm1.eps, layer.linear1.weight, layer.linear1.bias, layer.linear2.weight,
                                   layer.linear2.bias, layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)
    return y.view(B, T, d), y16
