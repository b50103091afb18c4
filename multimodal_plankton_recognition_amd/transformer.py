"""Transformer-encoder building blocks on the gfx950 kernels (fp32): LayerNorm (+ fused residual add),
multi-head self-attention with key-padding mask, bias + GELU/ReLU + dropout, embedding add.

Serves ``ProfileTransformer`` (torch's post-norm ``nn.TransformerEncoderLayer``,
/root/reference/src/profile_encoder.py:22-30,57-68) and the timm ViT image backbones (pre-norm).
Every GEMM -- QKV / output / MLP projections and the per-head QK^T, PV products with their
gradients -- runs on the exact-fp32 MFMA GEMM (``mpr_gemm_f32`` / ``mpr_gemm_f32_b2``).
"""
import math

import torch

from . import _native as N
from . import ops
from .layers import linear, next_seed

F32 = torch.float32

# 'fp32': every transformer GEMM on the exact-fp32 MFMA kernel (parity with the reference's CPU path at rtol 2e-4);
# 'bf16-mixed': fp32 residual stream, bf16 GEMM operands on the bf16 MFMA kernels, fused attention (transformer_mixed.py)
# -- what Lightning's `precision: 16-mixed` / `bf16-mixed` of the reference cards means on this hardware.
_PRECISION = ['fp32']


def set_precision(precision):
    """Trainer / card `precision` -> transformer arithmetic.  Returns the previous mode."""
    old = _PRECISION[0]
    name = str(precision).lower() if precision is not None else '32'
    _PRECISION[0] = 'bf16-mixed' if ('16' in name and 'mixed' in name) or name in ('bf16', '16', 'bf16-true', '16-true') \
        else 'fp32'
    return old


def mixed():
    return _PRECISION[0] == 'bf16-mixed'


class AddLayerNormFn(torch.autograd.Function):
    """y = LayerNorm(x + residual) (residual may be None).  Both inputs receive the same gradient."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, eps):
        x = x.contiguous()
        D = x.shape[-1]
        rows = x.numel() // D
        y = torch.empty_like(x)
        s = torch.empty_like(x) if residual is not None else x
        mean = torch.empty(rows, dtype=F32, device=x.device)
        rstd = torch.empty(rows, dtype=F32, device=x.device)
        N.call('mpr_add_layernorm_fwd', x, residual.contiguous() if residual is not None else None, gamma.detach(),
               beta.detach(), float(eps), y, s if residual is not None else None, mean, rstd, rows, D)
        ctx.save_for_backward(s, gamma, mean, rstd)
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        s, gamma, mean, rstd = ctx.saved_tensors
        D = s.shape[-1]
        rows = s.numel() // D
        dy = dy.contiguous()
        ds = torch.empty_like(s)
        dgamma = torch.empty(D, dtype=F32, device=s.device)
        dbeta = torch.empty(D, dtype=F32, device=s.device)
        ws = torch.empty(N.query('mpr_layernorm_bwd_workspace_floats', rows, D), dtype=F32, device=s.device)
        N.call('mpr_layernorm_bwd', dy, s, gamma.detach(), mean, rstd, None, ds, dgamma, dbeta, ws, 0, rows, D)
        return ds, (ds if ctx.has_res else None), dgamma, dbeta, None


def add_layer_norm(x, residual, gamma, beta, eps):
    return AddLayerNormFn.apply(x, residual, gamma, beta, eps)


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        y = torch.empty_like(a)
        N.call('mpr_add_f32', a, b, y, a.numel())
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class BiasActFn(torch.autograd.Function):
    """y = dropout(act(x + bias));  act: 0 none, 1 exact-erf GELU, 2 ReLU."""

    @staticmethod
    def forward(ctx, x, bias, act, p_drop):
        x = x.contiguous()
        D = x.shape[-1]
        y = torch.empty_like(x)
        mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device) if p_drop > 0 else None
        N.call('mpr_bias_act_fwd', x, bias.detach() if bias is not None else None, act, float(p_drop),
               next_seed() if p_drop > 0 else 0, y, mask, x.numel(), D)
        ctx.save_for_backward(x, bias, mask)
        ctx.cfg = (act, p_drop)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, bias, mask = ctx.saved_tensors
        act, p_drop = ctx.cfg
        D = x.shape[-1]
        dx = torch.empty_like(x)
        N.call('mpr_bias_act_bwd', dy.contiguous(), x, bias.detach() if bias is not None else None, act, float(p_drop), mask, dx,
               x.numel(), D)
        db = None
        if bias is not None:
            rows = x.numel() // D
            ones = torch.ones(rows, 1, dtype=F32, device=x.device)
            db = ops.gemm(ones, dx.view(rows, D), trans_a=True).reshape(-1)
        return dx, db, None, None


def activation_dropout(x, act, p_drop):
    code = {'none': 0, 'gelu': 1, 'relu': 2}[act]
    if code == 0 and p_drop == 0:
        return x
    return BiasActFn.apply(x, None, code, p_drop)


class EmbeddingAddFn(torch.autograd.Function):
    """y[r] = x[r] + table[index[r]]  (nn.Embedding lookup fused with the add; padding row gets no gradient)."""

    @staticmethod
    def forward(ctx, x, table, index, padding_idx):
        x = x.contiguous()
        D = x.shape[-1]
        rows = x.numel() // D
        index = index.contiguous().view(-1)
        y = torch.empty_like(x)
        N.call('mpr_embedding_add_fwd', x, table.detach(), index, y, rows, D)
        ctx.save_for_backward(index)
        ctx.cfg = (table.shape[0], D, -1 if padding_idx is None else int(padding_idx))
        return y

    @staticmethod
    def backward(ctx, dy):
        (index,) = ctx.saved_tensors
        trows, D, pad = ctx.cfg
        dy = dy.contiguous()
        dtable = torch.empty(trows, D, dtype=F32, device=dy.device)
        N.call('mpr_embedding_bwd', dy, index, dtable, trows, dy.numel() // D, D, pad)
        return dy, dtable, None, None


class AttentionFn(torch.autograd.Function):
    """Multi-head self-attention core on a packed qkv [B, T, 3*d] (q | k | v, heads interleaved as torch's MHA /
    timm's Attention do): softmax(q k^T / sqrt(hd) + key-padding mask) v -> [B, T, d]."""

    @staticmethod
    def forward(ctx, qkv, key_padding_mask, heads, p_drop):
        qkv = qkv.contiguous()
        B, T, d3 = qkv.shape
        d = d3 // 3
        hd = d // heads
        dev = qkv.device
        scale = 1.0 / math.sqrt(hd)
        P = torch.empty(B * heads, T, T, dtype=F32, device=dev)
        q, k, v = qkv, qkv.view(-1)[d:], qkv.view(-1)[2 * d:]
        sq = (T * d3, hd)
        # S = Q K^T per (batch, head)
        N.call('mpr_gemm_f32_b2', q, k, P, T, T, hd, d3, d3, T, 0, 1, 1.0, 0.0, B, heads, *sq, *sq, heads * T * T, T * T)
        mask8 = key_padding_mask.contiguous().view(torch.uint8) if key_padding_mask is not None else None
        N.call('mpr_masked_softmax_fwd', P, mask8, scale, B, heads, T, T)
        Pd, dmask = P, None
        if p_drop > 0:
            Pd = torch.empty_like(P)
            dmask = torch.empty(P.shape, dtype=torch.uint8, device=dev)
            N.call('mpr_bias_act_fwd', P, None, 0, float(p_drop), next_seed(), Pd, dmask, P.numel(), T)
        out = torch.empty(B, T, d, dtype=F32, device=dev)
        N.call('mpr_gemm_f32_b2', Pd, v, out, T, hd, T, T, d3, d, 0, 0, 1.0, 0.0, B, heads, heads * T * T, T * T, *sq,
               T * d, hd)
        ctx.save_for_backward(qkv, P, Pd if p_drop > 0 else None, dmask)
        ctx.cfg = (heads, p_drop, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, P, Pd, dmask = ctx.saved_tensors
        heads, p_drop, scale = ctx.cfg
        B, T, d3 = qkv.shape
        d = d3 // 3
        hd = d // heads
        dev = qkv.device
        dout = dout.contiguous()
        if Pd is None:
            Pd = P
        q, k, v = qkv, qkv.view(-1)[d:], qkv.view(-1)[2 * d:]
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = dqkv, dqkv.view(-1)[d:], dqkv.view(-1)[2 * d:]
        sq = (T * d3, hd)
        so = (T * d, hd)
        sp = (heads * T * T, T * T)
        # dPd = dO V^T ; dV = Pd^T dO
        dP = torch.empty_like(P)
        N.call('mpr_gemm_f32_b2', dout, v, dP, T, T, hd, d, d3, T, 0, 1, 1.0, 0.0, B, heads, *so, *sq, *sp)
        N.call('mpr_gemm_f32_b2', Pd, dout, dv, T, hd, T, T, d, d3, 1, 0, 1.0, 0.0, B, heads, *sp, *so, *sq)
        if p_drop > 0:
            N.call('mpr_bias_act_bwd', dP, None, None, 0, float(p_drop), dmask, dP, dP.numel(), T)
        N.call('mpr_softmax_bwd', dP, P, scale, B * heads * T, T)        # dP <- dS (w.r.t. the raw q.k products)
        # dQ = dS K ; dK = dS^T Q
        N.call('mpr_gemm_f32_b2', dP, k, dq, T, hd, T, T, d3, d3, 0, 0, 1.0, 0.0, B, heads, *sp, *sq, *sq)
        N.call('mpr_gemm_f32_b2', dP, q, dk, T, hd, T, T, d3, d3, 1, 0, 1.0, 0.0, B, heads, *sp, *sq, *sq)
        return dqkv, None, None, None


def attention(qkv, key_padding_mask, heads, p_drop=0.0):
    return AttentionFn.apply(qkv, key_padding_mask, heads, p_drop)


def post_norm_layer(layer, x, key_padding_mask, p_drop, training):
    """One torch ``nn.TransformerEncoderLayer`` (norm_first=False, batch_first=True): parameters are read from
    the torch module (state_dict keys stay the reference's), arithmetic runs on the HIP kernels."""
    B, T, d = x.shape
    p = p_drop if training else 0.0
    sa = layer.self_attn
    qkv = linear(x.reshape(B * T, d), sa.in_proj_weight, sa.in_proj_bias).view(B, T, 3 * d)
    a = attention(qkv, key_padding_mask, sa.num_heads, p)
    a = linear(a.view(B * T, d), sa.out_proj.weight, sa.out_proj.bias)
    a = activation_dropout(a, 'none', p)
    x2 = add_layer_norm(x.reshape(B * T, d), a, layer.norm1.weight, layer.norm1.bias, layer.norm1.eps)
    act = 'gelu' if getattr(layer.activation, '__name__', str(layer.activation)).find('gelu') >= 0 else 'relu'
    h = linear(x2, layer.linear1.weight, layer.linear1.bias)
    h = activation_dropout(h, act, p)
    h = linear(h, layer.linear2.weight, layer.linear2.bias)
    h = activation_dropout(h, 'none', p)
    x3 = add_layer_norm(x2, h, layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)
    return x3.view(B, T, d)


def pre_norm_block(blk, x, heads, p_drop, training):
    """One timm ViT block: x += proj(attn(LN(x))); x += fc2(gelu(fc1(LN(x))))."""
    if mixed():
        from . import transformer_mixed
        return transformer_mixed.pre_norm_block(blk, x, heads, p_drop, training)
    B, T, d = x.shape
    p = p_drop if training else 0.0
    x2 = x.reshape(B * T, d)
    h = add_layer_norm(x2, None, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
    qkv = linear(h, blk.attn.qkv.weight, blk.attn.qkv.bias).view(B, T, 3 * d)
    a = attention(qkv, None, heads, 0.0)
    a = linear(a.view(B * T, d), blk.attn.proj.weight, blk.attn.proj.bias)
    x2 = AddFn.apply(x2, activation_dropout(a, 'none', p))
    h = add_layer_norm(x2, None, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
    h = activation_dropout(linear(h, blk.mlp.fc1.weight, blk.mlp.fc1.bias), 'gelu', p)
    h = linear(h, blk.mlp.fc2.weight, blk.mlp.fc2.bias)
    x2 = AddFn.apply(x2, activation_dropout(h, 'none', p))
    return x2.view(B, T, d)


def post_norm_stack(layers, x, key_padding_mask, p_drop, training):
    """A stack of torch post-norm encoder layers (nn.TransformerEncoder without final norm)."""
    if mixed():
        from . import transformer_mixed
        x16 = None
        for layer in layers:
            x, x16 = transformer_mixed.post_norm_layer(layer, x, x16, key_padding_mask, p_drop, training)
        return x
    for layer in layers:
        x = post_norm_layer(layer, x, key_padding_mask, p_drop, training)
    return x
