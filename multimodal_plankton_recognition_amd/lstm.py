"""nn.LSTM (batch_first, unidirectional, inter-layer dropout) on the gfx950 kernels, fp32 -- the recurrent core of
ProfileLSTM (/root/reference/src/profile_encoder.py:71-108).

One autograd Function for the whole stack.  Activations are time-major ``[T, B, *]`` so that step t of the layer-wide
input projection ``x W_ih^T + b_ih`` (ONE exact-fp32 MFMA GEMM per layer) is a contiguous ``[B, 4d]`` slice; the
recurrent product ``h_{t-1} W_hh^T`` accumulates into that slice (``mpr_gemm_f32`` with beta = 1) and the pointwise cell
(``mpr_lstm_cell_fwd``) applies the gates.  Backward walks the steps in reverse (``mpr_lstm_cell_bwd`` + one
``dG_t W_hh`` product per step) and forms every weight gradient with one GEMM over all steps.  The reference reads the
output at ``last_idx`` only, so the gather (and its scatter in backward) is part of the Function.
"""
import torch

from . import _native as N
from . import ops
from .layers import next_seed

F32 = torch.float32


def _ones(n, dev):
    return torch.ones(n, 1, dtype=F32, device=dev)


class LSTMStackFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, last_idx, p_drop, *weights):
        B, T, _ = x.shape
        dev = x.device
        L = len(weights) // 4
        inp = x.transpose(0, 1).contiguous()                     # [T, B, d_in]
        saved, masks = [], []
        for l in range(L):
            w_ih, w_hh, b_ih, b_hh = weights[4 * l:4 * l + 4]
            d = w_hh.shape[1]
            G = ops.gemm(inp.view(T * B, -1), w_ih.detach(), trans_b=True, bias=b_ih.detach()).view(T, B, 4 * d)
            act = torch.empty(T, B, 4 * d, dtype=F32, device=dev)
            C = torch.empty(T, B, d, dtype=F32, device=dev)
            H = torch.empty(T, B, d, dtype=F32, device=dev)
            for t in range(T):
                if t > 0:
                    ops.gemm(H[t - 1], w_hh.detach(), trans_b=True, out=G[t], beta=1.0)
                N.call('mpr_lstm_cell_fwd', G[t], b_hh.detach(), C[t - 1] if t else None, act[t], C[t], H[t], B, d)
            saved += [inp, act, C, H]
            mask = None
            if l < L - 1 and p_drop > 0:                         # nn.LSTM: dropout on every layer's output but the last
                out = torch.empty_like(H)
                mask = torch.empty(H.shape, dtype=torch.uint8, device=dev)
                N.call('mpr_bias_act_fwd', H, None, 0, float(p_drop), next_seed(), out, mask, H.numel(), d)
                inp = out
            else:
                inp = H
            masks.append(mask)
        batch = torch.arange(B, device=dev)
        ctx.save_for_backward(last_idx, *saved, *[m for m in masks if m is not None], *weights)
        ctx.cfg = (B, T, L, p_drop, [m is not None for m in masks])
        return H[last_idx, batch].contiguous()                   # x[arange(B), last_idx]  (:101)

    @staticmethod
    def backward(ctx, dsel):
        B, T, L, p_drop, has_mask = ctx.cfg
        st = ctx.saved_tensors
        last_idx, saved = st[0], st[1:1 + 4 * L]
        nm = sum(has_mask)
        mask_list = list(st[1 + 4 * L:1 + 4 * L + nm])
        weights = st[1 + 4 * L + nm:]
        dev = dsel.device
        d_top = weights[4 * (L - 1) + 1].shape[1]
        dH = torch.zeros(T, B, d_top, dtype=F32, device=dev)
        dH[last_idx, torch.arange(B, device=dev)] = dsel.contiguous().float()
        grads = [None] * (4 * L)
        for l in reversed(range(L)):
            inp, act, C, H = saved[4 * l:4 * l + 4]
            w_ih, w_hh, b_ih, b_hh = weights[4 * l:4 * l + 4]
            d = w_hh.shape[1]
            dG = torch.empty(T, B, 4 * d, dtype=F32, device=dev)
            dc = torch.zeros(B, d, dtype=F32, device=dev)
            dh_rec = None
            for t in reversed(range(T)):
                N.call('mpr_lstm_cell_bwd', act[t], C[t - 1] if t else None, C[t], dH[t], dh_rec, dc, dG[t], B, d)
                if t > 0:
                    dh_rec = ops.gemm(dG[t], w_hh.detach())
            dG2 = dG.view(T * B, 4 * d)
            grads[4 * l] = ops.gemm(dG2, inp.view(T * B, -1), trans_a=True)
            grads[4 * l + 1] = (ops.gemm(dG[1:].reshape((T - 1) * B, 4 * d), H[:-1].reshape((T - 1) * B, d), trans_a=True)
                                if T > 1 else torch.zeros_like(w_hh))
            db = ops.gemm(_ones(T * B, dev), dG2, trans_a=True).reshape(-1)
            grads[4 * l + 2], grads[4 * l + 3] = db, db.clone()
            if l > 0 or ctx.needs_input_grad[0]:
                dinp = ops.gemm(dG2, w_ih.detach()).view(T, B, -1)
                if l > 0 and has_mask[l - 1]:
                    mask = mask_list[sum(has_mask[:l - 1])]
                    N.call('mpr_bias_act_bwd', dinp, None, None, 0, float(p_drop), mask, dinp, dinp.numel(), dinp.shape[-1])
                dH = dinp
        dx = dH.transpose(0, 1).contiguous() if ctx.needs_input_grad[0] else None
        return (dx, None, None, *grads)


def lstm_stack(lstm, x, last_idx, training):
    """x: [B, T, d] fp32 -> h_T-like features [B, d] read at last_idx.  `lstm` is the nn.LSTM parameter container."""
    weights = []
    for l in range(lstm.num_layers):
        weights += [getattr(lstm, f'weight_ih_l{l}'), getattr(lstm, f'weight_hh_l{l}'), getattr(lstm, f'bias_ih_l{l}'),
                    getattr(lstm, f'bias_hh_l{l}')]
    p = float(lstm.dropout) if training else 0.0
    return LSTMStackFn.apply(x.contiguous().float(), last_idx.contiguous(), p, *weights)
