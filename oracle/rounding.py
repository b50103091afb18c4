"""Oracle (test infrastructure): optional emulation of the HIP path's storage precision.

The shipped kernels keep feature maps and conv weights in bf16 (fp32 accumulate, fp32 statistics,
fp32 heads/loss).  With ``emulate_bf16()`` active, the SAME oracle functions round at exactly those
storage points (straight-through for autograd), which turns the parity check of the kernels into a
tight one (accumulation-order differences only).  With it inactive (default) the oracle is the
reference's plain fp32 arithmetic, which is what the golden fixtures pin.
"""
import contextlib
import torch

_STATE = {'on': False}


@contextlib.contextmanager
def emulate_bf16(on=True):
    old = _STATE['on']
    _STATE['on'] = on
    try:
        yield
    finally:
        _STATE['on'] = old


def r(x):
    """Round a stored activation / packed weight to bf16 (identity gradient)."""
    if not _STATE['on']:
        return x
    return x + (x.to(torch.bfloat16).to(x.dtype) - x).detach()
