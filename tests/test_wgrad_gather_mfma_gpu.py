"""LDS-DMA gather weight gradient (csrc/conv_wgrad.hip: conv_wgrad_dma_kernel -- stride-2 and 1x1 convolutions behind
/root/reference/src/image_encoder.py:24, the transformer's linears behind /root/reference/src/profile_encoder.py:22-30) on
v_mfma_f32_16x16x32_bf16 (mpr_conv_debug_wgrad_mfma16: another contraction order over a chunk's pixels, a 32-byte-unit swizzle on
the DMA's source side; built, not the default) and on its 32x32x16 form against fp32 torch, on every tile shape the launcher picks."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('B,H,W,C,K,k,stride,pad,tile', [
    (24, 56, 56, 64, 128, 3, 2, 1, 1),       # <2,3>
    (90, 28, 28, 128, 256, 3, 2, 1, 1),      # <2,2>
    (24, 56, 56, 64, 128, 1, 2, 0, 1),       # <2,1>
    (24, 28, 28, 64, 64, 1, 1, 0, 1),        # <1,1>
    (24, 28, 28, 128, 64, 1, 1, 0, 1),       # <1,2>
    (90, 28, 28, 64, 64, 3, 2, 1, 1),        # <1,3>
    (18, 32, 32, 768, 1536, 1, 1, 0, 1),     # <4,4> (16 waves: one fragment set)
    (18, 32, 32, 768, 1536, 1, 1, 0, 2),     # <4,2>
    (18, 32, 32, 512, 768, 1, 1, 0, 3),      # <2,4>
    (100, 29, 23, 64, 128, 3, 2, 1, 1)])     # ragged: a partial last chunk
def test_gather_weight_gradient_mfma_shapes(B, H, W, C, K, k, stride, pad, tile):
    from multimodal_plankton_recognition_amd import ops, _native as N
    g = ops.ConvGeom((K, C, k, k), stride, pad)
    gen = torch.Generator().manual_seed(B + C + K + tile)
    x = torch.randn(B, H, W, C, generator=gen).to(DEV).to(torch.bfloat16)
    P, Q = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    dy = (torch.randn(B, P, Q, K, generator=gen) * 0.1).to(DEV).to(torch.bfloat16)
    ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (K, C, k, k), dy.float().permute(0, 3, 1, 2), stride, pad)
    auto, ops.AUTOTUNE = ops.AUTOTUNE, False
    old_tile = N.query('mpr_conv_set_wgrad_tile', tile)
    out = {}
    try:
        for form, flag in (('m16', 1), ('m32', 0)):
            old = N.query('mpr_conv_debug_wgrad_mfma16', flag)
            try:
                out[form] = ops.conv_wgrad(x, dy, g, (K, C, k, k)).clone()
            finally:
                N.query('mpr_conv_debug_wgrad_mfma16', old)
    finally:
        N.query('mpr_conv_set_wgrad_tile', old_tile)
        ops.AUTOTUNE = auto
    scale = ref.abs().max().item()
    assert (out['m32'] - ref).abs().max().item() <= 5e-6 * scale
    assert (out['m16'] - ref).abs().max().item() <= 5e-6 * scale
