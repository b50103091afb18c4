"""Sliding-window weight gradient (csrc/conv_wgrad_win.hip), the round-2 forms: K = 64 on twelve waves (two pixel halves of
a 128-pixel chunk, accumulators joined through LDS) and the K % 128 == 0 kernel, against fp32 torch on the same bf16-rounded
operands -- ragged rasters (W + 1 not a multiple of anything), pixel counts that are not multiples of the chunk, splits that
leave the last workgroup a partial range -- and against the round-1 six-wave form (same products, another summation order)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('B,H,W,C', [(64, 56, 56, 64), (38, 19, 23, 64), (9, 56, 40, 64), (340, 7, 7, 64), (40, 28, 20, 128),
                                     (170, 9, 11, 256)])
@pytest.mark.parametrize('target', [256, 512])
def test_window_wgrad_matches_fp32(B, H, W, C, target):
    from multimodal_plankton_recognition_amd import ops, _native as N
    K = C
    g = ops.ConvGeom((K, C, 3, 3), 1, 1)
    gen = torch.Generator().manual_seed(B * 31 + W)
    x = torch.randn(B, H, W, C, generator=gen).to(DEV).to(torch.bfloat16)
    dy = (torch.randn(B, H, W, K, generator=gen) * 0.1).to(DEV).to(torch.bfloat16)
    ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (K, C, 3, 3), dy.float().permute(0, 3, 1, 2), padding=1)
    auto, ops.AUTOTUNE = ops.AUTOTUNE, False
    old_t = N.query('mpr_conv_set_wgrad_target_wgs', target)
    try:
        got = ops.conv_wgrad(x, dy, g, (K, C, 3, 3))
        old_w = N.query('mpr_conv_set_wgrad_window', 2)          # round-1 form
        try:
            six = ops.conv_wgrad(x, dy, g, (K, C, 3, 3))
        finally:
            N.query('mpr_conv_set_wgrad_window', old_w)
    finally:
        N.query('mpr_conv_set_wgrad_target_wgs', old_t)
        ops.AUTOTUNE = auto
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 3e-6 * scale
    assert (six - ref).abs().max().item() <= 3e-6 * scale
    assert (got - six).abs().max().item() <= 3e-6 * scale
