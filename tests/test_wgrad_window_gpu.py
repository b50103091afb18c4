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
            # the 32x32x16 form of the round-2 / round-3 loop (debug bit 6; the default is v_mfma_f32_16x16x32_bf16, whose lane
            # groups contract over another pixel order and whose DMA swizzles the 32-byte units as well)
            N.query('mpr_conv_set_wgrad_window', 1 | (64 << 8))
            m32 = ops.conv_wgrad(x, dy, g, (K, C, 3, 3))
            N.query('mpr_conv_set_wgrad_window', 2 | (64 << 8))
            six32 = ops.conv_wgrad(x, dy, g, (K, C, 3, 3))
        finally:
            N.query('mpr_conv_set_wgrad_window', old_w)
    finally:
        N.query('mpr_conv_set_wgrad_target_wgs', old_t)
        ops.AUTOTUNE = auto
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 3e-6 * scale
    assert (six - ref).abs().max().item() <= 3e-6 * scale
    assert (got - six).abs().max().item() <= 3e-6 * scale
    assert (m32 - ref).abs().max().item() <= 3e-6 * scale and (six32 - ref).abs().max().item() <= 3e-6 * scale


@pytest.mark.parametrize('B,H,W,C', [(40, 28, 20, 128), (9, 56, 40, 64)])
def test_wgrad_ex_carries_its_choices_as_arguments(B, H, W, C):
    """mpr_conv_wgrad_ex (round 3): kernel choice, workgroup target and scratch loan are ARGUMENTS of the launch -- the
    sliding-window kernel with partial slices, with fp32 atomics (no scratch), the gather kernel, and the library defaults all
    give the same gradient; the process-global knobs are left untouched."""
    from multimodal_plankton_recognition_amd import ops, _native as N
    K = C
    g = ops.ConvGeom((K, C, 3, 3), 1, 1)
    gen = torch.Generator().manual_seed(7 * B + C)
    x = torch.randn(B, H, W, C, generator=gen).to(DEV).to(torch.bfloat16)
    dy = (torch.randn(B, H, W, K, generator=gen) * 0.1).to(DEV).to(torch.bfloat16)
    ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (K, C, 3, 3), dy.float().permute(0, 3, 1, 2), padding=1)
    knobs = (N.query('mpr_conv_set_wgrad_target_wgs', 512), N.query('mpr_conv_set_wgrad_window', 1))
    N.query('mpr_conv_set_wgrad_target_wgs', knobs[0]); N.query('mpr_conv_set_wgrad_window', knobs[1])
    scale = ref.abs().max().item()

    def run(choice, scratch=True):
        ws = torch.empty(K * 9 * C, dtype=torch.float32, device=DEV)
        dw = torch.empty(K, C, 3, 3, dtype=torch.float32, device=DEV)
        old = ops.USE_WGRAD_SCRATCH
        ops.USE_WGRAD_SCRATCH = scratch
        try:
            ops._wgrad_call(x, dy, ws, dw, 0, B, H, W, C, K, *g.tail, choice=choice)
        finally:
            ops.USE_WGRAD_SCRATCH = old
        return dw

    for choice, scratch in [(None, True), ((1, 160), True), ((1, 256), False), ((0, 256), True), ((-1, 0), True)]:
        got = run(choice, scratch)
        assert (got - ref).abs().max().item() <= 3e-6 * scale, (choice, scratch)
    after = (N.query('mpr_conv_set_wgrad_target_wgs', knobs[0]), N.query('mpr_conv_set_wgrad_window', knobs[1]))
    assert after == knobs
