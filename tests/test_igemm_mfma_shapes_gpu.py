"""LDS-DMA implicit GEMM (csrc/conv_igemm.hip: stride-2 / 1x1 convolutions behind /root/reference/src/image_encoder.py:24, the
transformer's linears behind /root/reference/src/profile_encoder.py:22-30) on v_mfma_f32_16x16x32_bf16 (the default) against its
32x32x16 form (debug bit 5 of mpr_conv_debug_drop_operand): forward and data gradient, bit for bit -- an instruction pair sums the
same 32-term group -- and against fp32 torch on the same bf16-rounded operands."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('B,H,W,C,K,k,stride,pad', [(32, 28, 28, 128, 256, 3, 2, 1), (24, 56, 56, 64, 128, 3, 2, 1),
                                                    (32, 28, 28, 128, 256, 1, 2, 0), (90, 14, 14, 256, 512, 3, 2, 1),
                                                    (340, 7, 7, 512, 512, 3, 1, 1), (16, 32, 32, 768, 2304, 1, 1, 0),
                                                    (40, 29, 23, 64, 64, 3, 2, 1), (20, 30, 30, 192, 320, 1, 1, 0)])
def test_dma_implicit_gemm_mfma_shapes_agree(B, H, W, C, K, k, stride, pad):
    from multimodal_plankton_recognition_amd import ops, _native as N
    g = ops.ConvGeom((K, C, k, k), stride, pad)
    gen = torch.Generator().manual_seed(B + C + K)
    w = (torch.randn(K, C, k, k, generator=gen) * 0.05).to(DEV)
    wf, wd = ops.packed_weights(w, g)
    x = torch.randn(B, H, W, C, generator=gen).to(DEV).to(torch.bfloat16)
    wq = w.to(torch.bfloat16).float()
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wq, None, stride, pad).permute(0, 2, 3, 1)
    dy = torch.randn(*ref.shape, generator=gen).to(DEV).to(torch.bfloat16)
    refd = torch.nn.grad.conv2d_input((B, C, H, W), wq, dy.float().permute(0, 3, 1, 2), stride, pad).permute(0, 2, 3, 1)
    old_win = N.query('mpr_conv_set_window', 0)            # (3x3 / stride 1 would take the window kernel)
    out = {}
    try:
        for form, bits in (('m16', 0), ('m32', 32)):
            old = N.query('mpr_conv_debug_drop_operand', bits)
            try:
                y, st = ops.conv_fwd(x, wf, g, True)
                dx = ops.conv_dgrad(dy, wd, g, tuple(x.shape))
                out[form] = (y.clone(), st.double().sum(0), dx.clone())
            finally:
                N.query('mpr_conv_debug_drop_operand', old)
    finally:
        N.query('mpr_conv_set_window', old_win)
    (y6, s6, d6), (y3, s3, d3) = out['m16'], out['m32']
    assert torch.equal(y6, y3) and torch.equal(d6, d3)
    assert (s6 - s3).abs().max().item() <= 1e-4 * s3.abs().max().item()
    assert (y6.float() - ref).abs().max().item() <= 8e-3 * ref.abs().max().item()
    assert (d6.float() - refd).abs().max().item() <= 8e-3 * refd.abs().max().item()


@pytest.mark.parametrize('B,H,W,C,K,k,stride,pad', [(24, 28, 28, 96, 24, 1, 1, 0), (90, 14, 14, 480, 112, 1, 1, 0),
                                                    (90, 14, 14, 672, 192, 1, 1, 0), (24, 28, 28, 96, 160, 3, 2, 1),
                                                    (24, 28, 28, 160, 96, 1, 1, 0)])
def test_dma_implicit_gemm_on_32_channel_multiples(B, H, W, C, K, k, stride, pad):
    """Source channels a multiple of 32 but not of 64 (EfficientNet-B0's 96 / 480 / 672-wide maps,
    /root/reference/model_cards/example_multi.yaml:9): the LDS-DMA ring on 32-deep chunks, forward (source = C) and data gradient
    (source = K), against fp32 torch on the same bf16-rounded operands."""
    from multimodal_plankton_recognition_amd import ops
    g = ops.ConvGeom((K, C, k, k), stride, pad)
    gen = torch.Generator().manual_seed(B + C + K)
    w = (torch.randn(K, C, k, k, generator=gen) * 0.05).to(DEV)
    wf, wd = ops.packed_weights(w, g)
    x = torch.randn(B, H, W, C, generator=gen).to(DEV).to(torch.bfloat16)
    wq = w.to(torch.bfloat16).float()
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wq, None, stride, pad).permute(0, 2, 3, 1)
    dy = torch.randn(*ref.shape, generator=gen).to(DEV).to(torch.bfloat16)
    refd = torch.nn.grad.conv2d_input((B, C, H, W), wq, dy.float().permute(0, 3, 1, 2), stride, pad).permute(0, 2, 3, 1)
    y, st = ops.conv_fwd(x, wf, g, True)
    dx = ops.conv_dgrad(dy, wd, g, tuple(x.shape))
    assert (y.float() - ref).abs().max().item() <= 8e-3 * ref.abs().max().item()
    assert (dx.float() - refd).abs().max().item() <= 8e-3 * refd.abs().max().item()
    yq = y.double().reshape(-1, K)
    want = torch.stack([yq.sum(0), (yq * yq).sum(0)])
    assert (st.double().sum(0) - want).abs().max().item() <= 1e-4 * want.abs().max().item()
