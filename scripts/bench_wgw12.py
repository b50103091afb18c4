"""Sliding-window weight gradient: round-2 forms (K = 64: twelve waves on two pixel halves; K % 128 == 0: two pixel blocks per
chunk) against the round-1 forms and fp32 torch."""
import sys, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
import torch.nn.functional as F
ops.AUTOTUNE = False
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for B, H, W, C in [(512, 56, 56, 64), (64, 56, 56, 64), (37, 19, 23, 64), (512, 28, 28, 128), (512, 14, 14, 256), (512, 7, 7, 512), (40, 28, 20, 128), (33, 9, 11, 256)]:
    K = C
    g = ops.ConvGeom((K, C, 3, 3), 1, 1)
    x = torch.randn(B, H, W, C, device='cuda').to(torch.bfloat16)
    dy = (torch.randn(B, H, W, K, device='cuda') * 0.1).to(torch.bfloat16)
    ref = None
    if B * H * W <= 64 * 56 * 56:
        ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (K, C, 3, 3), dy.float().permute(0, 3, 1, 2), padding=1)
    for tgt in (160, 256, 512):
        N.query('mpr_conv_set_wgrad_target_wgs', tgt)
        for name, mode in [('round 2', 1), ('round 1', 2)]:
            N.query('mpr_conv_set_wgrad_window', mode)
            dw = ops.conv_wgrad(x, dy, g, (K, C, 3, 3))
            t = timeit(lambda: ops.conv_wgrad(x, dy, g, (K, C, 3, 3)))
            err = float((dw - ref).abs().max() / ref.abs().max()) if ref is not None else float('nan')
            print(f'B={B} {H}x{W} target {tgt} {name:9s}: {t:7.1f} us   rel err vs fp32 torch {err:.2e}  checksum {dw.double().sum().item():.8e}', flush=True)
N.query('mpr_conv_set_wgrad_window', 1)
