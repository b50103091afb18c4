"""Sliding-window weight gradient on v_mfma_f32_16x16x32_bf16 (default) against the 32x32x16 form (debug bit 6) and fp32 torch."""
import sys, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
ops.AUTOTUNE = False
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for rep in range(2):
  for B, H, W, C in [(512, 56, 56, 64), (512, 28, 28, 128), (512, 14, 14, 256), (512, 7, 7, 512), (64, 56, 56, 64), (37, 19, 23, 64), (40, 28, 20, 128), (33, 9, 11, 256)]:
    K = C
    g = ops.ConvGeom((K, C, 3, 3), 1, 1)
    x = torch.randn(B, H, W, C, device='cuda').to(torch.bfloat16)
    dy = (torch.randn(B, H, W, K, device='cuda') * 0.1).to(torch.bfloat16)
    ref = None
    if B * H * W <= 64 * 56 * 56:
        ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (K, C, 3, 3), dy.float().permute(0, 3, 1, 2), padding=1)
    for tgt in (160, 256):
        N.query('mpr_conv_set_wgrad_target_wgs', tgt)
        out = {}
        for name, mode in [('16x16x32', 1), ('32x32x16', 1 | (64 << 8))]:
            N.query('mpr_conv_set_wgrad_window', mode)
            dw = ops.conv_wgrad(x, dy, g, (K, C, 3, 3))
            t = timeit(lambda: ops.conv_wgrad(x, dy, g, (K, C, 3, 3)))
            out[name] = (dw, t)
        N.query('mpr_conv_set_wgrad_window', 1)
        a, b = out['16x16x32'][0], out['32x32x16'][0]
        err = float((a - ref).abs().max() / ref.abs().max()) if ref is not None else float('nan')
        err0 = float((b - ref).abs().max() / ref.abs().max()) if ref is not None else float('nan')
        print(f"B={B} {H}x{W} C={C} target {tgt}: 32x32x16 {out['32x32x16'][1]:7.1f} us -> 16x16x32 {out['16x16x32'][1]:7.1f} us | "
              f"diff {float((a - b).abs().max() / b.abs().max()):.1e}  vs fp32 torch {err:.1e} ({err0:.1e})", flush=True)
