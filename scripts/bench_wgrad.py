"""Weight-gradient kernel on the ResNet-18 shapes (batch 512) for several workgroup-count targets."""
import sys, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
B = 512
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
targets = [int(v) for v in sys.argv[1:]] or [768, 512, 1024]
for name, H, C, K, R, st, pad in [('l1 3x3 64->64', 56, 64, 64, 3, 1, 1), ('l2 3x3 128->128', 28, 128, 128, 3, 1, 1), ('l3 3x3 256->256', 14, 256, 256, 3, 1, 1),
                                 ('l4 3x3 512->512', 7, 512, 512, 3, 1, 1), ('l2.0 3x3/2 64->128', 56, 64, 128, 3, 2, 1), ('l3.0 3x3/2 128->256', 28, 128, 256, 3, 2, 1),
                                 ('l4.0 3x3/2 256->512', 14, 256, 512, 3, 2, 1), ('l2.ds 1x1/2', 56, 64, 128, 1, 2, 0), ('l3.ds 1x1/2', 28, 128, 256, 1, 2, 0), ('l4.ds 1x1/2', 14, 256, 512, 1, 2, 0)]:
    g = ops.ConvGeom((K, C, R, R), st, pad)
    P = (H + 2 * pad - R) // st + 1
    x = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
    dy = torch.randn(B, P, P, K, device='cuda').to(torch.bfloat16)
    flop = 2.0 * B * P * P * K * C * R * R
    row = [f'{name:20s}']
    N.query('mpr_conv_set_wgrad_window', 0)
    N.query('mpr_conv_set_wgrad_target_wgs', 768)
    ref = ops.conv_wgrad(x, dy, g, (K, C, R, R))
    t = timeit(lambda: ops.conv_wgrad(x, dy, g, (K, C, R, R)))
    row.append(f'gather kernel: {t:6.1f}us {flop/t/1e6:4.0f}TF')
    for tg in targets:
        N.query('mpr_conv_set_wgrad_window', 1 + 256)
        N.query('mpr_conv_set_wgrad_target_wgs', tg)
        t = timeit(lambda: ops.conv_wgrad(x, dy, g, (K, C, R, R)))
        row.append(f'target {tg} NO EPILOGUE: {t:6.1f}us')
    N.query('mpr_conv_set_wgrad_window', 1)
    for tg in targets:
        N.query('mpr_conv_set_wgrad_target_wgs', tg)
        got = ops.conv_wgrad(x, dy, g, (K, C, R, R))
        err = ((got - ref).abs().max() / ref.abs().max()).item()
        t = timeit(lambda: ops.conv_wgrad(x, dy, g, (K, C, R, R)))
        row.append(f'target {tg}: {t:6.1f}us {flop/t/1e6:4.0f}TF (rel err {err:.1e})')
    print(' | '.join(row), flush=True)
N.query('mpr_conv_set_wgrad_target_wgs', 512)
