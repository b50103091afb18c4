"""Micro-benchmark of the implicit-GEMM conv kernels on the ResNet-18 layer shapes at batch 512."""
import sys, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
import os
B = 512
if os.environ.get('MPR_WIN') == '0':
    N.query('mpr_conv_set_window', 0)
shapes = [('l1 3x3 64->64', 56, 64, 64, 3, 1, 1), ('l2 3x3 128->128', 28, 128, 128, 3, 1, 1),
          ('l3 3x3 256->256', 14, 256, 256, 3, 1, 1), ('l4 3x3 512->512', 7, 512, 512, 3, 1, 1),
          ('l2.0 3x3/2 64->128', 56, 64, 128, 3, 2, 1)]
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
variants = [(0, 0), (1, 1), (0, 2)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for name, H, C, K, R, st, pad in shapes:
    g = ops.ConvGeom((K, C, R, R), st, pad)
    w = torch.randn(K, C, R, R, device='cuda') * 0.05
    wf, wd = ops.packed_weights(w, g)
    x = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
    P = (H + 2 * pad - R) // st + 1
    dy = torch.randn(B, P, P, K, device='cuda').to(torch.bfloat16)
    flop = 2.0 * B * P * P * K * C * R * R
    row = [name]
    for nv, wv in variants:
        N.query('mpr_conv_set_variant', nv, wv)
        tf = timeit(lambda: ops.conv_fwd(x, wf, g, True))
        td = timeit(lambda: ops.conv_dgrad(dy, wd, g, x.shape))
        row.append(f'v({nv},{wv}) fwd {tf:6.1f}us {flop/tf/1e6:5.0f}TF dgrad {td:6.1f}us {flop/td/1e6:5.0f}TF')
    N.query('mpr_conv_set_variant', 0, 0)
    tw = timeit(lambda: ops.conv_wgrad(x, dy, g, (K, C, R, R)))
    row.append(f'wgrad {tw:6.1f}us {flop/tw/1e6:5.0f}TF')
    print(' | '.join(row))
