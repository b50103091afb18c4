"""Stride-2 data gradient: parity-class walk vs. all-taps-with-holes, on the three ResNet-18 transition shapes."""
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
for name, H, C, K, R, pad in [('l2.0 3x3/2 64->128', 56, 64, 128, 3, 1), ('l3.0 3x3/2 128->256', 28, 128, 256, 3, 1),
                              ('l4.0 3x3/2 256->512', 14, 256, 512, 3, 1), ('l2.ds 1x1/2 64->128', 56, 64, 128, 1, 0),
                              ('l3.ds 1x1/2 128->256', 28, 128, 256, 1, 0), ('l4.ds 1x1/2 256->512', 14, 256, 512, 1, 0)]:
    g = ops.ConvGeom((K, C, R, R), 2, pad)
    w = torch.randn(K, C, R, R, device='cuda') * 0.05
    wf, wd = ops.packed_weights(w, g)
    P = (H + 2 * pad - R) // 2 + 1
    dy = torch.randn(B, P, P, K, device='cuda').to(torch.bfloat16)
    flop = 2.0 * B * P * P * K * C * R * R
    row = [name]
    outs = []
    for par in (0, 1):
        N.query('mpr_conv_set_dgrad_parity', par)
        td = timeit(lambda: ops.conv_dgrad(dy, wd, g, (B, H, H, C)))
        outs.append(ops.conv_dgrad(dy, wd, g, (B, H, H, C)))
        row.append(f'parity={par} {td:6.1f}us {flop/td/1e6:5.0f}TF')
    row.append('bit-equal' if torch.equal(outs[0], outs[1]) else f'DIFF {(outs[0].float()-outs[1].float()).abs().max().item():.3g}')
    print(' | '.join(row))
