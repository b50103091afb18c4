"""Sliding-window weight gradient, round-3 experiments: dbg flag sets (bit 4 = 16: s_setprio by k-step) x workgroup targets."""
import sys, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
ops.AUTOTUNE = False
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
flags = [int(v) for v in sys.argv[1:]] or [0, 16]
for B, H, W, C in [(512, 56, 56, 64), (512, 28, 28, 128), (512, 14, 14, 256), (512, 7, 7, 512)]:
    K = C
    g = ops.ConvGeom((K, C, 3, 3), 1, 1)
    x = torch.randn(B, H, W, C, device='cuda').to(torch.bfloat16)
    dy = (torch.randn(B, H, W, K, device='cuda') * 0.1).to(torch.bfloat16)
    flop = 2.0 * B * H * W * K * C * 9
    for fl in flags:
        N.query('mpr_conv_set_wgrad_window', 1 | (fl << 8))
        row = []
        for tgt in (160, 192, 256):
            N.query('mpr_conv_set_wgrad_target_wgs', tgt)
            dw = ops.conv_wgrad(x, dy, g, (K, C, 3, 3))
            t = timeit(lambda: ops.conv_wgrad(x, dy, g, (K, C, 3, 3)))
            row.append(f'{tgt}: {t:6.1f} us {flop / t / 1e6:4.0f} TF')
        print(f'{H}x{W}x{C} dbg {fl:2d} | ' + ' | '.join(row) + f' | checksum {dw.double().sum().item():.6e}', flush=True)
N.query('mpr_conv_set_wgrad_window', 1)
N.query('mpr_conv_set_wgrad_target_wgs', 512)
