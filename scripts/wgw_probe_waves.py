"""Per-WAVE shader-clock breakdown of the sliding-window weight-gradient kernel (dbg bit 2): who waits at the barrier."""
import sys, ctypes, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
ops.AUTOTUNE = False
B = 512
N.query('mpr_conv_set_wgrad_target_wgs', 160)
for name, H, C, K in [('l2 128->128 @28', 28, 128, 128)]:
    g = ops.ConvGeom((K, C, 3, 3), 1, 1)
    x = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
    dy = torch.randn(B, H, H, K, device='cuda').to(torch.bfloat16)
    for _ in range(3): ops.conv_wgrad(x, dy, g, (K, C, 3, 3))
    buf = torch.zeros(4096 * 8 * 4, dtype=torch.int64, device='cuda')
    N.lib().mpr_conv_debug_wgrad_probe(ctypes.c_void_p(buf.data_ptr()))
    N.query('mpr_conv_set_wgrad_window', 1 | (2 << 8))
    ops.conv_wgrad(x, dy, g, (K, C, 3, 3))
    torch.cuda.synchronize()
    N.query('mpr_conv_set_wgrad_window', 1)
    N.lib().mpr_conv_debug_wgrad_probe(None)
    t = buf.view(-1, 16, 8).cpu().double()
    t = t[t[:, 0, 5] > 0]
    n = t[0, 0, 5].item()
    print(name, len(t), 'WGs', n, 'chunks')
    for w in range(12):
        print(f'  wave {w:2d} (SIMD {w%4}): per chunk wait {t[:,w,1].mean()/n:6.0f} barrier {t[:,w,2].mean()/n:6.0f} issue {t[:,w,3].mean()/n:6.0f} compute {t[:,w,4].mean()/n:6.0f}')
