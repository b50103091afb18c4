"""Per-chunk cycles of the sliding-window weight gradient under its debug bits (argv: dbg values), one shape, both MFMA shapes."""
import sys, ctypes, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
ops.AUTOTUNE = False
B = 512
N.query('mpr_conv_set_wgrad_target_wgs', 256)
for name, H, C, K in [('l1 64->64 @56', 56, 64, 64), ('l3 256->256 @14', 14, 256, 256)]:
    g = ops.ConvGeom((K, C, 3, 3), 1, 1)
    x = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
    dy = torch.randn(B, H, H, K, device='cuda').to(torch.bfloat16)
    for dbg in [int(a) for a in sys.argv[1:]] or [0]:
        for form, extra in (('16x16x32', 0), ('32x32x16', 64)):
            N.query('mpr_conv_set_wgrad_window', 1 | ((dbg | extra) << 8))
            for _ in range(3): ops.conv_wgrad(x, dy, g, (K, C, 3, 3))
            buf = torch.zeros(4096 * 8, dtype=torch.int64, device='cuda')
            N.lib().mpr_conv_debug_wgrad_probe(ctypes.c_void_p(buf.data_ptr()))
            ops.conv_wgrad(x, dy, g, (K, C, 3, 3))
            torch.cuda.synchronize()
            N.lib().mpr_conv_debug_wgrad_probe(None)
            t = buf.view(-1, 8).cpu().double()
            t = t[t[:, 5] > 0]
            n = t[:, 5].mean().item()
            print(f'{name} dbg {dbg:3d} {form}: per chunk: dma-wait {t[:,1].mean()/n:6.0f}  barrier {t[:,2].mean()/n:6.0f}  compute {t[:,4].mean()/n:6.0f}  sum {t[:,0].mean()/n:6.0f}', flush=True)
    N.query('mpr_conv_set_wgrad_window', 1)
