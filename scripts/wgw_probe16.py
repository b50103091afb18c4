"""Shader-clock breakdown of the sliding-window weight-gradient kernel's main loop (wave 0 of every workgroup), both MFMA shapes."""
import sys, ctypes, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
ops.AUTOTUNE = False
B = 512
for tgt in (160, 256):
  N.query('mpr_conv_set_wgrad_target_wgs', tgt)
  for name, H, C, K in [('l1 64->64 @56', 56, 64, 64), ('l2 128->128 @28', 28, 128, 128), ('l3 256->256 @14', 14, 256, 256), ('l4 512->512 @7', 7, 512, 512)]:
    g = ops.ConvGeom((K, C, 3, 3), 1, 1)
    x = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
    dy = torch.randn(B, H, H, K, device='cuda').to(torch.bfloat16)
    for form, mode in (('16x16x32', 1), ('32x32x16', 1 | (64 << 8))):
        N.query('mpr_conv_set_wgrad_window', mode)
        for _ in range(3): ops.conv_wgrad(x, dy, g, (K, C, 3, 3))
        buf = torch.zeros(4096 * 8, dtype=torch.int64, device='cuda')
        N.lib().mpr_conv_debug_wgrad_probe(ctypes.c_void_p(buf.data_ptr()))
        ops.conv_wgrad(x, dy, g, (K, C, 3, 3))
        torch.cuda.synchronize()
        N.lib().mpr_conv_debug_wgrad_probe(None)
        t = buf.view(-1, 8).cpu().double()
        t = t[t[:, 5] > 0]
        n = t[:, 5].mean().item()
        print(f'target {tgt} {name} {form}: {len(t)} WGs x {n:.0f} chunks | loop total {t[:,0].mean():8.0f} cyc | per chunk: dma-wait {t[:,1].mean()/n:6.0f}  barrier {t[:,2].mean()/n:6.0f}  '
              f'compute {t[:,4].mean()/n:6.0f}  sum {t[:,0].mean()/n:6.0f}', flush=True)
    N.query('mpr_conv_set_wgrad_window', 1)
