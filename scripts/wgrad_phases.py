"""Per-workgroup phase times of the LDS-DMA weight-gradient kernel (s_memrealtime stamps)."""
import sys, ctypes, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
B = 512
for name, H, C, K in [('l1 64->64 @56', 56, 64, 64), ('l2 128->128 @28', 28, 128, 128), ('l3 256->256 @14', 14, 256, 256), ('l4 512->512 @7', 7, 512, 512)]:
    g = ops.ConvGeom((K, C, 3, 3), 1, 1)
    x = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
    dy = torch.randn(B, H, H, K, device='cuda').to(torch.bfloat16)
    for _ in range(3): ops.conv_wgrad(x, dy, g, (K, C, 3, 3))
    buf = torch.zeros(16384 * 4, dtype=torch.int64, device='cuda')
    N.lib().mpr_conv_debug_wgrad_stamps(ctypes.c_void_p(buf.data_ptr()))
    ops.conv_wgrad(x, dy, g, (K, C, 3, 3))
    torch.cuda.synchronize()
    N.lib().mpr_conv_debug_wgrad_stamps(None)
    t = buf.view(-1, 4).cpu()
    t = t[t[:, 0] > 0].double() / 100.0
    t0 = t[:, 0].min()
    pro, main, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    nch = (B * H * H + 63) // 64
    print(f'{name}: {len(t)} WGs, span {t[:, 3].max() - t0:.1f} us | prologue {pro.mean():.2f} | main {main.mean():.2f} (max {main.max():.1f}) | '
          f'epilogue (atomics) {epi.mean():.2f} (max {epi.max():.1f}) | WG life {(t[:,3]-t[:,0]).mean():.2f} | mean alive {(t[:,3]-t[:,0]).sum()/(t[:,3].max()-t0):.0f}')
