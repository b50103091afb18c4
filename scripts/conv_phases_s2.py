"""Per-workgroup phase times (s_memrealtime stamps) of the LDS-DMA kernel on the stride-2 data gradients (parity classes)."""
import sys, ctypes, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
B = 512
for name, H, C in [('l2.0 dgrad 128->64 @56', 56, 64), ('l3.0 dgrad 256->128 @28', 28, 128), ('l4.0 dgrad 512->256 @14', 14, 256)]:
    K = 2 * C
    g = ops.ConvGeom((K, C, 3, 3), 2, 1)
    w = torch.randn(K, C, 3, 3, device='cuda') * 0.05
    wf, wd = ops.packed_weights(w, g)
    dy = torch.randn(B, H // 2, H // 2, K, device='cuda').to(torch.bfloat16)
    shape = (B, H, H, C)
    for _ in range(3): ops.conv_dgrad(dy, wd, g, shape)
    nwg = 16384
    buf = torch.zeros(nwg * 4, dtype=torch.int64, device='cuda')
    N.lib().mpr_conv_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
    ops.conv_dgrad(dy, wd, g, shape)
    torch.cuda.synchronize()
    N.lib().mpr_conv_debug_stamps(None)
    t = buf.view(-1, 4).cpu()
    t = t[t[:, 0] > 0].double() / 100.0
    t0 = t[:, 0].min()
    pro, main, epi = (t[:, 1] - t[:, 0]), (t[:, 2] - t[:, 1]), (t[:, 3] - t[:, 2])
    n = len(t)
    q = n // 4
    print(f'{name}: {n} WGs, span {t[:, 3].max() - t0:.1f} us | prologue {pro.mean():.2f} | main {main.mean():.2f} (min {main.min():.1f} max {main.max():.1f}) | '
          f'epilogue {epi.mean():.2f} (max {epi.max():.1f}) | WG life {(t[:,3]-t[:,0]).mean():.2f} us | alive {(t[:, 3] - t[:, 0]).sum() / (t[:, 3].max() - t0):.0f}')
    for c in range(4):
        s = slice(c * q, (c + 1) * q)
        print(f'      class {c}: main {main[s].mean():.2f}  epilogue {epi[s].mean():.2f}  life {(t[s,3]-t[s,0]).mean():.2f} us')
