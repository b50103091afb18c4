"""Per-workgroup phase times of the LDS-DMA conv kernel (s_memrealtime stamps): prologue / main loop / epilogue."""
import sys, ctypes, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
B = 512
for name, H, C, K in [('l1 3x3 64->64', 56, 64, 64), ('l2 3x3 128->128', 28, 128, 128), ('l3 3x3 256->256', 14, 256, 256), ('l4 3x3 512->512', 7, 512, 512)]:
    g = ops.ConvGeom((K, C, 3, 3), 1, 1)
    w = torch.randn(K, C, 3, 3, device='cuda') * 0.05
    wf, wd = ops.packed_weights(w, g)
    x = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
    for _ in range(3): ops.conv_fwd(x, wf, g, True)
    nwg = 8192
    buf = torch.zeros(nwg * 4, dtype=torch.int64, device='cuda')
    N.lib().mpr_conv_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
    ops.conv_fwd(x, wf, g, True)
    torch.cuda.synchronize()
    N.lib().mpr_conv_debug_stamps(None)
    t = buf.view(-1, 4).cpu()
    t = t[t[:, 0] > 0].double() / 100.0          # us
    t0 = t[:, 0].min()
    pro, main, epi = (t[:, 1] - t[:, 0]), (t[:, 2] - t[:, 1]), (t[:, 3] - t[:, 2])
    print(f'{name}: {len(t)} WGs, span {t[:, 3].max() - t0:.1f} us | prologue {pro.mean():.2f} (max {pro.max():.1f}) | main {main.mean():.2f} '
          f'(min {main.min():.1f} max {main.max():.1f}) | epilogue {epi.mean():.2f} (max {epi.max():.1f}) | WG life {(t[:,3]-t[:,0]).mean():.2f} us')
    starts = ((t[:, 0] - t0)).sort().values
    # concurrency: how many WGs alive on average
    alive = (t[:, 3] - t[:, 0]).sum() / (t[:, 3].max() - t0)
    print(f'      mean WGs alive {alive:.0f} (of 512 slots); last WG starts at {starts[-1]:.1f} us; starts quartiles {[round(float(starts[int(len(starts)*q)]),1) for q in (0.25,0.5,0.75)]}')
