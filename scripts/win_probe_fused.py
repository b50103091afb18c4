"""Cycle breakdown (wave 0 of every workgroup, shader clock) of conv_win_kernel's fused data gradients on the layer1 / layer2
shapes: plain, ReLU-recompute + BatchNorm sums (mode 2), mask from the block output + skip add (mode 1 + add)."""
import sys, ctypes, types, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
B = 512
base = N.query('mpr_conv_set_window_variant', 5)
N.query('mpr_conv_set_window_variant', base | 256)       # layer1 on conv_win_kernel as well
for name, H, C in [('l1 64 @56', 56, 64), ('l2 128 @28', 28, 128)]:
    g = ops.ConvGeom((C, C, 3, 3), 1, 1)
    w = torch.randn(C, C, 3, 3, device='cuda') * 0.05
    wf, wd = ops.packed_weights(w, g)
    dy = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
    x = torch.randn_like(dy); y = torch.randn_like(dy); add = torch.randn_like(dy)
    st = types.SimpleNamespace(mean=torch.zeros(C, device='cuda'), invstd=torch.ones(C, device='cuda'),
                               scale=torch.ones(C, device='cuda'), shift=torch.zeros(C, device='cuda'))
    fns = {'plain': lambda: ops.conv_dgrad(dy, wd, g, dy.shape), 'mode2': lambda: ops.conv_dgrad_bn(dy, wd, g, dy.shape, x, st, 2),
           'mode1+add': lambda: ops.conv_dgrad_bn(dy, wd, g, dy.shape, x, st, 1, mask_y=y, add=add)}
    for k, f in fns.items():
        for _ in range(3): f()
        buf = torch.zeros(16384 * 16, dtype=torch.int64, device='cuda')
        N.lib().mpr_conv_debug_probe(ctypes.c_void_p(buf.data_ptr()))
        f()
        torch.cuda.synchronize()
        N.lib().mpr_conv_debug_probe(None)
        t = buf.view(-1, 16).cpu().double()
        t = t[t[:, 0] > 0]
        nk = 9 * C // 64
        tot, wait, bar, comp, epi = [t[:, i].mean().item() for i in range(5)]
        print(f'{name} {k:10s}: {len(t)} WGs x {nk} chunks | WG total {tot:8.0f} cyc | loop: dma-wait {wait:6.0f} barrier {bar:6.0f} compute {comp:6.0f} | '
              f'prologue {tot-wait-bar-comp-epi:6.0f} | epilogue {epi:6.0f} (stage {t[:,8].mean():.0f} read+store {t[:,9].mean():.0f})', flush=True)
N.query('mpr_conv_set_window_variant', base)
