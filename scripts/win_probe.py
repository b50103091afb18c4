"""Where do the cycles of the shifted-window kernel's main loop go (wave 0 of every workgroup, shader clock)?"""
import sys, ctypes, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
B = 512
variants = [int(v) for v in sys.argv[1:]] or [4, 0]
for name, H, C, K in [('l1 64->64 @56', 56, 64, 64), ('l2 128->128 @28', 28, 128, 128), ('l3 256->256 @14', 14, 256, 256), ('l4 512->512 @7', 7, 512, 512)]:
    g = ops.ConvGeom((K, C, 3, 3), 1, 1)
    w = torch.randn(K, C, 3, 3, device='cuda') * 0.05
    wf, wd = ops.packed_weights(w, g)
    x = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
    for v in variants:
        N.query('mpr_conv_set_window_variant', v)
        for _ in range(3): ops.conv_fwd(x, wf, g, True)
        buf = torch.zeros(16384 * 16, dtype=torch.int64, device='cuda')
        N.lib().mpr_conv_debug_probe(ctypes.c_void_p(buf.data_ptr()))
        ops.conv_fwd(x, wf, g, True)
        torch.cuda.synchronize()
        N.lib().mpr_conv_debug_probe(None)
        t = buf.view(-1, 16).cpu().double()
        t = t[t[:, 0] > 0]
        nk = 9 * C // 64
        tot, wait, bar, comp, epi = [t[:, i].mean().item() for i in range(5)]
        print(f'{name} v{v}: {len(t)} WGs x {nk} chunks | WG total {tot:8.0f} cyc | per chunk: dma-wait {wait/nk:6.0f}  barrier {bar/nk:6.0f}  '
              f'compute {comp/nk:6.0f} | prologue {tot-wait-bar-comp-epi:6.0f} (reload barrier {t[:,6].mean():.0f} issue {t[:,7].mean():.0f}) | epilogue {epi:6.0f} (stage {t[:,8].mean():.0f} read+store {t[:,9].mean():.0f})', flush=True)
N.query('mpr_conv_set_window_variant', 0)
