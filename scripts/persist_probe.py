"""Shader-clock breakdown of the persistent 256 x 64 window kernel (layer1 forward), wave 0 of every workgroup."""
import sys, ctypes, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
B, H, C = 512, 56, 64
g = ops.ConvGeom((C, C, 3, 3), 1, 1)
w = torch.randn(C, C, 3, 3, device='cuda') * 0.05
wf, wd = ops.packed_weights(w, g)
x = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
for _ in range(3): ops.conv_fwd(x, wf, g, True)
buf = torch.zeros(16384 * 16, dtype=torch.int64, device='cuda')
N.lib().mpr_conv_debug_probe(ctypes.c_void_p(buf.data_ptr()))
ops.conv_fwd(x, wf, g, True)
torch.cuda.synchronize()
N.lib().mpr_conv_debug_probe(None)
t = buf.view(-1, 16).cpu().double()
t = t[t[:, 5] > 0]
n = t[:, 5].mean().item()
tot, wait, bar, comp, epi = [t[:, i].mean().item() for i in range(5)]
print(f'{len(t)} WGs x {n:.1f} tiles | per tile: total {tot/n:7.0f} cyc = dma-wait {wait/n:6.0f} + barrier {bar/n:6.0f} + MFMA steps {comp/n:6.0f} + epilogue {epi/n:6.0f} + rest {(tot-wait-bar-comp-epi)/n:6.0f} | epilogue: stage {t[:,6].mean()/n:6.0f}  read {t[:,7].mean()/n:6.0f}  decode+store+stats {t[:,8].mean()/n:6.0f}')
