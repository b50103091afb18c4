"""One weight-gradient shape a few times (for rocprofv3 counter passes).  argv: H C [target] [mode]"""
import sys, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
ops.AUTOTUNE = False
H, C = int(sys.argv[1]), int(sys.argv[2])
N.query('mpr_conv_set_wgrad_target_wgs', int(sys.argv[3]) if len(sys.argv) > 3 else 256)
N.query('mpr_conv_set_wgrad_window', int(sys.argv[4]) if len(sys.argv) > 4 else 1)
B, K = 512, C
g = ops.ConvGeom((K, C, 3, 3), 1, 1)
x = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
dy = torch.randn(B, H, H, K, device='cuda').to(torch.bfloat16)
for _ in range(5): ops.conv_wgrad(x, dy, g, (K, C, 3, 3))
torch.cuda.synchronize()
