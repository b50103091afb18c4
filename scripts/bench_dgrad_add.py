"""Data gradient WITH the fused skip-connection add (the block-input gradient of every BasicBlock) on the ResNet-18 body
shapes, batch 512, alone on the GPU."""
import sys, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops
B = 512
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for name, H, C in [('l1 64->64 @56', 56, 64), ('l2 128 @28', 28, 128), ('l3 256 @14', 14, 256), ('l4 512 @7', 7, 512)]:
    g = ops.ConvGeom((C, C, 3, 3), 1, 1)
    w = torch.randn(C, C, 3, 3, device='cuda') * 0.05
    wf, wd = ops.packed_weights(w, g)
    dy = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
    add = torch.randn_like(dy)
    flop = 2.0 * B * H * H * C * C * 9
    t0 = timeit(lambda: ops.conv_dgrad(dy, wd, g, dy.shape))
    t1 = timeit(lambda: ops.conv_dgrad(dy, wd, g, dy.shape, add=add))
    ref = ops.conv_dgrad(dy, wd, g, dy.shape, add=add)
    print(f'{name}: dgrad {t0:6.1f} us {flop/t0/1e6:5.0f} TF | dgrad + add {t1:6.1f} us {flop/t1/1e6:5.0f} TF | checksum {ref.float().sum().item():.6e} {ref.float().abs().sum().item():.6e}', flush=True)
