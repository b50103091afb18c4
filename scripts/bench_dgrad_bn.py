"""Data gradient with the fused ReLU mask + BatchNorm-backward sums (+ skip add) on the ResNet-18 body shapes, batch 512, alone:
mode 2 (mask recomputed from the BatchNorm input), mode 1 + add (mask from the block output, skip gradient added), against
the plain data gradient."""
import sys, types, torch
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
for rep in range(2):
    for name, H, C in [('l1 64 @56', 56, 64), ('l2 128 @28', 28, 128), ('l3 256 @14', 14, 256), ('l4 512 @7', 7, 512)]:
        g = ops.ConvGeom((C, C, 3, 3), 1, 1)
        w = torch.randn(C, C, 3, 3, device='cuda') * 0.05
        wf, wd = ops.packed_weights(w, g)
        dy = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
        x = torch.randn_like(dy); y = torch.randn_like(dy); add = torch.randn_like(dy)
        st = types.SimpleNamespace(mean=torch.zeros(C, device='cuda'), invstd=torch.ones(C, device='cuda'),
                                   scale=torch.ones(C, device='cuda'), shift=torch.zeros(C, device='cuda'))
        t0 = timeit(lambda: ops.conv_dgrad(dy, wd, g, dy.shape))
        t2 = timeit(lambda: ops.conv_dgrad_bn(dy, wd, g, dy.shape, x, st, 2))
        t1 = timeit(lambda: ops.conv_dgrad_bn(dy, wd, g, dy.shape, x, st, 1, mask_y=y, add=add))
        r2 = ops.conv_dgrad_bn(dy, wd, g, dy.shape, x, st, 2)
        r1 = ops.conv_dgrad_bn(dy, wd, g, dy.shape, x, st, 1, mask_y=y, add=add)
        print(f'{name}: plain {t0:6.1f}  mode2 {t2:6.1f}  mode1+add {t1:6.1f} us | checksums {r2[0].float().abs().sum().item():.6e} {r2[1].double().sum().item():.8e} '
              f'{r1[0].float().abs().sum().item():.6e} {r1[1].double().sum().item():.8e}', flush=True)
