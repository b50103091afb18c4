"""The strided convolutions of ResNet-18's downsampling blocks (3x3 / stride 2 and the 1x1 / stride 2 shortcut), batch 512, alone:
forward, data gradient (parity classes, + the shortcut gradient on the half-resolution grid), weight gradient."""
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
for rep in range(2):
    for name, H, C in [('l2.0 64->128 @56', 56, 64), ('l3.0 128->256 @28', 28, 128), ('l4.0 256->512 @14', 14, 256)]:
        K = 2 * C
        g3, g1 = ops.ConvGeom((K, C, 3, 3), 2, 1), ops.ConvGeom((K, C, 1, 1), 2, 0)
        w3 = torch.randn(K, C, 3, 3, device='cuda') * 0.05
        w1 = torch.randn(K, C, 1, 1, device='cuda') * 0.05
        wf3, wd3 = ops.packed_weights(w3, g3)
        wf1, wd1 = ops.packed_weights(w1, g1)
        x = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
        dy = torch.randn(B, H // 2, H // 2, K, device='cuda').to(torch.bfloat16)
        f3 = timeit(lambda: ops.conv_fwd(x, wf3, g3, True))
        f1 = timeit(lambda: ops.conv_fwd(x, wf1, g1, True))
        d3 = timeit(lambda: ops.conv_dgrad(dy, wd3, g3, x.shape))
        ds = timeit(lambda: ops.conv_dgrad_shortcut(dy, wd3, g3, tuple(x.shape), dy, wd1, g1))
        wg3 = timeit(lambda: ops.conv_wgrad(x, dy, g3, (K, C, 3, 3)))
        wg1 = timeit(lambda: ops.conv_wgrad(x, dy, g1, (K, C, 1, 1)))
        fl = 2.0 * B * (H // 2) ** 2 * K * C * 9
        print(f'{name}: fwd3x3 {f3:6.1f} ({fl/f3/1e6:4.0f} TF)  fwd1x1 {f1:5.1f}  dgrad3x3 {d3:6.1f} ({fl/d3/1e6:4.0f} TF)  dgrad3x3+shortcut {ds:6.1f}  wgrad3x3 {wg3:6.1f} ({fl/wg3/1e6:4.0f} TF)  wgrad1x1 {wg1:5.1f}', flush=True)
