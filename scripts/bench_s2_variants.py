"""Ring depth / chunk size variants of the LDS-DMA kernel on the stride-2 data gradients (parity classes) and forwards."""
import sys, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
B = 512
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for name, H, C in [('l2.0 64->128 @56', 56, 64), ('l3.0 128->256 @28', 28, 128), ('l4.0 256->512 @14', 14, 256)]:
    K = 2 * C
    g3 = ops.ConvGeom((K, C, 3, 3), 2, 1)
    w3 = torch.randn(K, C, 3, 3, device='cuda') * 0.05
    wf3, wd3 = ops.packed_weights(w3, g3)
    x = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
    dy = torch.randn(B, H // 2, H // 2, K, device='cuda').to(torch.bfloat16)
    ref = None
    for nv, wv in [(0, 1), (1, 0), (3, 2), (4, 3), (2, 4), (0, 5), (0, 6), (0, 1)]:
        N.query('mpr_conv_set_variant', nv, wv)
        f3 = timeit(lambda: ops.conv_fwd(x, wf3, g3, True))
        d3 = timeit(lambda: ops.conv_dgrad(dy, wd3, g3, x.shape))
        r = ops.conv_dgrad(dy, wd3, g3, x.shape)
        ref = r if ref is None else ref
        print(f'{name} narrow {nv} wide {wv}: fwd {f3:6.1f}  dgrad {d3:6.1f}  {"same" if torch.equal(r, ref) else "DIFF"}', flush=True)
N.query('mpr_conv_set_variant', 0, 7)
