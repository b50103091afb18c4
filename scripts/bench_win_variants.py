"""Window-kernel tile / ring variants per ResNet-18 body shape (batch 512, alone): forward and plain data gradient."""
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
for name, H, C in [('l2 128 @28', 28, 128), ('l3 256 @14', 14, 256), ('l4 512 @7', 7, 512)]:
    g = ops.ConvGeom((C, C, 3, 3), 1, 1)
    w = torch.randn(C, C, 3, 3, device='cuda') * 0.05
    wf, wd = ops.packed_weights(w, g)
    x = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
    out = []
    for v in (5, 0, 1, 2, 3, 4, 6, 5):
        N.query('mpr_conv_set_window_variant', v)
        out.append(f'v{v}: {timeit(lambda: ops.conv_fwd(x, wf, g, True)):5.1f}/{timeit(lambda: ops.conv_dgrad(x, wd, g, x.shape)):5.1f}')
    print(name, ' | '.join(out), flush=True)
N.query('mpr_conv_set_window_variant', 5)
