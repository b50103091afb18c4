"""Shifted-window conv kernel: variants vs the plain LDS-DMA implicit GEMM on the ResNet-18 body shapes (batch 512)."""
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
variants = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 3, 4]
for name, H, C, K in [('l1 64->64 @56', 56, 64, 64), ('l2 128->128 @28', 28, 128, 128), ('l3 256->256 @14', 14, 256, 256), ('l4 512->512 @7', 7, 512, 512)]:
    g = ops.ConvGeom((K, C, 3, 3), 1, 1)
    w = torch.randn(K, C, 3, 3, device='cuda') * 0.05
    wf, wd = ops.packed_weights(w, g)
    x = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
    dy = torch.randn(B, H, H, K, device='cuda').to(torch.bfloat16)
    flop = 2.0 * B * H * H * K * C * 9
    N.query('mpr_conv_set_window', 0)
    ref = ops.conv_fwd(x, wf, g, True)[0]
    refd = ops.conv_dgrad(dy, wd, g, x.shape)
    t0 = timeit(lambda: ops.conv_fwd(x, wf, g, True))
    row = [name, f'igemm {t0:6.1f}us {flop/t0/1e6:5.0f}TF']
    N.query('mpr_conv_set_window', 1)
    for v in variants:
        N.query('mpr_conv_set_window_variant', v)
        ok = torch.equal(ops.conv_fwd(x, wf, g, True)[0], ref) and torch.equal(ops.conv_dgrad(dy, wd, g, x.shape), refd)
        tf = timeit(lambda: ops.conv_fwd(x, wf, g, True))
        td = timeit(lambda: ops.conv_dgrad(dy, wd, g, x.shape))
        row.append(f'v{v} fwd {tf:6.1f}us {flop/tf/1e6:5.0f}TF dgrad {td:6.1f}us{"" if ok else " MISMATCH"}')
    N.query('mpr_conv_set_window_variant', 0)
    print(' | '.join(row), flush=True)
