"""What does each operand's L2->LDS stream cost the implicit-GEMM kernel?  (zero-record descriptors: timing only)"""
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
for name, H, C, K in [('l1 3x3 64->64', 56, 64, 64), ('l2 3x3 128->128', 28, 128, 128), ('l3 3x3 256->256', 14, 256, 256), ('l4 3x3 512->512', 7, 512, 512)]:
    g = ops.ConvGeom((K, C, 3, 3), 1, 1)
    w = torch.randn(K, C, 3, 3, device='cuda') * 0.05
    wf, wd = ops.packed_weights(w, g)
    x = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
    flop = 2.0 * B * H * H * K * C * 9
    row = [name]
    for mask, tag in ((0, 'both'), (3, 'zero-fill all'), (4, 'A not issued'), (8, 'B not issued'), (12, 'none issued'), (28, 'none issued, no epilogue'), (16, 'no epilogue')):
        N.query('mpr_conv_debug_drop_operand', mask)
        t = timeit(lambda: ops.conv_fwd(x, wf, g, True))
        row.append(f'{tag}: {t:6.1f}us {flop/t/1e6:5.0f}TF')
    N.query('mpr_conv_debug_drop_operand', 0)
    print(' | '.join(row))
