"""Micro-benchmark of the BatchNorm chain at the ResNet-18 activation shapes (batch 512): GB/s against algorithmic bytes."""
import sys, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
from multimodal_plankton_recognition_amd.layers import BatchNormParams
B = 512
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for H, C in [(56, 64), (28, 128), (14, 256), (7, 512)]:
    bn = BatchNormParams(C).cuda().train()
    x = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
    res = torch.randn_like(x)
    dy = torch.randn_like(x)
    nb = x.numel() * 2

    rows = x.numel() // C
    parts = torch.empty(N.query('mpr_bn_reduce_rows', rows, C), 2, C, dtype=torch.float32, device='cuda')
    t_stats = timeit(lambda: N.call('mpr_bn_stats', x, parts, rows, C))

    print(f'{H}x{H}x{C}: tensor {nb/1e6:.0f} MB | bn_stats {t_stats:6.1f}us {nb/t_stats/1e3:5.0f} GB/s | partial rows {parts.shape[0]}')
    scale = torch.ones(C, device='cuda'); shift = torch.zeros(C, device='cuda')
    y = torch.empty_like(x)
    t = timeit(lambda: N.call('mpr_bn_apply', x, scale, shift, None, 1, y, rows, C))
    print(f'    bn_apply relu          {t:6.1f}us {2*nb/t/1e3:5.0f} GB/s')
    t = timeit(lambda: N.call('mpr_bn_apply', x, scale, shift, res, 1, y, rows, C))
    print(f'    bn_apply relu+res      {t:6.1f}us {3*nb/t/1e3:5.0f} GB/s')
    mean = torch.zeros(C, device='cuda'); invstd = torch.ones(C, device='cuda')
    for mode, nread in ((0, 2), (1, 3), (2, 2)):
        t = timeit(lambda: N.call('mpr_bn_bwd_reduce', dy, y, x, mean, invstd, scale, shift, mode, parts, rows, C))
        print(f'    bn_bwd_reduce mode {mode}   {t:6.1f}us {nread*nb/t/1e3:5.0f} GB/s')
    coef = torch.empty(3, C, device='cuda'); dg = torch.empty(C, device='cuda'); db = torch.empty(C, device='cuda')
    for npart in (parts.shape[0], 64):
        t = timeit(lambda: N.call('mpr_bn_bwd_finalize', parts, npart, rows, scale, mean, invstd, dg, db, 0, coef, C))
        print(f'    bn_bwd_finalize nparts {npart:4d} {t:6.1f}us')
    p64 = torch.empty(64, 2, C, device='cuda')
    t = timeit(lambda: N.call('mpr_bn_reduce_partials', parts, parts.shape[0], p64, 64, C))
    print(f'    bn_reduce_partials     {t:6.1f}us')
    dx = torch.empty_like(x)
    for mode, nread in ((0, 2), (1, 3), (2, 2)):
        t = timeit(lambda: N.call('mpr_bn_bwd_apply', dy, y, x, coef, scale, shift, mode, dx, None, rows, C))
        print(f'    bn_bwd_apply mode {mode}    {t:6.1f}us {(nread+1)*nb/t/1e3:5.0f} GB/s')
