"""Every tile / ring variant of the LDS-DMA conv kernel must give bit-identical results (same k order), then time them."""
import sys, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
N.query('mpr_conv_set_dma_min_rows', 0)
def run(B, H, C, K, R, st, pad, nv, wv):
    N.query('mpr_conv_set_variant', nv, wv)
    g = ops.ConvGeom((K, C, R, R), st, pad)
    torch.manual_seed(0)
    w = torch.randn(K, C, R, R, device='cuda') * 0.05
    wf, wd = ops.packed_weights(w, g)
    x = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
    y, stats = ops.conv_fwd(x, wf, g, True)
    dy = torch.randn_like(y)
    add = torch.randn_like(x)
    dx = ops.conv_dgrad(dy, wd, g, x.shape, add=add)
    return y, stats.sum(0), dx
ok = True
for shape in [(3, 12, 64, 64, 3, 1, 1), (2, 14, 128, 256, 3, 1, 1), (5, 9, 256, 128, 3, 1, 1), (4, 16, 64, 128, 3, 2, 1),
              (2, 10, 128, 256, 1, 2, 0), (3, 13, 64, 192, 3, 2, 1), (7, 7, 512, 512, 3, 1, 1), (6, 16, 256, 512, 3, 2, 1), (5, 12, 512, 256, 1, 1, 0)]:
    ref = run(*shape, 0, 1)
    for nv, wv in [(1, 0), (2, 2), (3, 3), (4, 4), (0, 5), (0, 6), (0, 7), (0, 8), (0, 9)]:
        got = run(*shape, nv, wv)
        same = [torch.equal(a, b) for a, b in zip(ref, got)]
        # stats rows are per row tile: their SUM over tiles may differ in the last bits when the tile height differs
        good = same[0] and same[2] and torch.allclose(ref[1], got[1], rtol=1e-5, atol=1e-3)
        ok &= good
        print(shape, (nv, wv), 'OK' if good else f'MISMATCH {same}', flush=True)
N.query('mpr_conv_set_variant', 0, 7)
print('ALL OK' if ok else 'FAILED')
sys.exit(0 if ok else 1)
