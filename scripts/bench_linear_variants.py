"""Tile / ring variants of the LDS-DMA implicit-GEMM kernel on the transformer linears (1x1 convs), forward and data gradient."""
import sys, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
variants = [int(v) for v in sys.argv[1:]] or [1, 0, 2, 5, 6]
for name, Bt, T, C, K in [('vit qkv', 128, 197, 768, 2304), ('vit proj', 128, 197, 768, 768), ('vit fc1', 128, 197, 768, 3072),
                          ('vit fc2', 128, 197, 3072, 768), ('prof fc1', 128, 225, 256, 1024)]:
    g = ops.ConvGeom((K, C, 1), 1, 0)
    w = torch.randn(K, C, 1, device='cuda') * 0.02
    wf, wd = ops.packed_weights(w, g)
    x = torch.randn(Bt, T, C, device='cuda').to(torch.bfloat16)
    dy = torch.randn(Bt, T, K, device='cuda').to(torch.bfloat16)
    flop = 2.0 * Bt * T * K * C
    row = [f'{name:9s}']
    ref = None
    for v in variants:
        N.query('mpr_conv_set_variant', 0, v)
        y = ops.conv_fwd(x, wf, g, False)[0]
        ok = ref is None or torch.equal(y, ref)
        ref = y if ref is None else ref
        tf = timeit(lambda: ops.conv_fwd(x, wf, g, False))
        td = timeit(lambda: ops.conv_dgrad(dy, wd, g, x.shape))
        row.append(f'v{v} fwd {tf:6.1f}us {flop/tf/1e6:4.0f}TF dgrad {td:6.1f}us {flop/td/1e6:4.0f}TF{"" if ok else " MISMATCH"}')
    print(' | '.join(row), flush=True)
N.query('mpr_conv_set_variant', 0, 7)

print('weight gradient, tile knob 0..3 (us):')
for name, Bt, T, C, K in [('vit qkv', 128, 197, 768, 2304), ('vit proj', 128, 197, 768, 768), ('vit fc1', 128, 197, 768, 3072),
                          ('vit fc2', 128, 197, 3072, 768)]:
    g = ops.ConvGeom((K, C, 1), 1, 0)
    x = torch.randn(Bt, T, C, device='cuda').to(torch.bfloat16)
    dy = torch.randn(Bt, T, K, device='cuda').to(torch.bfloat16)
    flop = 2.0 * Bt * T * K * C
    row, ref = [f'{name:9s}'], None
    for v in (0, 1, 2, 3):
        N.query('mpr_conv_set_wgrad_tile', v)
        dw = ops.conv_wgrad(x, dy, g, (K, C, 1))
        err = 0.0 if ref is None else float((dw - ref).abs().max() / ref.abs().max())
        ref = dw if ref is None else ref
        tw = timeit(lambda: ops.conv_wgrad(x, dy, g, (K, C, 1)))
        row.append(f'tile{v} {tw:6.1f}us {flop/tw/1e6:4.0f}TF (rel diff {err:.1e})')
    print(' | '.join(row), flush=True)
N.query('mpr_conv_set_wgrad_tile', 0)
