"""conv_win_kernel as v_mfma_f32_16x16x32_bf16 (variant bit 10) against the 32x32x16 form: time and the largest difference,
forward / data gradient / fused data gradients, ResNet-18 body shapes at batch 512 (layer1 through conv_win_kernel: bit 8)."""
import sys, types, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
B = 512
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
BASE = 5 | 256 | 512
for rep in range(2):
    for name, H, C in [('l1 64 @56', 56, 64), ('l2 128 @28', 28, 128), ('l3 256 @14', 14, 256), ('l4 512 @7', 7, 512)]:
        g = ops.ConvGeom((C, C, 3, 3), 1, 1)
        w = torch.randn(C, C, 3, 3, device='cuda') * 0.05
        wf, wd = ops.packed_weights(w, g)
        dy = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
        x = torch.randn_like(dy); y = torch.randn_like(dy); add = torch.randn_like(dy)
        st = types.SimpleNamespace(mean=torch.zeros(C, device='cuda'), invstd=torch.ones(C, device='cuda'),
                                   scale=torch.ones(C, device='cuda'), shift=torch.zeros(C, device='cuda'))
        fns = {'fwd': lambda: ops.conv_fwd(x, wf, g, True)[0], 'fwd_stats': lambda: ops.conv_fwd(x, wf, g, True)[1],
               'dgrad': lambda: ops.conv_dgrad(dy, wd, g, dy.shape),
               'mode2': lambda: ops.conv_dgrad_bn(dy, wd, g, dy.shape, x, st, 2)[0],
               'mode2_sums': lambda: ops.conv_dgrad_bn(dy, wd, g, dy.shape, x, st, 2)[1],
               'mode1+add': lambda: ops.conv_dgrad_bn(dy, wd, g, dy.shape, x, st, 1, mask_y=y, add=add)[0],
               'mode1+add_sums': lambda: ops.conv_dgrad_bn(dy, wd, g, dy.shape, x, st, 1, mask_y=y, add=add)[1]}
        out, tm = {}, {}
        for v in (BASE, BASE | 1024):
            N.query('mpr_conv_set_window_variant', v)
            for k, f in fns.items():
                out[v, k] = f().float().clone()
                if not k.endswith('s'): tm[v, k] = timeit(f)
        N.query('mpr_conv_set_window_variant', 5 | 512)
        row = []
        for k in fns:
            a, b = out[BASE, k], out[BASE | 1024, k]
            err = float((a - b).abs().max() / a.abs().max())
            row.append(f'{k} ' + (f'{tm[BASE, k]:6.1f} -> {tm[BASE | 1024, k]:6.1f} us ' if (BASE, k) in tm else '') + f'd {err:.1e}')
        print(f'{name}: ' + ' | '.join(row), flush=True)
