"""64 -> 64 channel 3x3 convolutions (ResNet layer1): the filter-in-registers kernel (conv_win_l1_kernel, round 3) against the
persistent / plain window kernels (mpr_conv_set_window_variant bit 8 switches the new kernel off): outputs and BatchNorm
partial sums compared, times at batch 512."""
import sys, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
base = N.query('mpr_conv_set_window_variant', 5)
N.query('mpr_conv_set_window_variant', base)
for B, H, W, C, K in [(512, 56, 56, 64, 64), (37, 19, 23, 64, 64), (9, 56, 40, 64, 64), (340, 7, 7, 64, 64), (16, 33, 12, 64, 40), (3, 56, 56, 64, 64)]:
    g = ops.ConvGeom((K, C, 3, 3), 1, 1)
    torch.manual_seed(B)
    w = torch.randn(K, C, 3, 3, device='cuda') * 0.05
    wf, wd = ops.packed_weights(w, g)
    x = torch.randn(B, H, W, C, device='cuda').to(torch.bfloat16)
    dy = torch.randn(B, H, W, K, device='cuda').to(torch.bfloat16)
    res = {}
    for name, var in [('l1', base & ~256), ('old', base | 256)]:
        N.query('mpr_conv_set_window_variant', var)
        y, st = ops.conv_fwd(x, wf, g, True)
        ssum = st.double().sum(0) if st is not None else None
        dx = ops.conv_dgrad(dy, wd, g, tuple(x.shape))
        tf = t(lambda: ops.conv_fwd(x, wf, g, True)) if B >= 256 else 0.
        td = t(lambda: ops.conv_dgrad(dy, wd, g, tuple(x.shape))) if B >= 256 else 0.
        res[name] = (y, ssum, dx, tf, td)
    N.query('mpr_conv_set_window_variant', base)
    y0, s0, d0, _, _ = res['old']; y1, s1, d1, tf, td = res['l1']
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), padding=1).permute(0, 2, 3, 1)
    print(f'B={B} {H}x{W} {C}->{K}: fwd equal {bool((y0 == y1).all())} (max |diff| {float((y0.float() - y1.float()).abs().max()):.3e}, vs fp32 torch {float((y1.float() - ref).abs().max() / ref.abs().max()):.2e}); '
          f'stats rel {float(((s0 - s1).abs() / (s0.abs() + 1e-3)).max()):.2e}; dgrad equal {bool((d0 == d1).all())} | '
          f'fwd {res["old"][3]:.1f} -> {tf:.1f} us, dgrad {res["old"][4]:.1f} -> {td:.1f} us', flush=True)
