"""Fused data-gradient epilogues of the 64 -> 64 convolutions on the filter-in-registers kernel against conv_win_kernel
(mpr_conv_set_window_variant bit 9 keeps them on the old kernel): skip add alone, BatchNorm-backward sums with either mask mode."""
import sys, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
ops.SLICE_ARENA = False
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
C = K = 64
for B, H, W in [(512, 56, 56), (150, 19, 23), (30, 56, 40), (1400, 7, 7), (11, 56, 56)]:
    g = ops.ConvGeom((K, C, 3, 3), 1, 1)
    torch.manual_seed(B)
    w = torch.randn(K, C, 3, 3, device='cuda') * 0.05
    wf, wd = ops.packed_weights(w, g)
    dy = torch.randn(B, H, W, K, device='cuda').to(torch.bfloat16)
    bn_x = torch.randn(B, H, W, C, device='cuda').to(torch.bfloat16)
    res = torch.randn(B, H, W, C, device='cuda').to(torch.bfloat16)
    gamma = torch.rand(C, device='cuda') + 0.5; beta = torch.randn(C, device='cuda') * 0.3
    y, st = ops.bn_apply(bn_x, ops.bn_stats_of(bn_x) if hasattr(ops, 'bn_stats_of') else None, gamma, beta) if False else (None, None)
    # BatchNorm state straight from the tensor
    xf = bn_x.float().reshape(-1, C)
    mean, var = xf.mean(0), xf.var(0, unbiased=False)
    st = ops.BNState() if hasattr(ops, 'BNState') and False else None
    class S: pass
    st = S(); st.mean = mean.contiguous(); st.invstd = (var + 1e-5).rsqrt().contiguous()
    st.scale = (gamma * st.invstd).contiguous(); st.shift = (beta - mean * st.scale).contiguous()
    mask_y = torch.relu(xf * st.scale + st.shift + res.float().reshape(-1, C)).to(torch.bfloat16).reshape(B, H, W, C)
    cases = [('add only', lambda: (ops.conv_dgrad(dy, wd, g, (B, H, W, C), add=res), None)),
             ('bnb mode 2', lambda: ops.conv_dgrad_bn(dy, wd, g, (B, H, W, C), bn_x, st, 2)),
             ('bnb mode 1', lambda: ops.conv_dgrad_bn(dy, wd, g, (B, H, W, C), bn_x, st, 1, mask_y=mask_y)),
             ('add + bnb 1', lambda: ops.conv_dgrad_bn(dy, wd, g, (B, H, W, C), bn_x, st, 1, mask_y=mask_y, add=res))]
    for name, fn in cases:
        out = {}
        for tag, var in (('l1', base & ~512), ('old', base | 512)):
            N.query('mpr_conv_set_window_variant', var)
            ops.reset_slice_arena() if hasattr(ops, 'reset_slice_arena') else None
            r = fn()
            if r is None:
                print(name, 'not served at', B, H, W); break
            dz, sl = r
            torch.cuda.synchronize()
            sums = sl.double().sum(0).clone() if sl is not None else None
            tm = t(lambda: fn()) if B >= 256 else 0.
            out[tag] = (dz.clone(), sums, tm)
        N.query('mpr_conv_set_window_variant', base)
        if len(out) < 2: continue
        (d1, s1, t1), (d0, s0, t0) = out['l1'], out['old']
        srel = float(((s1 - s0).abs() / (s0.abs() + 1e-2)).max()) if s1 is not None else 0.
        print(f'B={B} {H}x{W} {name:12s}: dz equal {bool(torch.equal(d1, d0))} (max |diff| {float((d1.float() - d0.float()).abs().max()):.2e}); sums rel {srel:.2e} | {t0:.1f} -> {t1:.1f} us', flush=True)
