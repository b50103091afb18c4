"""Stride-2 data gradients of ResNet-18's downsampling blocks in the form the training step runs them: parity classes + the 1x1
shortcut's gradient on the half-resolution grid + the previous block's ReLU mask and bn2 backward sums (mpr_conv_dgrad_s2_bn);
and the plain / shortcut-only forms.  Checksums: compare builds."""
import sys, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops
ops.SLICE_ARENA = False
B = 512
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
class St: pass
for name, H, C in [('l2.0 64->128 @56', 56, 64), ('l3.0 128->256 @28', 28, 128), ('l4.0 256->512 @14', 14, 256)]:
    torch.manual_seed(H)
    K = 2 * C
    g3, g1 = ops.ConvGeom((K, C, 3, 3), 2, 1), ops.ConvGeom((K, C, 1, 1), 2, 0)
    w3 = torch.randn(K, C, 3, 3, device='cuda') * 0.05
    w1 = torch.randn(K, C, 1, 1, device='cuda') * 0.05
    _, wd3 = ops.packed_weights(w3, g3)
    _, wd1 = ops.packed_weights(w1, g1)
    dy = torch.randn(B, H // 2, H // 2, K, device='cuda').to(torch.bfloat16)
    bn_x = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
    mask_y = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
    xf = bn_x.float().reshape(-1, C)
    st = St(); st.mean = xf.mean(0).contiguous(); st.invstd = (xf.var(0, unbiased=False) + 1e-5).rsqrt().contiguous()
    st.scale = st.invstd.clone(); st.shift = (-st.mean * st.invstd).contiguous()
    shape = (B, H, H, C)
    plain = ops.conv_dgrad(dy, wd3, g3, shape)
    t0 = timeit(lambda: ops.conv_dgrad(dy, wd3, g3, shape))
    r1 = ops.conv_dgrad_shortcut(dy, wd3, g3, shape, dy, wd1, g1)
    t1 = timeit(lambda: ops.conv_dgrad_shortcut(dy, wd3, g3, shape, dy, wd1, g1))
    r2 = ops.conv_dgrad_shortcut(dy, wd3, g3, shape, dy, wd1, g1, note=(bn_x, st), mask_y=mask_y)
    t2 = timeit(lambda: ops.conv_dgrad_shortcut(dy, wd3, g3, shape, dy, wd1, g1, note=(bn_x, st), mask_y=mask_y))
    cs = lambda t: float(t.double().sum())
    print(f'{name}: plain {t0:6.1f} us ({cs(plain):.6e}) | + shortcut {t1:6.1f} us ({cs(r1[0]):.6e}) | + mask + bn2 sums {t2:6.1f} us '
          f'({cs(r2[0]):.6e}, sums {cs(r2[1]):.6e})', flush=True)
