"""ResNet stem at the bench shape (batch 512, 224x224): the fused / recomputed kernels (csrc/stem_fused.hip) kernel by
kernel, and the whole stem (forward + backward) fused vs unfused."""
import sys, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
from multimodal_plankton_recognition_amd.image_encoder import ResNetBackbone
from multimodal_plankton_recognition_amd.layers import StemFn
B, H, W = 512, 224, 224
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
m = ResNetBackbone((1, 1, 1, 1), 1).cuda().train()
x = ((torch.randn(B, H, W, 1, device='cuda') * 0.0938 + 0.6136).clamp(0, 1) * 2 - 1)
dout = torch.randn(B, H // 4, W // 4, 64, device='cuda').to(torch.bfloat16)
xb = torch.empty(B, H + 6, W + 8, dtype=torch.bfloat16, device='cuda')
wp = torch.empty(64, 64, dtype=torch.bfloat16, device='cuda')
print(f'prep            {timeit(lambda: N.call("mpr_stemf_prep", x, m.conv1.weight.detach(), xb, wp, B, H, W)):7.1f} us')
stats = torch.zeros(8, 2, 64, device='cuda')
print(f'pass A (stats)  {timeit(lambda: N.call("mpr_stemf_stats", xb, wp, stats, 8, 0, B, H, W)):7.1f} us')
pooled, st, (xb, wp, idx) = ops.stemf_forward(x, m.conv1.weight, m.bn1, True, True)
cnt = B * (H // 2) * (W // 2)
bn = m.bn1
print(f'pass B (pool)   {timeit(lambda: N.call("mpr_stemf_pool", xb, wp, stats, 8, cnt, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, 0.1, 1e-5, st.scale, st.shift, st.mean, st.invstd, pooled, idx, B, H, W)):7.1f} us')
parts = N.query('mpr_stemf_bwd_parts', B, H, W)
partial = torch.empty(parts, 7168, device='cuda')
scratch = torch.empty(7168, dtype=torch.float64, device='cuda')
print(f'backward        {timeit(lambda: N.call("mpr_stemf_bwd", xb, dout, idx, partial, B, H, W)):7.1f} us  ({parts} partial blocks)')
dw = torch.empty(64, 1, 7, 7, device='cuda'); dg = torch.empty(64, device='cuda'); db = torch.empty(64, device='cuda')
print(f'bwd finalize    {timeit(lambda: N.call("mpr_stemf_bwd_finalize", partial, parts, scratch, wp, cnt, bn.weight.detach(), st.mean, st.invstd, 0.0, 0, dw, 0, dg, db, 0)):7.1f} us')
for fused in (True, False):
    ops.STEM_FUSED = fused
    def step():
        out = StemFn.apply(x, m.conv1.weight, m.bn1.weight, m.bn1.bias, m)
        out.backward(dout)
    print(f'whole stem fwd+bwd, fused={fused}: {timeit(step):8.1f} us')
ops.STEM_FUSED = True
