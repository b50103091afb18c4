"""Transformer linears on the 1x1 implicit-GEMM conv kernels (bf16 operands, fp32 accumulate): ViT-B/16 shapes at 128
images per GPU (M = 128 x 197 tokens) and the ProfileTransformer of the C5 card."""
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
for name, Bt, T, C, K in [('vit qkv', 128, 197, 768, 2304), ('vit proj', 128, 197, 768, 768), ('vit fc1', 128, 197, 768, 3072),
                          ('vit fc2', 128, 197, 3072, 768), ('prof qkv', 128, 225, 256, 768), ('prof fc1', 128, 225, 256, 1024)]:
    g = ops.ConvGeom((K, C, 1), 1, 0)
    w = torch.randn(K, C, 1, device='cuda') * 0.02
    wf, wd = ops.packed_weights(w, g)
    x = torch.randn(Bt, T, C, device='cuda').to(torch.bfloat16)
    dy = torch.randn(Bt, T, K, device='cuda').to(torch.bfloat16)
    flop = 2.0 * Bt * T * K * C
    tf = timeit(lambda: ops.conv_fwd(x, wf, g, False))
    td = timeit(lambda: ops.conv_dgrad(dy, wd, g, x.shape))
    tw = timeit(lambda: ops.conv_wgrad(x, dy, g, (K, C, 1)))
    print(f'{name:9s} M={Bt*T} C={C} K={K}: fwd {tf:6.1f}us {flop/tf/1e6:5.0f}TF | dgrad {td:6.1f}us {flop/td/1e6:5.0f}TF | wgrad {tw:6.1f}us {flop/tw/1e6:5.0f}TF', flush=True)
