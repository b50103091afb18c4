"""Fused attention kernels alone at ViT-B/16's size (128 images x 12 heads, 197 tokens, head size 64) and at the profile
transformer's (225 tokens): forward, and backward (dQ kernel + dK/dV kernel)."""
import sys, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import transformer_mixed as TM
torch.manual_seed(0)
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for B, T, heads, hd in [(128, 197, 12, 64), (128, 225, 12, 64), (64, 257, 2, 32)]:
    d = heads * hd
    qkv = (torch.randn(B * T, 3 * d, device='cuda') * 0.5).to(torch.bfloat16)
    bias = torch.randn(3 * d, device='cuda') * 0.1
    out, lse = TM.attn_fwd(qkv, bias, None, B, T, heads, 0.0, 0)
    dout = torch.randn_like(out)
    tf = timeit(lambda: TM.attn_fwd(qkv, bias, None, B, T, heads, 0.0, 0))
    tb = timeit(lambda: TM.attn_bwd(qkv, bias, None, out, dout, lse, B, T, heads, 0.0, 0))
    print(f'B={B} T={T} heads={heads} hd={hd}: forward {tf:6.1f} us   backward (dQ + dK/dV) {tb:6.1f} us   checksums {out.float().abs().sum().item():.8e} {TM.attn_bwd(qkv, bias, None, out, dout, lse, B, T, heads, 0.0, 0).float().abs().sum().item():.8e}', flush=True)
