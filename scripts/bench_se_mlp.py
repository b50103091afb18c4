"""mpr_se_mlp_fwd / _bwd alone at EfficientNet-B0's widths."""
import sys, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import _native as N
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for B in (256, 64):
    for C, rd in ((1152, 48), (672, 28), (240, 10), (96, 4)):
        dev = 'cuda'
        pooled = torch.randn(B, C, device=dev); w1 = torch.randn(rd, C, device=dev) * 0.1; b1 = torch.randn(rd, device=dev)
        w2 = torch.randn(C, rd, device=dev) * 0.1; b2 = torch.randn(C, device=dev)
        z1 = torch.empty(B, rd, device=dev); r = torch.empty_like(z1); gate = torch.empty(B, C, device=dev)
        t = timeit(lambda: N.call('mpr_se_mlp_fwd', pooled, w1, b1, w2, b2, z1, r, gate, B, C, rd))
        ref = torch.sigmoid(torch.nn.functional.silu(pooled @ w1.t() + b1) @ w2.t() + b2)
        print(f'B={B} C={C} rd={rd}: fwd {t:6.1f} us  err {float((gate - ref).abs().max()):.1e}', flush=True)
