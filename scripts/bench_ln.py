"""LayerNorm backward of the mixed path on ViT-B/16's residual stream (25216 x 768): rows per workgroup."""
import sys, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import transformer_mixed as TM, _native as N
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
rows, D = 25216, 768
x = torch.randn(rows, D, device='cuda')
g = torch.ones(D, device='cuda', requires_grad=True); b = torch.zeros(D, device='cuda', requires_grad=True)
s, _, y16, mean, rstd = TM.add_ln(x, None, None, 0.0, 0, g, b, 1e-6, True, False, True)
dy16 = torch.randn(rows, D, device='cuda').to(torch.bfloat16)
dskip = torch.randn(rows, D, device='cuda')
mb = (rows * D * (2 + 4 + 4 + 4)) / 1e6
for r in (32, 64, 16, 8, 4):
    N.query('mpr_tf_set_ln_bwd_rows', r)
    t = timeit(lambda: TM.ln_bwd(dy16, None, s, g, b, mean, rstd, dskip))
    print(f'rows per workgroup {r:3d}: {t:6.1f} us  ({mb / t:.2f} TB/s of algorithmic traffic)', flush=True)
N.query('mpr_tf_set_ln_bwd_rows', 32)
