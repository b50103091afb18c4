"""The exact-fp32 GEMM on the small products of the loss / projection stage (latency-bound: 64 workgroups)."""
import sys, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for name, M, N, K, ta, tb in [('image projection x W^T', 512, 512, 514, False, True), ('S = U V^T', 512, 512, 512, False, True),
                              ('G V', 512, 512, 512, False, False), ('dW = dy^T x', 512, 514, 512, True, False),
                              ('4096 x 4096 x 512 (DP global S)', 4096, 4096, 512, False, True)]:
    a = torch.randn((K, M) if ta else (M, K), device='cuda')
    b = torch.randn((N, K) if tb else (K, N), device='cuda')
    ref = (a.t() if ta else a).double() @ (b.t() if tb else b).double()
    out = ops.gemm(a, b, trans_a=ta, trans_b=tb)
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    t = timeit(lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb))
    print(f'{name:34s} {t:7.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TFLOP/s  (max rel err {err:.1e})', flush=True)
