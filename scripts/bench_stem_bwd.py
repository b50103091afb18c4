"""Stem backward (max-pool gradient gathered inside the two BatchNorm-backward passes) at the C3 shape."""
import sys, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
from multimodal_plankton_recognition_amd.layers import BatchNormParams
B, H, C = 512, 112, 64
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
bn = BatchNormParams(C).cuda().train()
y = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
st = ops.bn_coefs(None, y.numel() // C, bn, True, y)
pooled, idx = ops.bn_relu_maxpool_fwd(y, st)
dp = torch.randn_like(pooled)
t = timeit(lambda: ops.pool_bn_bwd(dp, idx, y, bn.weight, st))
nb = y.numel() * 2
print(f'pool_bn_bwd total {t:7.1f} us; algorithmic bytes (2 x read y, dpooled, idx; write dx) {(3*nb + 2*(dp.numel()*2 + idx.numel()))/1e6:.0f} MB -> {(3*nb + 2*(dp.numel()*2 + idx.numel()))/t/1e3:.0f} GB/s')
tf = timeit(lambda: ops.bn_relu_maxpool_fwd(y, st))
print(f'bn_relu_maxpool_fwd {tf:7.1f} us -> {(nb + dp.numel()*2 + idx.numel())/tf/1e3:.0f} GB/s')
