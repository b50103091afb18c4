"""Window conv kernels (forward, data gradient, data gradient with fused BatchNorm backward) on the ResNet-18 body shapes,
A/B of two builds of the library in child processes (MPR_HIP_LIB): bench_win_ab.py <lib A> <lib B>; alternating, 2 rounds."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from multimodal_plankton_recognition_amd import ops
B = 512
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
out = []
for name, H, C in [('l2', 28, 128), ('l3', 14, 256), ('l4', 7, 512)]:
    torch.manual_seed(0)
    g = ops.ConvGeom((C, C, 3, 3), 1, 1)
    w = torch.randn(C, C, 3, 3, device='cuda') * 0.05
    wf, wd = ops.packed_weights(w, g)
    x = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
    dy = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
    y = ops.conv_fwd(x, wf, g, True)[0]
    d = ops.conv_dgrad(dy, wd, g, x.shape)
    out.append('%%s fwd %%.1f dgrad %%.1f (sums %%.4e %%.4e)' %% (name, t(lambda: ops.conv_fwd(x, wf, g, True)), t(lambda: ops.conv_dgrad(dy, wd, g, x.shape)), float(y.float().sum()), float(d.float().sum())))
print(' | '.join(out))
''' % ROOT
for rep in range(2):
    for lib in sys.argv[1:]:
        r = subprocess.run([sys.executable, '-c', CHILD], env=dict(os.environ, MPR_HIP_LIB=os.path.join(ROOT, lib)), capture_output=True, text=True)
        print(os.path.basename(lib), r.stdout.strip(), r.stderr.strip()[-300:] if r.returncode else '', flush=True)
