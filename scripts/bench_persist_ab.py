"""Layer1's persistent forward kernel, A/B of two builds of the library in child processes (MPR_HIP_LIB):
usage: bench_persist_ab.py <lib A> <lib B>."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from multimodal_plankton_recognition_amd import ops
B, H, C, K = 512, 56, 64, 64
g = ops.ConvGeom((K, C, 3, 3), 1, 1)
w = torch.randn(K, C, 3, 3, device='cuda') * 0.05
wf, wd = ops.packed_weights(w, g)
torch.manual_seed(0)
x = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
y = ops.conv_fwd(x, wf, g, True)
print('fwd %%.1f us  checksum %%.6e %%.6e' %% (t(lambda: ops.conv_fwd(x, wf, g, True)), float(y[0].float().sum()), float(y[1].float().sum())))
''' % ROOT
for lib in sys.argv[1:]:
    for rep in range(2):
        r = subprocess.run([sys.executable, '-c', CHILD], env=dict(os.environ, MPR_HIP_LIB=os.path.join(ROOT, lib)), capture_output=True, text=True)
        print(lib, r.stdout.strip(), r.stderr.strip()[-300:] if r.returncode else '', flush=True)
