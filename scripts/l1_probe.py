"""Shader-clock breakdown of the filter-in-registers 64 -> 64 kernel (conv_win_l1_kernel), wave 0 of every workgroup.  argv: [dgrad]"""
import sys, ctypes, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
B, H, C = 512, 56, 64
dg = len(sys.argv) > 1 and sys.argv[1] == "d"
nostats = len(sys.argv) > 1 and sys.argv[1] == "n"
g = ops.ConvGeom((C, C, 3, 3), 1, 1)
w = torch.randn(C, C, 3, 3, device='cuda') * 0.05
wf, wd = ops.packed_weights(w, g)
x = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
run = (lambda: ops.conv_dgrad(x, wd, g, tuple(x.shape))) if dg else (lambda: ops.conv_fwd(x, wf, g, not nostats))
for _ in range(3): run()
buf = torch.zeros(16384 * 16, dtype=torch.int64, device='cuda')
N.lib().mpr_conv_debug_probe(ctypes.c_void_p(buf.data_ptr()))
run()
torch.cuda.synchronize()
N.lib().mpr_conv_debug_probe(None)
t = buf.view(-1, 16).cpu().double()
t = t[t[:, 5] > 0]
n = t[:, 5].mean().item()
print(f'{"dgrad" if dg else "fwd+stats"}: {len(t)} WGs x {n:.1f} tiles | per tile: total {t[:,0].mean()/n:7.0f} cyc = window wait+barrier {t[:,1].mean()/n:6.0f} + MFMA loop {t[:,2].mean()/n:6.0f} + barrier {t[:,3].mean()/n:6.0f} + table+DMA issue {t[:,4].mean()/n:6.0f} + epilogue {t[:,6].mean()/n:6.0f}')
import collections
st, en = t[:, 7], t[:, 8]
t0 = st.min()
hw = [int(v) for v in t[:, 9]]
cu = [((h >> 32) & 15, (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 15) for h in hw]
by = collections.defaultdict(list)
for c, s_, e_ in zip(cu, st.tolist(), en.tolist()): by[c].append((s_ - t0.item(), e_ - t0.item()))
ov = sum(1 for iv in by.values() for i in range(len(iv)) for j in range(i + 1, len(iv)) if min(iv[i][1], iv[j][1]) - max(iv[i][0], iv[j][0]) > 100)
print(f'  span {(en.max() - t0).item() / 100:.1f} us; mean workgroup life {(en - st).mean().item() / 100:.1f} us; implied clock {t[:,0].mean() / ((en - st).mean().item() / 100) / 1e3:.2f} GHz; {len(by)} CUs, workgroups per CU {dict(collections.Counter(len(v) for v in by.values()))}, pairs overlapping in time on a CU: {ov}')
