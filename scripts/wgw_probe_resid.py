"""Round 3: are two six-wave weight-gradient workgroups co-resident on a CU, and what do their waves wait for?
Per-wave probe (dbg bit 2) + HW_ID + 100 MHz start / end stamps.  argv: mode target [H C]"""
import sys, ctypes, collections, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd import ops, _native as N
ops.AUTOTUNE = False
mode, target = int(sys.argv[1]), int(sys.argv[2])
H, C = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (28, 128)
extra = int(sys.argv[5]) if len(sys.argv) > 5 else 0      # more dbg bits (4: no DMA address arithmetic, 8: no DMA issue)
B, K = 512, C
N.query('mpr_conv_set_wgrad_target_wgs', target)
N.query('mpr_conv_set_wgrad_window', mode)
g = ops.ConvGeom((K, C, 3, 3), 1, 1)
x = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
dy = torch.randn(B, H, H, K, device='cuda').to(torch.bfloat16)
for _ in range(3): ops.conv_wgrad(x, dy, g, (K, C, 3, 3))
buf = torch.zeros(4096 * 16 * 8, dtype=torch.int64, device='cuda')
N.lib().mpr_conv_debug_wgrad_probe(ctypes.c_void_p(buf.data_ptr()))
N.query('mpr_conv_set_wgrad_window', mode | ((2 | extra) << 8))
ops.conv_wgrad(x, dy, g, (K, C, 3, 3))
torch.cuda.synchronize()
N.query('mpr_conv_set_wgrad_window', 1)
N.lib().mpr_conv_debug_wgrad_probe(None)
t = buf.view(-1, 16, 8).cpu()
t = t[t[:, 0, 5] > 0]
nw = int((t[0, :, 5] > 0).sum())
n = float(t[0, 0, 5])
print(f'mode {mode} dbg {extra} target {target} {H}x{H}x{C}: {len(t)} WGs x {nw} waves, {n:.0f} chunks each')
for w in range(nw):
    tw = t[:, w].double()
    print(f'  wave {w:2d}: per chunk total {tw[:,0].mean()/n:6.0f}  wait {tw[:,1].mean()/n:6.0f} barrier {tw[:,2].mean()/n:6.0f} issue {tw[:,3].mean()/n:6.0f} compute {tw[:,4].mean()/n:6.0f}')
# residency: (xcc, se, sh, cu) of wave 0 + its [start, end) on the 100 MHz clock
hw = t[:, 0, 6]
cu = [((int(h) >> 32) & 15, (int(h) >> 13) & 7, (int(h) >> 12) & 1, (int(h) >> 8) & 15) for h in hw]
st = [int(v) & 0xFFFFFFFF for v in t[:, 0, 7]]
en = [(int(v) >> 32) & 0xFFFFFFFF for v in t[:, 0, 7]]
t0 = min(st)
by = collections.defaultdict(list)
for c, s, e in zip(cu, st, en): by[c].append((s - t0, e - t0))
ov = 0
for c, iv in by.items():
    iv.sort()
    for i in range(len(iv) - 1):
        if iv[i + 1][0] < iv[i][1] - 10: ov += 1
print(f'  {len(by)} distinct CUs; workgroups per CU: {collections.Counter(len(v) for v in by.values())}; pairs overlapping in time on one CU: {ov}')
print(f'  kernel span {(max(en) - t0) / 100:.1f} us; mean WG life {sum(e - s for s, e in zip(st, en)) / len(st) / 100:.1f} us')
simd = collections.Counter()
for i in range(len(t)):
    simd[tuple(sorted(((int(t[i, w, 6]) >> 4) & 3) for w in range(nw)))] += 1
print('  SIMD placement of a workgroup\'s waves:', dict(simd))
