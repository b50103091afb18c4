"""In-process A/B: layer1's data gradients (64 channels) WITHOUT the fused BatchNorm-backward epilogue -- the plain data gradient on
the filter-in-registers kernel + the separate reduce pass -- against the fused conv_win_kernel form (default) and the fused form
of the 64-channel kernel (variant bit 9 cleared)."""
import sys, time, yaml, torch
sys.path.insert(0, '.')
import bench
from multimodal_plankton_recognition_amd import ops, _native as N
from multimodal_plankton_recognition_amd.model import MultiModel
dev = torch.device('cuda', 0)
card = yaml.safe_load(open(bench.CARD))
torch.manual_seed(0)
model = MultiModel(card['dim_embedding'], card['image_encoder_args'], card['profile_encoder_args'],
                   card['coordination_args'], card['optim_args']).to(dev).train()
opt = model.configure_optimizers()
batch = bench.synthetic_batch(card['bs'], card['target_size'], dev, 1234)
batch['buckets'] = card['buckets']
def one_step():
    opt.zero_grad()
    loss = model.training_step(batch, 0)
    loss.backward()
    opt.step()
def timed(n=30):
    for _ in range(4): one_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): one_step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for _ in range(8): one_step()
real = ops.conv_dgrad_bn
def unfused64(dy, wd, g, x_shape, *a, **k):
    if x_shape[-1] == 64 and g.K == 64:
        return None
    return real(dy, wd, g, x_shape, *a, **k)
settings = ['default', 'layer1 unfused', 'layer1 fused on the 64-channel kernel']
res = {s: [] for s in settings}
for rep in range(4):
    for s in settings:
        if s == 'layer1 unfused': ops.conv_dgrad_bn = unfused64
        if s.startswith('layer1 fused on'): old = N.query('mpr_conv_set_window_variant', 5)
        res[s].append(timed())
        ops.conv_dgrad_bn = real
        if s.startswith('layer1 fused on'): N.query('mpr_conv_set_window_variant', old)
for s in settings:
    print(f'{s:42s} ' + ' '.join(f'{t:6.3f}' for t in res[s]) + f'   mean {sum(res[s]) / len(res[s]):6.3f} ms/step', flush=True)
