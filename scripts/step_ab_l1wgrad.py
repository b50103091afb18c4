"""In-process A/B: the workgroup target of layer1's four window weight gradients (they become ready last and run at the tail of
the step, beside little else) -- the tuner's CU-time choice against fixed targets."""
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
print('tuner:', {k: v for k, v in ops._wgrad_split.items()})
base = dict(ops._wgrad_split)
l1 = [k for k in base if k[1] == 56 and k[3] == 64 and k[4] == 64]
tgs = [int(v) for v in sys.argv[1:]]
settings = [('tuner', {})] + ([(f'all window -> {tg}', {k: (1, tg) for k, v in base.items() if v[0] == 1}) for tg in tgs] if tgs else
                             [(f'layer1 -> {tg}', {k: (1, tg) for k in l1}) for tg in (192, 256)] +
                             [('all window -> 256', {k: (1, 256) for k, v in base.items() if v[0] == 1})])
res = {n: [] for n, _ in settings}
for rep in range(4):
    for n, over in settings:
        ops._wgrad_split.clear(); ops._wgrad_split.update(base); ops._wgrad_split.update(over)
        res[n].append(timed())
ops._wgrad_split.clear(); ops._wgrad_split.update(base)
for n, _ in settings:
    print(f'{n:24s} ' + ' '.join(f'{t:6.3f}' for t in res[n]) + f'   mean {sum(res[n]) / len(res[n]):6.3f} ms/step', flush=True)
