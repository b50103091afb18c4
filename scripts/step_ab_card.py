"""In-process A/B of library knobs on ANY card (cf. step_ab_flags.py, which is fixed to C3):
    python scripts/step_ab_card.py <card> <batch> "N:function=value" ...   (each setting alone, interleaved with the default)"""
import sys, time, yaml, torch
sys.path.insert(0, '.')
import bench
from multimodal_plankton_recognition_amd import ops, _native as N, transformer as TF
from multimodal_plankton_recognition_amd.model import MultiModel
card = yaml.safe_load(open(sys.argv[1]))
B = int(sys.argv[2])
TF.set_precision((card.get('trainer_args') or {}).get('precision'))
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = MultiModel(card['dim_embedding'], card['image_encoder_args'], card['profile_encoder_args'],
                   card['coordination_args'], card['optim_args']).to(dev).train()
opt = model.configure_optimizers()
batch = bench.synthetic_batch(B, card['target_size'], dev, 1234, transformer='num_head' in card['profile_encoder_args'])
batch['buckets'] = card['buckets']
def one_step():
    opt.zero_grad()
    loss = model.training_step(batch, 0)
    loss.backward()
    opt.step()
def timed(n=10):
    for _ in range(2): one_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): one_step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for _ in range(4): one_step()
settings = ['default'] + sys.argv[3:]
res = {s: [] for s in settings}
for rep in range(3):
    for s in settings:
        old = None
        if s != 'default':
            fn, val = s[2:].split('=')
            old = N.query(fn, int(val))
        res[s].append(timed())
        if s != 'default':
            N.query(fn, old)
for s in settings:
    print(f'{s:42s}', ' '.join(f'{v:7.3f}' for v in res[s]), f'  mean {sum(res[s]) / len(res[s]):7.3f} ms/step', flush=True)
