"""Wall time per C3 step (15 steps after 4 warm-up) -- for A/B runs under different environment knobs."""
import os, sys, time, yaml, torch
sys.path.insert(0, '.')
import bench
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
res = []
for rep in range(3):
    for _ in range(4): one_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(15): one_step()
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 15 * 1e3)
print(' '.join(f'{k}={v}' for k, v in os.environ.items() if k.startswith('MPR_')) or 'default', '->', ' '.join(f'{r:.2f}' for r in res), 'ms/step', flush=True)
