"""C3 step with the image chain on a HIGH-priority stream (the profile branch and the weight-gradient side streams stay at
normal priority: this device has two levels, 0 and -1) against everything on normal priority.  Two processes would differ by
box noise: both variants run in this one, alternating."""
import sys, time, yaml, torch
sys.path.insert(0, '.')
import bench
from multimodal_plankton_recognition_amd.model import MultiModel
dev = torch.device('cuda', 0)
card = yaml.safe_load(open(bench.CARD))
def build():
    torch.manual_seed(0)
    m = MultiModel(card['dim_embedding'], card['image_encoder_args'], card['profile_encoder_args'], card['coordination_args'],
                   card['optim_args']).to(dev).train()
    return m, m.configure_optimizers()
batch = bench.synthetic_batch(card['bs'], card['target_size'], dev, 1234)
batch['buckets'] = card['buckets']
def timed(model, opt, n=15):
    def one():
        opt.zero_grad(); loss = model.training_step(batch, 0); loss.backward(); opt.step()
    for _ in range(6): one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print('priority range', torch.cuda.Stream.priority_range())
m0, o0 = build()
hi = torch.cuda.Stream(priority=-1)
m1, o1 = build()
for rep in range(4):
    t_norm = timed(m0, o0)
    with torch.cuda.stream(hi):
        t_hi = timed(m1, o1)
    print(f'normal {t_norm:6.2f} ms/step   chain on the high-priority stream {t_hi:6.2f} ms/step', flush=True)
