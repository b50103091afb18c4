"""In-process A/B of module-level switches of ops.py on the C3 step (same GPU, same clocks, alternating):
    python scripts/step_ab_flags.py "DUAL_BN_APPLY=False" "N:mpr_conv_set_window_fwd_min_width=8" ...
(NAME=value: attribute of ops; N:function=value: a knob of include/mpr_hip_debug.h, restored from its return value.)  Every setting is applied alone (all others at their defaults) and measured `reps` times, interleaved with the default."""
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
settings = ['default'] + sys.argv[1:]
res = {s: [] for s in settings}
for rep in range(4):
    for s in settings:
        saved = {}
        if s != 'default':
            for kv in s.split(','):
                k, v = kv.split('=')
                if k.startswith('N:'):
                    saved[k] = N.query(k[2:], int(v))
                else:
                    saved[k] = getattr(ops, k)
                    setattr(ops, k, eval(v))
        res[s].append(timed())
        for k, v in saved.items():
            if k.startswith('N:'):
                N.query(k[2:], v)
            else:
                setattr(ops, k, v)
for s in settings:
    print(f'{s:40s} ' + ' '.join(f'{t:6.3f}' for t in res[s]) + f'   mean {sum(res[s]) / len(res[s]):6.3f} ms/step', flush=True)
