"""A/B inside ONE process (same GPU, same clocks): C3 step time under a list of settings.
    python scripts/step_ab.py"""
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
def timed(n=15):
    for _ in range(4): one_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): one_step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for _ in range(6): one_step()
for rep in range(2):
    for name, setup in [('default', lambda: None),
                        ('one stream for both encoders', lambda: setattr(model, 'two_streams', False)),
                        ('weight gradients in-stream', lambda: setattr(ops, 'ASYNC_WGRAD', False)),
                        ('BN partial lists always pre-reduced', lambda: setattr(ops, 'FIN_DIRECT_FLOATS', 0)),
                        ('BN backward sums via partial rows + pre-reduce', lambda: setattr(ops, 'BWD_ATOMIC_SLICES', False)),
                        ('conv BN sums one row per tile + pre-reduce', lambda: N.query('mpr_conv_set_stat_slices', 0)),
                        ('slice rows zeroed per call (no arena)', lambda: setattr(ops, 'SLICE_ARENA', False)),
                        ('both off', lambda: (setattr(model, 'two_streams', False), setattr(ops, 'ASYNC_WGRAD', False)))]:
        model.two_streams, ops.ASYNC_WGRAD, ops.FIN_DIRECT_FLOATS, ops.BWD_ATOMIC_SLICES = True, True, 4096, True
        ops.SLICE_ARENA = True
        N.query('mpr_conv_set_stat_slices', 8)
        setup()
        print(f'{name:32s} {timed():6.2f} ms/step', flush=True)
model.two_streams, ops.ASYNC_WGRAD = True, True
# profile branch alone / image branch alone (forward + backward of one encoder, no loss): what each costs by itself
pe, ie = model.profile_encoder, model.image_encoder
def branch(enc, **kw):
    out = enc(**kw)
    out.sum().backward()
for nm, fn in [('profile encoder fwd+bwd alone', lambda: branch(pe, profile=batch['profile'], profile_len=batch['profile_len'])),
               ('image encoder fwd+bwd alone', lambda: branch(ie, image=batch['image'], image_shape=batch['image_shape']))]:
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    print(f'{nm:32s} {(time.perf_counter() - t0) / 10 * 1e3:6.2f} ms', flush=True)
