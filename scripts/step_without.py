"""What the C3 step would cost WITHOUT one family of kernels (their launches replaced by nothing: results are wrong, timing only):
a bound on what any speed-up of that family can return inside the step.  usage: python scripts/step_without.py"""
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
def timed(n=20):
    for _ in range(4): one_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): one_step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for _ in range(8): one_step()
real_call = N.call
def without(names):
    def call(fn, *a, **k):
        if fn in names:
            return None
        return real_call(fn, *a, **k)
    return call
cases = [('default', ()),
         ('no weight gradients', ('mpr_conv_wgrad', 'mpr_conv_wgrad_ex')),
         ('no weight gradients, no stem weight gradient', ('mpr_conv_wgrad', 'mpr_conv_wgrad_ex', 'mpr_stem_wgrad', 'mpr_stemf_bwd')),
         ('no BatchNorm backward apply passes', ('mpr_bn_bwd_apply_fin', 'mpr_bn_bwd_apply')),
         ('no BatchNorm forward apply passes', ('mpr_bn_apply', 'mpr_bn_apply_fin', 'mpr_bn_apply_dual', 'mpr_bn_apply_dual_fin')),
         ('no SGD / repack', ('mpr_sgd_multi', 'mpr_conv_pack_weights_multi'))]
for rep in range(2):
    for name, names in cases:
        N.call = without(set(names))
        ops.N.call = N.call
        try:
            t = timed()
        except Exception as e:
            t = float('nan'); print(name, 'failed:', repr(e)[:120])
        N.call = real_call; ops.N.call = real_call
        print(f'{name:50s} {t:7.3f} ms/step', flush=True)
