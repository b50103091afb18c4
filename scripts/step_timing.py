"""Where does a C3 step go: wall per step, host enqueue time per step, with/without the event profiler and parity walk."""
import os, sys, time, yaml, torch
sys.path.insert(0, '.')
import bench
from multimodal_plankton_recognition_amd import _native as N
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
    return loss
def run(tag, steps=15):
    for _ in range(3): one_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): one_step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{tag:34s} wall {(t2-t0)/steps*1e3:6.2f} ms/step   host enqueue {(t1-t0)/steps*1e3:6.2f} ms/step', flush=True)
lib = N.lib()
run('default')
lib.mpr_prof_enable(1); run('event profiler on'); lib.mpr_prof_enable(0)
N.query('mpr_conv_set_dgrad_parity', 0); run('parity walk off'); N.query('mpr_conv_set_dgrad_parity', 1)
run('default again')
