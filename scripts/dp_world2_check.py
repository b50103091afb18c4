"""Whole data-parallel step at world size 2 on ONE GPU (gloo rehearsal: both ranks share cuda:0, collectives staged through
the host) against a single-process reference of the same semantics -- per-rank BatchNorm statistics, contrastive loss over
the GLOBAL batch, summed gradients.  Each mode writes its post-training parameters to <outdir>/<mode>.pt.

    RANK=r WORLD_SIZE=2 MPR_DIST_BACKEND=gloo python scripts/dp_world2_check.py rank <outdir> [method]
    python scripts/dp_world2_check.py ref <outdir> [method] [steps] [accumulate] [precision]

accumulate > 1: every optimizer step is a window of that many micro-batches (Lightning's accumulate_grad_batches: each
micro-batch's loss / accumulate); precision 32: the conv stacks' fp32 parity mode (no atomics: the comparison is tight).
"""
import os, sys, yaml, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multimodal_plankton_recognition_amd import distributed as D
from multimodal_plankton_recognition_amd.model import MultiModel

mode, outdir = sys.argv[1], sys.argv[2]
method = sys.argv[3] if len(sys.argv) > 3 else 'clip'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
card = yaml.safe_load(open(os.path.join(ROOT, 'model_cards', 'smoke_multi.yaml')))
card['image_encoder_args']['dropout'] = 0.0
card['profile_encoder_args']['dropout'] = 0.0
card['coordination_args'] = {'method': method}
B, T, WORLD = 8, card['target_size'], 2
STEPS = int(sys.argv[4]) if len(sys.argv) > 4 else 1
OF = int(sys.argv[5]) if len(sys.argv) > 5 else 1
if len(sys.argv) > 6 and sys.argv[6] != '-':
    from multimodal_plankton_recognition_amd import layers_f32
    layers_f32.set_conv_precision(sys.argv[6])
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
torch.manual_seed(0)
model = MultiModel(card['dim_embedding'], card['image_encoder_args'], card['profile_encoder_args'],
                   card['coordination_args'], card['optim_args']).to(dev).train()
opt = model.configure_optimizers()


def shard(step, r):
    g = bench.synthetic_batch(WORLD * B, T, dev, 500 + step)
    out = {k: v[r * B:(r + 1) * B].contiguous() for k, v in g.items()}
    out['buckets'] = 1
    return out


losses = []
if mode == 'rank':
    rank = int(os.environ['RANK'])
    D.init(dev, backend='gloo')
    stepper = D.DataParallelStep(model, opt, WORLD)
    for s in range(STEPS):
        for micro in range(OF):
            losses.append(float(stepper.step(shard(s * OF + micro, rank), micro=micro, of=OF).detach()))
    assert stepper.verified == [True] * min(STEPS, 2), stepper.verified
    D.barrier()
    D.shutdown()
    tag = f'rank{rank}'
else:
    for s in range(STEPS):
        opt.zero_grad()
        for micro in range(OF):
            embs = [model.encode(**shard(s * OF + micro, r)) for r in range(WORLD)]          # per-rank BatchNorm statistics
            loss = model.loss(image_emb=torch.cat([e['image_emb'] for e in embs]),
                              profile_emb=torch.cat([e['profile_emb'] for e in embs]), buckets=1)
            (loss / OF if OF > 1 else loss).backward()
            losses.append(float(loss.detach()))
        opt.step()
    tag = 'ref'
torch.cuda.synchronize()
sd = {k: v.detach().float().cpu() for k, v in model.named_parameters()}
torch.save({'params': sd, 'losses': losses}, os.path.join(outdir, tag + '.pt'))
print(tag, 'losses', losses, flush=True)
