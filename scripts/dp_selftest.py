"""RCCL self-test of the data-parallel step on ONE GPU: a world-size-1 `nccl` process group exercises every collective
call of distributed.DataParallelStep (all_gather_into_tensor, all_reduce on the optimizer's flat gradient buffer,
barrier, max_over_ranks) through real RCCL, and the result must equal the plain single-GPU step bit for bit in the loss.
    python scripts/dp_selftest.py [card] [batch]"""
import os, sys, yaml, torch
sys.path.insert(0, '.')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
import bench
from multimodal_plankton_recognition_amd import distributed as D, transformer as TF
from multimodal_plankton_recognition_amd.model import MultiModel
card = yaml.safe_load(open(sys.argv[1] if len(sys.argv) > 1 else bench.CARD))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
TF.set_precision((card.get('trainer_args') or {}).get('precision'))
D.init(dev, backend='nccl')
card['image_encoder_args']['dropout'] = 0.0
card['profile_encoder_args']['dropout'] = 0.0


def run(dp):
    torch.manual_seed(0)
    model = MultiModel(card['dim_embedding'], card['image_encoder_args'], card['profile_encoder_args'],
                       card['coordination_args'], card['optim_args']).to(dev).train()
    opt = model.configure_optimizers()
    batch = bench.synthetic_batch(B, card['target_size'], dev, 1234, transformer='num_head' in card['profile_encoder_args'])
    batch['buckets'] = 1
    stepper = D.DataParallelStep(model, opt, 1) if dp else None
    losses = []
    for _ in range(3):
        if dp:
            loss = stepper.step(batch)
        else:
            opt.zero_grad()
            loss = model.training_step(batch, 0)
            loss.backward()
            opt.step()
        losses.append(float(loss.detach()))
    return losses


a, b = run(False), run(True)
D.barrier()
assert abs(D.max_over_ranks(1.5) - 1.5) < 1e-12
print('single-GPU step :', a)
print('DP step (nccl 1):', b)
assert all(abs(x - y) <= 2e-3 * max(1.0, abs(x)) for x, y in zip(a, b)), 'DP step diverges from the single-GPU step'
D.shutdown()
print('dp_selftest ok')
