"""Soak: N steps of a card, reporting step time and allocated memory every 50 steps (leaks / drift would show)."""
import sys, time, yaml, torch
sys.path.insert(0, '.')
import bench
from multimodal_plankton_recognition_amd import transformer as TF, ops
from multimodal_plankton_recognition_amd.model import MultiModel
card = yaml.safe_load(open(sys.argv[1]))
B, steps = int(sys.argv[2]), int(sys.argv[3])
TF.set_precision((card.get('trainer_args') or {}).get('precision'))
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = MultiModel(card['dim_embedding'], card['image_encoder_args'], card['profile_encoder_args'],
                   card['coordination_args'], card['optim_args']).to(dev).train()
opt = model.configure_optimizers()
batch = bench.synthetic_batch(B, card['target_size'], dev, 1234, transformer='num_head' in card['profile_encoder_args'])
batch['buckets'] = card['buckets']
t0 = time.perf_counter()
for i in range(1, steps + 1):
    opt.zero_grad()
    loss = model.training_step(batch, 0)
    loss.backward()
    opt.step()
    model.train_loss.clear()
    if i % 50 == 0:
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50 * 1e3
        print(f'step {i:4d}: {dt:6.2f} ms/step, loss {float(loss.detach()):.4f}, allocated {torch.cuda.memory_allocated() / 2**20:8.1f} MiB, '
              f'reserved {torch.cuda.memory_reserved() / 2**20:8.1f} MiB, pack entries {len(ops.pack_registry.entries)}', flush=True)
        t0 = time.perf_counter()
