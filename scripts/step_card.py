"""Wall time per full optimisation step of any model card on synthetic data:
    python scripts/step_card.py model_cards/vit_base_transformer_siglip.yaml 128 [precision] [steps]
(C5 per-GPU share: ViT-B/16 + ProfileTransformer + SigLIP at batch 128; precision defaults to the card's
trainer_args.precision; '32' = exact-fp32 transformer kernels)."""
import sys, time, yaml, torch
sys.path.insert(0, '.')
import bench
from multimodal_plankton_recognition_amd import transformer as TF
from multimodal_plankton_recognition_amd.model import MultiModel
card = yaml.safe_load(open(sys.argv[1]))
B = int(sys.argv[2]) if len(sys.argv) > 2 else card['bs']
prec = sys.argv[3] if len(sys.argv) > 3 else (card.get('trainer_args') or {}).get('precision')
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
TF.set_precision(prec)
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
    return loss
for _ in range(3): loss = one_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): loss = one_step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
print(f'{sys.argv[1]} batch {B} precision {TF._PRECISION[0]}: {ms:.2f} ms/step, {B / ms * 1e3:.0f} samples/s, loss {float(loss.detach()):.4f}, '
      f'peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB', flush=True)
