"""cProfile of the host side of a C3 step (where does the enqueue time go)."""
import cProfile, pstats, sys, yaml, torch
sys.path.insert(0, '.')
import bench
from multimodal_plankton_recognition_amd.model import MultiModel
dev = torch.device('cuda', 0)
card = yaml.safe_load(open(sys.argv[2] if len(sys.argv) > 2 else bench.CARD))
torch.manual_seed(0)
model = MultiModel(card['dim_embedding'], card['image_encoder_args'], card['profile_encoder_args'],
                   card['coordination_args'], card['optim_args']).to(dev).train()
opt = model.configure_optimizers()
batch = bench.synthetic_batch(int(sys.argv[3]) if len(sys.argv) > 3 else card["bs"], card['target_size'], dev, 1234)
batch['buckets'] = card['buckets']
def one_step():
    opt.zero_grad()
    loss = model.training_step(batch, 0)
    loss.backward()
    opt.step()
for _ in range(3): one_step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5): one_step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 35)
