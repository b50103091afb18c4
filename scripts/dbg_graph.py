import sys, time, torch, yaml
sys.path.insert(0, '.')
from bench import synthetic_batch, CARD
from multimodal_plankton_recognition_amd.model import MultiModel
card = yaml.safe_load(open(CARD))
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = MultiModel(card['dim_embedding'], card['image_encoder_args'], card['profile_encoder_args'],
                   card['coordination_args'], card['optim_args']).to(dev).train()
opt = model.configure_optimizers()
batch = synthetic_batch(512, 224, dev, 1234); batch['buckets'] = 1
def step():
    opt.zero_grad()
    loss = model.training_step(batch, 0)
    loss.backward()
    opt.step()
    return loss
for _ in range(3): step()
torch.cuda.synchronize()
# CPU enqueue time vs GPU time
t0 = time.perf_counter()
for _ in range(5): l = step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('eager: enqueue %.2f ms/step, total %.2f ms/step, loss %.5f' % ((t1 - t0) / 5 * 1e3, (t2 - t0) / 5 * 1e3, l.item()))
del l
model.train_loss.clear()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    gl = step()
torch.cuda.synchronize()
losses = []
t0 = time.perf_counter()
for _ in range(10):
    g.replay(); losses.append(gl.clone())
torch.cuda.synchronize(); t1 = time.perf_counter()
print('graph: %.2f ms/step' % ((t1 - t0) / 10 * 1e3), [round(x.item(), 4) for x in losses])
