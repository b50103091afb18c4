import sys, numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd.profile_encoder import ProfileCNN
from multimodal_plankton_recognition_amd.layers import StemFn
from oracle.profile_encoder import _bn, _basic_block_1d
from oracle.rounding import emulate_bf16, r
torch.manual_seed(0)
B = 32
m = ProfileCNN(dim_in=6, blocks=[2, 2, 2, 2], base_channels=32, dropout=0.0)
sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
m.cuda().train()
x = torch.rand(B, 224, 6) * 2 - 1
def rl2(a, b):
    a = a.detach().float().cpu(); b = torch.as_tensor(b).detach().float()
    return round(float((a - b).norm() / b.norm().clamp_min(1e-12)), 5), round(float((a-b).abs().max()), 4)
with emulate_bf16():
    o = r(F.conv1d(x.transpose(1, 2), sd['conv1.weight'], None, 2, 1))
    o = F.max_pool1d(r(F.relu(_bn(sd, 'bn1', o, True))), 3, 2, 1)
    mine = StemFn.apply(x.cuda(), m.conv1.weight, m.bn1.weight, m.bn1.bias, m)
    print('stem', rl2(mine.transpose(1, 2), o))
    for li in range(1, 5):
        for bi, blk in enumerate(getattr(m, f'layer{li}')):
            o_in = o
            o = _basic_block_1d(sd, f'layer{li}.{bi}.', o, 2 if (li > 1 and bi == 0) else 1, True)
            mine = blk(mine)
            # isolated: feed oracle's (bf16-exact) input to my block
            iso = blk(o_in.transpose(1, 2).contiguous().to(torch.bfloat16).cuda())
            print(f'layer{li}.{bi}', 'chained', rl2(mine.transpose(1, 2), o), 'isolated', rl2(iso.transpose(1, 2), o))
