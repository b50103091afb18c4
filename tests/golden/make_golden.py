"""Generate the golden fixtures in this directory by importing the REFERENCE's own
modules (``/root/reference/src/coordination.py`` and ``src/profile_encoder.py`` -- the
only hot-path files importable in the build container).

Run in the build container ONLY (the reference never travels to the GPU box):

    python3 -B tests/golden/make_golden.py

``-B`` / PYTHONDONTWRITEBYTECODE keeps /root/reference free of __pycache__.
The fixtures are data (inputs, seeded state_dicts, expected outputs and gradients);
no reference source text is stored.
"""
import os
import sys
import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')
import torch  # noqa: E402
from src.coordination import CLIPLoss, SigLIPLoss, CLIPPlus, SigLIPPlus, RankLoss  # noqa: E402
from src.profile_encoder import ProfileCNN, ProfileTransformer, ProfileLSTM  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)


def npy(t):
    return t.detach().cpu().numpy().copy()   # copy: state_dict tensors are live views


def randn(seed, *shape):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32))


def uniform(seed, lo, hi, *shape):
    return torch.from_numpy(np.random.RandomState(seed).uniform(lo, hi, shape).astype(np.float32))


def randomize_bn(module, seed):
    """Give every BatchNorm non-trivial affine parameters and running statistics."""
    rs = np.random.RandomState(seed)
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            n = m.num_features
            with torch.no_grad():
                m.weight.copy_(torch.from_numpy(rs.uniform(0.5, 1.5, n).astype(np.float32)))
                m.bias.copy_(torch.from_numpy(rs.uniform(-0.3, 0.3, n).astype(np.float32)))
                m.running_mean.copy_(torch.from_numpy(rs.uniform(-0.2, 0.2, n).astype(np.float32)))
                m.running_var.copy_(torch.from_numpy(rs.uniform(0.5, 2.0, n).astype(np.float32)))


# ----------------------------------------------------------------------------- losses
def loss_fixtures():
    out = {}
    cases = [(16, 32, 1), (16, 32, 4), (64, 512, 1), (64, 512, 4), (48, 40, 3)]
    for ci, (b, d, k) in enumerate(cases):
        a0, p0 = randn(100 + ci, b, d), randn(200 + ci, b, d) * 1.5 + 0.1
        out[f'case{ci}_shape'] = np.array([b, d, k])
        out[f'case{ci}_image_emb'] = npy(a0)
        out[f'case{ci}_profile_emb'] = npy(p0)
        for name, ctor in [('clip', CLIPLoss), ('siglip', SigLIPLoss),
                           ('clipplus', lambda: CLIPPlus(beta=.25)), ('siglipplus', lambda: SigLIPPlus(beta=.25))]:
            mod = ctor()
            # move the loss parameters off their init values so the gradients w.r.t. them are exercised
            with torch.no_grad():
                for pn, pv in mod.named_parameters():
                    pv.copy_(torch.tensor(1.3 if pn.endswith('logit_scale') else -4.0))
            a, p = a0.clone().requires_grad_(True), p0.clone().requires_grad_(True)
            loss = mod(a, p, k)
            loss.backward()
            pre = f'case{ci}_{name}_'
            out[pre + 'loss'] = npy(loss)
            out[pre + 'd_image'] = npy(a.grad)
            out[pre + 'd_profile'] = npy(p.grad)
            for pn, pv in mod.named_parameters():
                out[pre + 'param_' + pn] = npy(pv)
                out[pre + 'dparam_' + pn] = npy(pv.grad)
            # init-valued parameters too (logit_scale=1, bias=-10): plain forward value
            out[pre + 'loss_init'] = npy(ctor()(a0, p0, k))
        a, p = a0.clone().requires_grad_(True), p0.clone().requires_grad_(True)
        loss = RankLoss(margin=.25)(a, p)
        loss.backward()
        out[f'case{ci}_rank_loss'] = npy(loss)
        out[f'case{ci}_rank_d_image'] = npy(a.grad)
        out[f'case{ci}_rank_d_profile'] = npy(p.grad)
    # the SURVEY's spot values (seed 0, B=64, D=512, randn via torch.manual_seed)
    torch.manual_seed(0)
    a, p = torch.randn(64, 512), torch.randn(64, 512)
    out['survey_clip'] = np.array([CLIPLoss()(a, p, 1).item(), CLIPLoss()(a, p, 4).item()])
    out['survey_siglip'] = np.array([SigLIPLoss()(a, p, 1).item(), SigLIPLoss()(a, p, 4).item()])
    out['survey_rank'] = np.array([RankLoss(.25)(a, p).item()])
    # big case: scalars + gradient digests only (inputs are regenerated from the seeds)
    a0, p0 = randn(900, 512, 512), randn(901, 512, 512)
    for name, ctor in [('clip', CLIPLoss), ('siglip', SigLIPLoss)]:
        a, p = a0.clone().requires_grad_(True), p0.clone().requires_grad_(True)
        mod = ctor()
        loss = mod(a, p, 1)
        loss.backward()
        out[f'big_{name}_loss'] = npy(loss)
        out[f'big_{name}_d_image_rows'] = npy(a.grad[:4])
        out[f'big_{name}_d_profile_rows'] = npy(p.grad[-4:])
        out[f'big_{name}_d_image_abs_sum'] = np.array(a.grad.double().abs().sum().item())
        out[f'big_{name}_d_profile_abs_sum'] = np.array(p.grad.double().abs().sum().item())
        out[f'big_{name}_dscale'] = npy(mod.logit_scale.grad)
    # margin-controlled retrieval case: profile = permuted image embedding + small noise
    a0 = randn(910, 32, 64)
    perm = np.random.RandomState(911).permutation(32)
    p0 = (a0[torch.from_numpy(perm)] * 1.7 + 0.05 * randn(912, 32, 64))
    u = torch.nn.functional.normalize(a0)
    v = torch.nn.functional.normalize(p0)
    s = u @ v.T
    top2 = s.topk(2, dim=1).values
    assert (top2[:, 0] - top2[:, 1]).min() > 0.3, "margin too small"
    out['margin_image_emb'] = npy(a0)
    out['margin_profile_emb'] = npy(p0)
    out['margin_row_argmax'] = npy(s.argmax(1))
    out['margin_col_argmax'] = npy(s.argmax(0))
    out['margin_clip_loss'] = npy(CLIPLoss()(a0, p0, 1))
    np.savez_compressed(os.path.join(HERE, 'losses.npz'), **out)


# ----------------------------------------------------------------------------- ProfileCNN
def cnn_fixture(tag, blocks, base, batch, length, seed):
    torch.manual_seed(seed)
    m = ProfileCNN(dim_in=6, blocks=blocks, base_channels=base, dropout=0.0)
    randomize_bn(m, seed + 1)
    out = {'blocks': np.array(blocks), 'base': np.array(base)}
    for k, v in m.state_dict().items():
        out['sd.' + k] = npy(v)
    x = uniform(seed + 2, -1, 1, batch, length, 6)
    plen = torch.from_numpy(np.random.RandomState(seed + 3).randint(8, 1024, (batch, 1)))
    wsum = randn(seed + 4, batch, m.dim_out)
    out['profile'] = npy(x)
    out['profile_len'] = npy(plen)
    out['wsum'] = npy(wsum)
    m.eval()
    with torch.no_grad():
        out['eval.features'] = npy(m.forward_features(x))
        out['eval.out'] = npy(m(x, profile_len=plen.clone()))
    m.train()
    y = m(x, profile_len=plen.clone())
    out['train.out'] = npy(y)
    (y * wsum).sum().backward()
    for k, v in m.named_parameters():
        out['train.grad.' + k] = npy(v.grad)
    for k, v in m.state_dict().items():
        if 'running' in k or 'num_batches' in k:
            out['train.after.' + k] = npy(v)
    np.savez_compressed(os.path.join(HERE, f'profile_cnn_{tag}.npz'), **out)


# ----------------------------------------------------------------------------- ProfileTransformer
def transformer_fixture():
    torch.manual_seed(7)
    cfg = dict(dim_in=6, dim_hidden=32, target_size=224, num_head=2, num_layers=2, dim_feedforward=64,
               dropout=0.0, activation='gelu')
    m = ProfileTransformer(**cfg)
    with torch.no_grad():   # torch inits biases / LN to 0 / 1: perturb so that they matter
        rs = np.random.RandomState(8)
        for k, v in m.named_parameters():
            if v.ndim == 1:
                v.add_(torch.from_numpy(rs.uniform(-0.2, 0.2, v.shape).astype(np.float32)))
    out = {}
    for k, v in m.state_dict().items():
        out['sd.' + k] = npy(v)
    out['padding_idx'] = np.array(m.padding_idx)
    for tag, lens in [('ragged', [5, 17, 224]), ('fixed', [224, 224])]:
        profs = [uniform(20 + i, -1, 1, n, 6) for i, n in enumerate(lens)]
        tok = m.tokenize(profs)
        plen = torch.tensor([[n] for n in lens])
        for i, p in enumerate(profs):
            out[f'{tag}.in{i}'] = npy(p)
        out[f'{tag}.lens'] = np.array(lens)
        for k, v in tok.items():
            out[f'{tag}.tok.{k}'] = npy(v)
        m.train()   # dropout=0: train == eval arithmetic, and avoids the nested-tensor fast path
        m.zero_grad()
        y = m(**tok, profile_len=plen.clone())
        wsum = randn(30 + len(lens), *y.shape)
        (y * wsum).sum().backward()
        out[f'{tag}.out'] = npy(y)
        out[f'{tag}.wsum'] = npy(wsum)
        for k, v in m.named_parameters():
            out[f'{tag}.grad.{k}'] = npy(v.grad)
    np.savez_compressed(os.path.join(HERE, 'profile_transformer.npz'), **out)


# ----------------------------------------------------------------------------- ProfileLSTM
def lstm_fixture():
    torch.manual_seed(11)
    m = ProfileLSTM(dim_in=6, dim_hidden=16, num_layers=2, dropout=0.0)
    out = {}
    for k, v in m.state_dict().items():
        out['sd.' + k] = npy(v)
    profs = [uniform(40 + i, -1, 1, n, 6) for i, n in enumerate([12, 7, 9])]
    tok = m.tokenize(profs)
    plen = torch.tensor([[12], [7], [9]])
    for k, v in tok.items():
        out['tok.' + k] = npy(v)
    out['profile_len'] = npy(plen)
    out['out'] = npy(m(**tok, profile_len=plen.clone()))
    np.savez_compressed(os.path.join(HERE, 'profile_lstm.npz'), **out)


# ----------------------------------------------------------------------------- composed step
def composed_step_fixture():
    """ProfileCNN -> Linear(no bias) || given image features -> Linear(no bias) -> CLIPLoss -> 2 SGD steps
    (nesterov + weight decay, covering the loss parameter) -- SURVEY 8c fixture (6)."""
    torch.manual_seed(21)
    enc = ProfileCNN(dim_in=6, blocks=[1, 1, 1, 1], base_channels=8, dropout=0.0)
    randomize_bn(enc, 22)
    pproj = torch.nn.Linear(enc.dim_out, 32, bias=False)
    iproj = torch.nn.Linear(18, 32, bias=False)
    loss_mod = CLIPLoss()
    mods = {'profile_encoder.': enc, 'profile_projection.': pproj, 'image_projection.': iproj, 'loss.': loss_mod}
    out = {}
    for pre, m in mods.items():
        for k, v in m.state_dict().items():
            out['sd0.' + pre + k] = npy(v)
    params = [p for m in mods.values() for p in m.parameters()]
    opt = torch.optim.SGD(params, lr=5e-2, momentum=0.9, weight_decay=1e-3, nesterov=True)
    for step in range(2):
        img_feat = randn(50 + step, 16, 18)
        prof = uniform(60 + step, -1, 1, 16, 64, 6)
        plen = torch.from_numpy(np.random.RandomState(70 + step).randint(8, 1024, (16, 1)))
        out[f'step{step}.image_feat'] = npy(img_feat)
        out[f'step{step}.profile'] = npy(prof)
        out[f'step{step}.profile_len'] = npy(plen)
        opt.zero_grad()
        loss = loss_mod(iproj(img_feat), pproj(enc(prof, profile_len=plen.clone())), 2)
        loss.backward()
        opt.step()
        out[f'step{step}.loss'] = npy(loss)
    for pre, m in mods.items():
        for k, v in m.state_dict().items():
            out['sd2.' + pre + k] = npy(v)
    np.savez_compressed(os.path.join(HERE, 'composed_step.npz'), **out)


if __name__ == '__main__':
    loss_fixtures()
    cnn_fixture('b8_2222', [2, 2, 2, 2], 8, 4, 224, 1)
    cnn_fixture('b16_1111', [1, 1, 1, 1], 16, 6, 96, 5)
    transformer_fixture()
    lstm_fixture()
    composed_step_fixture()
    assert not os.path.exists('/root/reference/src/__pycache__'), "bytecode leaked into the reference tree"
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, 'KiB')
