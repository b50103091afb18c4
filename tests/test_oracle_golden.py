"""CPU: pin the oracle (oracle/*.py) against the fixtures generated from the reference's own
modules (tests/golden/make_golden.py).  fp32 throughout; tolerances are summation-order level."""
import numpy as np
import pytest
import torch

from oracle import coordination as C
from oracle import model as M
from oracle.profile_encoder import (profile_cnn_features, profile_cnn_forward, profile_lstm_forward,
                                    profile_transformer_forward, transformer_tokenize)

T = torch.from_numpy
RTOL, ATOL = 2e-5, 2e-6


def close(a, b, rtol=RTOL, atol=ATOL):
    np.testing.assert_allclose(a.detach().numpy() if torch.is_tensor(a) else a, b, rtol=rtol, atol=atol)


def _loss_call(name, a, p, params, k):
    if name == 'clip':
        return C.clip_loss(a, p, params['logit_scale'], k)
    if name == 'siglip':
        return C.siglip_loss(a, p, params['logit_scale'], params['bias'], k)
    if name == 'clipplus':
        return C.clip_plus(a, p, params['clip.logit_scale'], k, .25)
    return C.siglip_plus(a, p, params['siglip.logit_scale'], params['siglip.bias'], k, .25)


@pytest.mark.parametrize('ci', range(5))
@pytest.mark.parametrize('name', ['clip', 'siglip', 'clipplus', 'siglipplus'])
def test_losses_match_reference(golden, ci, name):
    g = golden('losses')
    b, d, k = g[f'case{ci}_shape']
    a = T(g[f'case{ci}_image_emb']).requires_grad_(True)
    p = T(g[f'case{ci}_profile_emb']).requires_grad_(True)
    pre = f'case{ci}_{name}_'
    params = {key[len(pre + 'param_'):]: T(g[key]).clone().requires_grad_(True)
              for key in g if key.startswith(pre + 'param_')}
    loss = _loss_call(name, a, p, params, int(k))
    loss.backward()
    close(loss, g[pre + 'loss'])
    close(a.grad, g[pre + 'd_image'], rtol=1e-4, atol=1e-7)
    close(p.grad, g[pre + 'd_profile'], rtol=1e-4, atol=1e-7)
    for pn, pv in params.items():
        close(pv.grad, g[pre + 'dparam_' + pn], rtol=1e-4, atol=1e-6)
    init = {kk: torch.tensor(1.0 if kk.endswith('logit_scale') else -10.0) for kk in params}
    close(_loss_call(name, a.detach(), p.detach(), init, int(k)), g[pre + 'loss_init'])


@pytest.mark.parametrize('ci', range(5))
def test_rank_loss_matches_reference(golden, ci):
    g = golden('losses')
    a = T(g[f'case{ci}_image_emb']).requires_grad_(True)
    p = T(g[f'case{ci}_profile_emb']).requires_grad_(True)
    loss = C.rank_loss(a, p, .25)
    loss.backward()
    close(loss, g[f'case{ci}_rank_loss'])
    close(a.grad, g[f'case{ci}_rank_d_image'], rtol=1e-4, atol=1e-7)
    close(p.grad, g[f'case{ci}_rank_d_profile'], rtol=1e-4, atol=1e-7)


def test_survey_spot_values(golden):
    g = golden('losses')
    torch.manual_seed(0)
    a, p = torch.randn(64, 512), torch.randn(64, 512)
    one = torch.ones([])
    close(torch.stack([C.clip_loss(a, p, one, 1), C.clip_loss(a, p, one, 4)]), g['survey_clip'])
    close(torch.stack([C.siglip_loss(a, p, one, -10 * one, 1), C.siglip_loss(a, p, one, -10 * one, 4)]),
          g['survey_siglip'])
    close(C.rank_loss(a, p, .25), g['survey_rank'][0])
    # the values printed in SURVEY.md section 8a13-a16
    assert abs(g['survey_clip'][0] - 4.176510334) < 1e-6 and abs(g['survey_clip'][1] - 2.793492317) < 1e-6
    assert abs(g['survey_siglip'][0] - 10.015448570) < 1e-5 and abs(g['survey_rank'][0] - 0.277970612) < 1e-6


def test_big_loss_and_margin_indices(golden):
    g = golden('losses')
    rs = np.random.RandomState
    a = T(rs(900).standard_normal((512, 512)).astype(np.float32)).requires_grad_(True)
    p = T(rs(901).standard_normal((512, 512)).astype(np.float32)).requires_grad_(True)
    ls = torch.ones([], requires_grad=True)
    loss = C.clip_loss(a, p, ls, 1)
    loss.backward()
    close(loss, g['big_clip_loss'])
    close(a.grad[:4], g['big_clip_d_image_rows'], rtol=1e-4, atol=1e-8)
    close(p.grad[-4:], g['big_clip_d_profile_rows'], rtol=1e-4, atol=1e-8)
    close(ls.grad, g['big_clip_dscale'], rtol=1e-4)
    assert abs(a.grad.double().abs().sum().item() / g['big_clip_d_image_abs_sum'] - 1) < 1e-5
    r, c = C.retrieval_top1(T(g['margin_image_emb']), T(g['margin_profile_emb']))
    assert torch.equal(r, T(g['margin_row_argmax'])) and torch.equal(c, T(g['margin_col_argmax']))
    close(C.clip_loss(T(g['margin_image_emb']), T(g['margin_profile_emb']), torch.ones([]), 1), g['margin_clip_loss'])


def _sd(g, prefix='sd.'):
    return {k[len(prefix):]: T(v.copy()) for k, v in g.items() if k.startswith(prefix)}


@pytest.mark.parametrize('tag', ['b8_2222', 'b16_1111'])
def test_profile_cnn_matches_reference(golden, tag):
    g = golden('profile_cnn_' + tag)
    blocks = [int(b) for b in g['blocks']]
    x, plen, wsum = T(g['profile']), T(g['profile_len']), T(g['wsum'])
    sd = _sd(g)
    close(profile_cnn_features(sd, x, blocks), g['eval.features'], rtol=1e-4, atol=1e-5)
    close(profile_cnn_forward(sd, x, plen, blocks), g['eval.out'], rtol=1e-4, atol=1e-5)
    params = {k: v.requires_grad_(True) for k, v in sd.items() if M.is_param(k)}
    y = profile_cnn_forward(sd, x, plen, blocks, train=True)
    close(y, g['train.out'], rtol=1e-4, atol=1e-5)
    (y * wsum).sum().backward()
    for k, v in params.items():
        close(v.grad, g['train.grad.' + k], rtol=2e-3, atol=2e-5)
    for k in g:
        if k.startswith('train.after.'):
            close(sd[k[len('train.after.'):]], g[k], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('tag', ['ragged', 'fixed'])
def test_profile_transformer_matches_reference(golden, tag):
    g = golden('profile_transformer')
    sd = _sd(g)
    lens = [int(n) for n in g[f'{tag}.lens']]
    profs = [T(g[f'{tag}.in{i}']) for i in range(len(lens))]
    tok = transformer_tokenize(profs, int(g['padding_idx']))
    for k in ('profile', 'time', 'padding_mask'):
        assert np.array_equal(tok[k].numpy(), g[f'{tag}.tok.{k}']), k
    params = {k: v.requires_grad_(True) for k, v in sd.items()}
    plen = torch.tensor([[n] for n in lens])
    y = profile_transformer_forward(sd, tok['profile'], tok['time'], tok['padding_mask'], plen, 2, 2)
    close(y, g[f'{tag}.out'], rtol=1e-4, atol=1e-5)
    (y * T(g[f'{tag}.wsum'])).sum().backward()
    for k, v in params.items():
        ref = g[f'{tag}.grad.{k}']
        got = v.grad if v.grad is not None else torch.zeros_like(v)
        close(got, ref, rtol=2e-3, atol=2e-5)


def test_profile_lstm_matches_reference(golden):
    g = golden('profile_lstm')
    sd = _sd(g)
    y = profile_lstm_forward(sd, T(g['tok.profile']), T(g['tok.last_idx']), T(g['profile_len']), 2)
    close(y, g['out'], rtol=1e-4, atol=1e-6)


def test_composed_step_matches_reference(golden):
    """ProfileCNN -> projection || image features -> projection -> CLIP(buckets=2) -> 2x SGD(nesterov, wd)."""
    g = golden('composed_step')
    sd = _sd(g, 'sd0.')
    bufs = {}
    optim = dict(lr=5e-2, momentum=0.9, weight_decay=1e-3, nesterov=True)
    for step in range(2):
        params = {k: v for k, v in sd.items() if M.is_param(k) and v.is_floating_point()}
        for v in params.values():
            v.requires_grad_(True)
        feat = profile_cnn_forward(M.sub(sd, 'profile_encoder.'), T(g[f'step{step}.profile']),
                                   T(g[f'step{step}.profile_len']), [1, 1, 1, 1], train=True)
        loss = C.clip_loss(T(g[f'step{step}.image_feat']) @ sd['image_projection.weight'].T,
                           feat @ sd['profile_projection.weight'].T, sd['loss.logit_scale'], 2)
        grads = dict(zip(params, torch.autograd.grad(loss, list(params.values()))))
        for v in params.values():
            v.requires_grad_(False)
        M.sgd_update(params, grads, bufs, **optim)
        close(loss, g[f'step{step}.loss'], rtol=1e-5)
    for k, v in sd.items():
        close(v, g['sd2.' + k], rtol=1e-4, atol=2e-6)
