"""GPU augmentation kernels against the host transforms of data.py, decision by decision."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def test_gpu_augmentation_matches_host_transforms():
    from multimodal_plankton_recognition_amd.augment import GpuAugment
    from multimodal_plankton_recognition_amd import data as D
    T = 48
    aug = GpuAugment(T, noise=0.0, seed=3)
    S = aug.S
    g = torch.Generator().manual_seed(1)
    B = 9
    u8 = torch.randint(0, 256, (B, S, S), generator=g, dtype=torch.uint8)
    lens = [5, 40, 51, 51, 200, 333, 17, 50, 64]
    raws = [torch.rand(n, 6, generator=g).mul(9).exp().sub(1).clamp_min(0) for n in lens]
    raw = torch.zeros(B, max(lens), 6)
    for i, r in enumerate(raws):
        raw[i, :r.shape[0]] = r
    d = aug.draw(B)
    d['vflip'][:2] = torch.tensor([0, 1], dtype=torch.uint8)
    d['pair_flip'][:4] = torch.tensor([0, 1, 0, 1], dtype=torch.uint8)
    img, prof = aug(u8.to(DEV), raw.to(DEV), torch.tensor(lens), decisions=d)
    for i in range(B):
        # host chain of data.ImageTransformTrain / PairAugmentation on the same cached bytes and decisions
        x = (u8[i].float() / 255.0).unsqueeze(0) * 2 - 1
        t, l = int(d['top'][i]), int(d['left'][i])
        x = x[:, t:t + T, l:l + T]
        if d['vflip'][i]:
            x = x.flip(1)
        if d['pair_flip'][i]:
            x = x.flip(-1)
        assert torch.equal(img[i].cpu(), x), i                                   # bit-exact
        p = D._profile_base(raws[i].numpy(), S)                                   # [C, S]
        pl = int(d['prof_left'][i])
        p = p[:, pl:pl + T].t()
        if d['pair_flip'][i]:
            p = p.flip(0)
        np.testing.assert_allclose(prof[i].cpu().numpy(), p.float().numpy(), rtol=0, atol=5e-6, err_msg=str(i))


def test_gpu_augmentation_random_decisions_and_noise():
    from multimodal_plankton_recognition_amd.augment import GpuAugment
    T = 32
    aug = GpuAugment(T, noise=1e-3, seed=0)
    S = aug.S
    B = 64
    u8 = torch.randint(0, 256, (B, S, S), dtype=torch.uint8, device=DEV)
    raw = torch.rand(B, 100, 6, device=DEV) * 1000
    lengths = torch.full((B,), 100)
    d = aug.draw(B)
    assert int(d['top'].max()) <= S - T and int(d['vflip'].max()) <= 1 and 0 < int(d['pair_flip'].sum()) < B
    a, p = aug(u8, raw, lengths, decisions=d)
    assert a.shape == (B, 1, T, T) and p.shape == (B, T, 6)
    assert float(a.min()) >= -1 and float(a.max()) <= 1
    clean = aug.profiles(raw, lengths, d, noise=0.0)
    resid = (p - clean).flatten()
    assert abs(float(resid.std()) - 1e-3) < 1e-4 and abs(float(resid.mean())) < 1e-4
    again = aug.profiles(raw, lengths, d)
    assert torch.equal(again, p)                                                   # noise is a function of the seed
