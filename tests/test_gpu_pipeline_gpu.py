"""The device-side input pipeline end to end: an on-disk dataset in the reference's layout (train.csv /
test.csv with image / profile / class columns, JPEG-like images with a 25-px scale bar, CSV profiles) read through
CachedMultiSet + cached_collate + DevicePipeline, against the host MultiSet with the reference's test transforms; and the
CLI with --gpu-augment."""
import glob
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = 'cuda'


def _make_dataset(root, n=24, seed=0):
    from PIL import Image
    rs = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, 'images'))
    os.makedirs(os.path.join(root, 'profiles'))
    rows = []
    for i in range(n):
        h, w = rs.randint(60, 140), rs.randint(60, 180)
        g = rs.randint(0, 255, (h + 25, w), dtype=np.uint8)
        Image.fromarray(np.stack([g, g, g], -1)).save(os.path.join(root, 'images', f'{i}.png'))      # lossless: exact bytes
        L = rs.randint(20, 300)
        prof = rs.uniform(0, 4000, (L, 6))
        np.savetxt(os.path.join(root, 'profiles', f'{i}.csv'), prof, delimiter=',', header='a,b,c,d,e,f', comments='')
        rows.append((f'images/{i}.png', f'profiles/{i}.csv', f'class_{i % 3}'))
    import pandas as pd
    for name in ('train.csv', 'test.csv'):
        pd.DataFrame(rows, columns=['image', 'profile', 'class']).to_csv(os.path.join(root, name))
    return root


def test_cached_eval_pipeline_matches_host_test_transforms(tmp_path):
    from multimodal_plankton_recognition_amd import data as D
    from multimodal_plankton_recognition_amd.augment import DevicePipeline
    from multimodal_plankton_recognition_amd.model import MultiModel
    root = _make_dataset(str(tmp_path / 'ds'))
    T = 48
    model = MultiModel(dim_embed=32, image_encoder_args=dict(name='resnet18'),
                       profile_encoder_args=dict(dim_in=6, blocks=[1, 1, 1, 1], base_channels=8),
                       coordination_args=dict(method='clip'), optim_args=dict(lr=1e-3))
    host = D.MultiSet(os.path.join(root, 'test.csv'), D.ImageTransformTest(T), D.ProfileTransformTest(T))
    cached = D.CachedMultiSet(os.path.join(root, 'test.csv'), T, train=False)
    idx = list(range(8))
    hb = D.make_multi_collate(model, 1)([host[i] for i in idx])
    cb = D.cached_collate([cached[i] for i in idx])
    out = DevicePipeline(model, T, 1)({k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in cb.items()}, training=False)
    assert set(out) == set(hb)
    assert torch.equal(out['image_shape'].cpu(), hb['image_shape']) and torch.equal(out['profile_len'].cpu(), hb['profile_len'])
    # images: the cache converts to luma bytes before the Lanczos resize, the host chain resizes RGB and weights the channels
    # afterwards (0.2989 + 0.587 + 0.114 = 0.9999): one grey level of 255 apart at most
    assert float((out['image'].cpu() - hb['image']).abs().max()) <= 2.0 / 255 * 2 + 1e-6
    np.testing.assert_allclose(out['profile'].cpu().numpy(), hb['profile'].numpy(), rtol=0, atol=5e-6)
    # training mode: shapes, ranges, and a different crop / flip draw per call
    tr = D.CachedMultiSet(os.path.join(root, 'train.csv'), T, train=True)
    tb = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in D.cached_collate([tr[i] for i in idx]).items()}
    pipe = DevicePipeline(model, T, 1)
    a, b = pipe(tb, True), pipe(tb, True)
    assert a['image'].shape == (8, 1, T, T) and a['profile'].shape == (8, T, 6) and a['buckets'] == 1
    assert float(a['image'].min()) >= -1 and float(a['image'].max()) <= 1 and not torch.equal(a['image'], b['image'])


def test_train_multi_cli_with_gpu_augment(tmp_path):
    root = _make_dataset(str(tmp_path / 'plankton' / 'fold0'), n=40)
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, 'train_multi.py', '-d', root, '-m', '../model_cards/smoke_multi.yaml', '--gpu-augment',
           '--max-epochs', '2', '--logdir', str(tmp_path / 'logs')]
    out = subprocess.run(cmd, cwd=os.path.join(ROOT, 'scripts'), env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    run = os.path.join(str(tmp_path / 'logs'), 'smoke_multi_plankton_fold0', 'version_0')        # <card>_<last two dirs of -d>
    lines = [json.loads(l) for l in open(os.path.join(run, 'metrics.jsonl'))]
    assert any('train_loss' in l for l in lines) and all(np.isfinite(l.get('valid_loss', 0.0)) for l in lines)
    assert glob.glob(os.path.join(run, 'checkpoints', 'epoch=*_valid_loss=*.ckpt'))
