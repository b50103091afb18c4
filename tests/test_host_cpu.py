"""CPU: host-side pieces around the hot path -- trainer callbacks / checkpoint naming, data schema, collate,
model cards.  (No kernel runs here.)"""
import json
import os

import numpy as np
import pytest
import torch
import yaml

from multimodal_plankton_recognition_amd import data as D
from multimodal_plankton_recognition_amd.trainer import EarlyStopping, ModelCheckpoint, TensorBoardLogger, Trainer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_checkpoint_name_matches_lightning_rendering():
    ck = ModelCheckpoint(filename="{epoch}_{valid_loss:.5f}", monitor="valid_loss", save_top_k=2, mode="min")
    assert ck.format_name(86, {'valid_loss': 0.923021}) == 'epoch=86_valid_loss=0.92302.ckpt'   # experiments.ipynb:134


def test_topk_checkpoints_and_early_stopping(tmp_path):
    logger = TensorBoardLogger(str(tmp_path), 'run')
    assert logger.log_dir.endswith('version_0')
    assert TensorBoardLogger(str(tmp_path), 'run').log_dir.endswith('version_1')
    tr = Trainer(logger=logger, device='cpu')
    model = torch.nn.Linear(2, 2)
    ck = ModelCheckpoint(filename="{epoch}_{valid_loss:.5f}", save_top_k=2)
    es = EarlyStopping(patience=2, check_finite=False)
    for epoch, loss in enumerate([1.0, 0.5, 0.7, 0.4, 0.9, 0.95]):
        tr.current_epoch = epoch
        ck.on_validation_end(tr, model, {'valid_loss': loss})
        es.on_validation_end(tr, model, {'valid_loss': loss})
    kept = sorted(os.listdir(os.path.join(logger.log_dir, 'checkpoints')))
    assert kept == ['epoch=1_valid_loss=0.50000.ckpt', 'epoch=3_valid_loss=0.40000.ckpt']
    assert tr.should_stop                                           # two epochs without improvement after 0.4
    ckpt = torch.load(os.path.join(logger.log_dir, 'checkpoints', kept[0]), weights_only=False)
    assert set(ckpt) >= {'state_dict', 'hyper_parameters', 'epoch'}
    es2 = EarlyStopping(patience=1, check_finite=False)             # reference: check_finite=False -> NaN never stops
    tr2 = Trainer(device='cpu')
    es2.on_validation_end(tr2, model, {'valid_loss': float('nan')})
    assert not tr2.should_stop


def test_synthetic_dataset_schema_and_collate():
    ds = D.SyntheticMultiSet(8, target_size=32)
    s = ds[3]
    assert list(s) == ['image', 'profile', 'label', 'image_shape', 'profile_length']     # src/data.py:57-59 order
    assert s['image'].shape == (1, 32, 32) and s['profile'].shape == (32, 6)
    assert s['image'].min() >= -1 and s['image'].max() <= 1
    assert torch.equal(ds[3]['image'], s['image'])                                        # deterministic per index

    class _Enc:
        def tokenize(self, p):
            return {'profile': torch.stack(list(p))}

    class _M:
        profile_encoder = _Enc()
    batch = D.make_multi_collate(_M(), buckets=2)([ds[i] for i in range(4)])
    assert set(batch) == {'image', 'profile', 'image_shape', 'profile_len', 'buckets'}
    assert batch['image'].shape == (4, 1, 32, 32) and batch['profile_len'].shape == (4, 1)
    assert batch['image_shape'].dtype == torch.int64 and batch['buckets'] == 2


def test_transforms_follow_reference_ranges(tmp_path):
    from PIL import Image
    rs = np.random.RandomState(0)
    img = Image.fromarray(rs.randint(0, 255, (125, 300, 3), dtype=np.uint8))
    x = D.ImageTransformTest(64)(img)
    assert x.shape == (1, 64, 64) and -1 <= float(x.min()) and float(x.max()) <= 1
    assert D.ImageTransformTrain(64)(img).shape == (1, 64, 64)
    prof = rs.uniform(0, 5000, (93, 6))
    p = D.ProfileTransformTest(64)(prof)
    assert p.shape == (64, 6) and p.dtype == torch.float32
    assert D.ProfileTransformTrain(64)(prof).shape == (64, 6)
    a, b = D.PairAugmentation()(x, p)
    assert a.shape == x.shape and b.shape == p.shape


@pytest.mark.parametrize('card', ['resnet18_cnn_2_512_clip.yaml', 'example_multi.yaml', 'smoke_multi.yaml',
                                  'vit_base_transformer_siglip.yaml', 'vit_base_transformer_clip.yaml',
                                  'vit_base_transformer_base_clip.yaml'])
def test_model_cards_follow_the_schema_the_script_reads(card):
    from multimodal_plankton_recognition_amd.model import MultiModel
    c = yaml.safe_load(open(os.path.join(ROOT, 'model_cards', card)))
    for key in ('target_size', 'bs', 'dim_embedding', 'buckets', 'num_workers', 'patience', 'trainer_args'):
        assert key in c, key
    m = MultiModel(c['dim_embedding'], c['image_encoder_args'], c['profile_encoder_args'], c['coordination_args'],
                   c['optim_args'])
    assert m.image_projection.weight.shape[0] == c['dim_embedding']
    assert set(m.hparams) == {'dim_embed', 'image_encoder_args', 'profile_encoder_args', 'coordination_args',
                              'optim_args'}


def test_single_modality_cards_build_their_models():
    """BASELINE configs C1 / C2: the model half of the (out-of-scope) single-modality scripts."""
    from multimodal_plankton_recognition_amd.model import ImageModel, ProfileModel
    names = [f'class{i}' for i in range(7)]
    c1 = yaml.safe_load(open(os.path.join(ROOT, 'model_cards', 'example_profile.yaml')))
    pm = ProfileModel(c1['profile_encoder_args'], c1['optim_args'], names)
    assert type(pm.profile_encoder).__name__ == 'ProfileTransformer'
    c2 = yaml.safe_load(open(os.path.join(ROOT, 'model_cards', 'example_image.yaml')))
    im = ImageModel(c2['image_encoder_args'], c2['optim_args'], names)
    assert im.image_encoder.dim_out == 512 + 2


def _aa_resize_restated(x, out_len):
    """numpy restatement of aten::_upsample_bilinear2d_aa along one axis (what torchvision's tensor Resize computes,
    /root/reference/src/data.py:133,152): triangle filter, half-width max(scale, 1), weights normalised per output."""
    import numpy as np
    L = x.shape[-1]
    scale = L / out_len
    support = scale if scale >= 1 else 1.0
    inv = 1.0 / scale if scale >= 1 else 1.0
    out = np.zeros(x.shape[:-1] + (out_len,))
    for j in range(out_len):
        center = scale * (j + 0.5)
        xmin = max(0, int(center - support + 0.5))
        xmax = min(L, int(center + support + 0.5))
        w = np.array([max(0.0, 1.0 - abs((i - center + 0.5) * inv)) for i in range(xmin, xmax)])
        out[..., j] = (x[..., xmin:xmax] * w).sum(-1) / w.sum()
    return out


@pytest.mark.parametrize('L,T', [(1000, 236), (300, 224), (17, 236), (224, 224), (5, 64)])
def test_profile_resize_is_torchvisions_antialiased_bilinear(L, T):
    """The profile resize of ProfileTransformTrain / Test: long profiles are area-averaged (anti-aliasing), short ones
    plainly interpolated -- against an independent restatement of the filter."""
    import numpy as np
    from multimodal_plankton_recognition_amd import data as D
    rs = np.random.RandomState(L)
    prof = rs.rand(L, 6) * 2000
    got = D._profile_base(prof, T).numpy()                                         # [6, T]
    base = np.log(prof + 1) / np.array(D.PROFILE_CEIL)[None, :] * 2 - 1
    ref = _aa_resize_restated(base.T, T)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-12)
    if L > 2 * T:                                                                  # really an average: far from 2-point sampling
        two_point = torch.nn.functional.interpolate(torch.from_numpy(base.T)[None], size=T, mode='linear',
                                                    align_corners=False)[0].numpy()
        assert np.abs(got - two_point).max() > 1e-2
    assert D.ProfileTransformTest(T)(prof).dtype == torch.float32
