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


def test_val_check_interval_follows_lightning():
    """Trainer(val_check_interval=...) as lightning.Trainer reads it (the reference passes the card's value through:
    scripts/train_multi.py:99-104): float = fraction of the training epoch, int = batches, None / 1.0 = end of epoch."""
    loader = list(range(10))
    assert Trainer(device='cpu')._val_interval(loader) is None
    assert Trainer(device='cpu', val_check_interval=1.0)._val_interval(loader) is None
    assert Trainer(device='cpu', val_check_interval=0.25)._val_interval(loader) == 2
    assert Trainer(device='cpu', val_check_interval=0.01)._val_interval(loader) == 1
    assert Trainer(device='cpu', val_check_interval=4)._val_interval(loader) == 4
    assert Trainer(device='cpu', val_check_interval=0.5, limit_train_batches=4)._val_interval(loader) == 2
    for bad in (0, -1, 1.5, 0.0, 11, 'often', True):
        with pytest.raises(ValueError):
            Trainer(device='cpu', val_check_interval=bad)._val_interval(loader)
    with pytest.raises(ValueError):
        Trainer(device='cpu', val_check_interval=0.5)._val_interval(iter(loader))


def test_trainer_runs_validation_inside_the_epoch_and_keeps_accumulation_windows():
    """Host logic of Trainer.fit on a CPU double: val_check_interval = 2 batches -> two validation runs in a 5-batch epoch
    (none added at the epoch's end), accumulate_grad_batches = 2 -> optimizer steps after batches 2, 4 and the short last
    window (Lightning steps on the epoch's last batch)."""
    class Double(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(()))
            self.events, self.current_epoch = [], 0

        def configure_optimizers(self):
            outer = self

            class Opt(torch.optim.SGD):
                def step(self, *a, **k):
                    outer.events.append('step')
                    return super().step(*a, **k)
            return Opt(self.parameters(), lr=0.1)

        def training_step(self, batch, i):
            self.events.append(f'train{i}')
            return (self.w - batch['x']) ** 2

        def validation_step(self, batch, i):
            self.events.append('val')

        def on_train_epoch_end(self):
            self.events.append('epoch_end')

        def on_validation_epoch_end(self):
            self.events.append('val_end')

    m = Double()
    import multimodal_plankton_recognition_amd.trainer as TR
    old = TR.ops.backward
    TR.ops.backward = lambda loss: loss.backward()          # (the product's cached-root backward needs a device)
    try:
        t = Trainer(device='cpu', max_epochs=1, accumulate_grad_batches=2, val_check_interval=2)
        t.fit(m, [{'x': torch.tensor(float(i))} for i in range(5)], [{'x': torch.tensor(0.)}])
    finally:
        TR.ops.backward = old
    assert m.events == ['train0', 'train1', 'step', 'val', 'val_end', 'train2', 'train3', 'step', 'val', 'val_end', 'train4',
                        'step', 'epoch_end'], m.events
    assert t.global_step == 3


def test_backbone_with_a_live_block_chain_can_be_copied_and_pickled():
    """BlockChain holds non-leaf tensors of the last forward; a deep copy / pickle of the module gets an empty
    table shared by the copy's blocks."""
    import copy
    import pickle
    from multimodal_plankton_recognition_amd.image_encoder import ResNetBackbone
    m = ResNetBackbone((1, 1, 1, 1))
    m._chain.notes[5] = (torch.zeros(2, requires_grad=True) * 2, None, None)
    for m2 in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
        assert m2._chain is not m._chain and not m2._chain.notes and not m2._chain.sums
        assert all(b.chain is m2._chain for li in range(1, 5) for b in getattr(m2, f'layer{li}'))
    assert 5 in m._chain.notes


def test_conv_precision_switch():
    from multimodal_plankton_recognition_amd import layers_f32
    old = layers_f32._PRECISION[0]
    try:
        for name, want in (('32', True), ('32-true', True), (32, True), ('16-mixed', False), ('bf16-mixed', False),
                           (None, False)):
            layers_f32.set_conv_precision(name)
            assert layers_f32.conv_f32() is want, name
    finally:
        layers_f32._PRECISION[0] = old
